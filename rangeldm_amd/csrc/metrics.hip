// metrics.hip -- BEV-histogram evaluation of generated point clouds on gfx950: the JSD / MMD numbers
// the reference publishes (metrics/metric.py:141-156).
//
//   bev_histogram_kernel   load_point_cloud_xyz depth mask + point_cloud_to_histogram (np.histogramdd, 100 x 100 bins
//                          over +-80 m)                          metrics/metrics/histogram/mmd.py:39-44, histogram.py:4-18
//   hist_colsum / jsd      jsd_2d over the summed, normalised histograms (scipy jensenshannon, natural log)
//                                                                metrics/metrics/histogram/jsd.py:14-16,90-101
//   spectral_sq_kernel     the `gaussian` kernel's distance: np.linalg.norm(x - y, 2) of two 2-D pmfs is the SPECTRAL norm
//                          (largest singular value) of their difference   metrics/metrics/histogram/dist_helper.py:84-104
//   mmd_reduce_kernel      compute_mmd: s1 + s2 - 2 cross with k = exp(-d^2 / (2 sigma^2))   dist_helper.py:156-172
//
// Integer work (bin indices, counts) is bit-exact: bin edges are the float64 linspace values numpy uses and the candidate
// bin is corrected against them.  The spectral norm is fp32 (VALU FMA over LDS-resident 100 x 100 matrices): Gram matrix,
// S trace-normalised squarings (power 2^S), two polishing products and a Rayleigh quotient on the Gram matrix -- the
// quotient's error is the SQUARE of the eigenvector error, worst case 1 / (2e 2^S) relative.
#include "common.h"
#include "../../include/rangeldm_hip.h"

#include <cmath>

#pragma clang fp contract(off)      // (+ -ffp-contract=off in the Makefile) edges and depths are single IEEE operations

namespace {

// ---- histogram ----------------------------------------------------------------------------------------------------
__device__ inline double edge_at(int i, int bins, double lo, double hi, double step) {
    return i == bins ? hi : (double)i * step + lo;          // np.linspace: arange * step + start, last = stop
}

// one thread per point; sample s owns points [offsets[s], offsets[s + 1])
__global__ __launch_bounds__(256) void bev_histogram_kernel(const float* __restrict__ pts, const int* __restrict__ offsets,
                                                            int S, int stride, double lo, double hi, double step, int bins,
                                                            float min_depth, float max_depth, unsigned* __restrict__ hist) {
    const int s = blockIdx.y;
    const int begin = offsets[s], end = offsets[s + 1];
    for (int i = begin + blockIdx.x * 256 + threadIdx.x; i < end; i += gridDim.x * 256) {
        const float* p = pts + (size_t)i * stride;
        const float x = p[0], y = p[1], z = p[2];
        const float d = sqrtf((x * x + y * y) + z * z);          // np.linalg.norm(pc[:, :3], 2, axis=1) in fp32
        if (!(d > min_depth && d < max_depth)) continue;
        int b[2];
        bool ok = true;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const double v = (double)(a == 0 ? x : y);
            if (!(v >= lo && v <= hi)) { ok = false; break; }
            int k = (int)floor((v - lo) / step);
            k = k < 0 ? 0 : (k > bins - 1 ? bins - 1 : k);
            while (k < bins - 1 && edge_at(k + 1, bins, lo, hi, step) <= v) ++k;      // searchsorted(side='right') - 1,
            while (k > 0 && edge_at(k, bins, lo, hi, step) > v) --k;                   // v == hi lands in the last bin
            b[a] = k;
        }
        if (ok) atomicAdd(hist + ((size_t)s * bins + b[0]) * bins + b[1], 1u);
    }
}

// ---- JSD ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hist_colsum_kernel(const unsigned* __restrict__ h, int S, int nb,
                                                          unsigned long long* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nb) return;
    unsigned long long acc = 0;
    for (int s = 0; s < S; ++s) acc += h[(size_t)s * nb + i];
    out[i] = acc;
}

__device__ inline double block_sum(double v, double* sh) {
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    return t;
}

__global__ __launch_bounds__(1024) void jsd_kernel(const unsigned long long* __restrict__ px,
                                                   const unsigned long long* __restrict__ py, int nb, double* __restrict__ out) {
    __shared__ double sh[16];
    double sx = 0.0, sy = 0.0;
    for (int i = threadIdx.x; i < nb; i += 1024) { sx += (double)px[i]; sy += (double)py[i]; }
    sx = block_sum(sx, sh);
    sy = block_sum(sy, sh);
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += 1024) {
        const double p = (double)px[i] / sx, q = (double)py[i] / sy, m = 0.5 * (p + q);
        if (p > 0.0) acc += p * log(p / m);                      // scipy.special.rel_entr
        if (q > 0.0) acc += q * log(q / m);
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = sqrt(acc / 2.0);
}

// ---- spectral norm of pmf differences -------------------------------------------------------------------------------
constexpr int NBMAX = 104;          // bins <= 104 (three NBMAX x NBMAX fp32 matrices in LDS = 130 KB)
constexpr int PITCH = NBMAX;

// C = (A^T A) * scale, n x n, all in LDS.  For symmetric A this is A^2.  A thread owns 4 x 4 blocks; both operands of a
// k-step are float4 reads of row k.
__device__ inline void ata(const float* __restrict__ A, float* __restrict__ C, int n, float scale) {
    const int nb4 = n >> 2, nblk = nb4 * nb4;
    for (int blk = threadIdx.x; blk < nblk; blk += blockDim.x) {
        const int r0 = (blk / nb4) << 2, c0 = (blk % nb4) << 2;
        float acc[4][4] = {};
        for (int k = 0; k < n; ++k) {
            const float4 a = *reinterpret_cast<const float4*>(A + k * PITCH + r0);
            const float4 b = *reinterpret_cast<const float4*>(A + k * PITCH + c0);
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(C + (r0 + i) * PITCH + c0) =
                make_float4(acc[i][0] * scale, acc[i][1] * scale, acc[i][2] * scale, acc[i][3] * scale);
    }
}

__device__ inline float block_sum_f(float v, float* sh) {
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    return t;
}
__device__ inline float block_max_f(float v, float* sh) {
    for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t = fmaxf(t, sh[w]);
    return t;
}

// grid (ny, nx): pair (i = blockIdx.y of X, j = blockIdx.x of Y); symmetric: only j > i is computed (diagonal = 0, the rest
// mirrored by the reduction).  out[i * ny + j] = sigma_max(x_i / sum(x_i) - y_j / sum(y_j))^2.
template <int SQUARINGS>
__global__ __launch_bounds__(256) void spectral_sq_kernel(const unsigned* __restrict__ hx, const unsigned* __restrict__ hy,
                                                          const double* __restrict__ inv_sx, const double* __restrict__ inv_sy,
                                                          int n, int symmetric, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[8];
    const int i = blockIdx.y, j = blockIdx.x;
    if (symmetric && j <= i) return;
    float* G = lds;
    float* P = lds + NBMAX * PITCH;
    float* Q = P + NBMAX * PITCH;
    const unsigned* x = hx + (size_t)i * n * n;
    const unsigned* y = hy + (size_t)j * n * n;
    const double ix = inv_sx[i], iy = inv_sy[j];
    // D (into P), scaled to max |D| = 1
    float mx = 0.f;
    for (int e = threadIdx.x; e < n * n; e += 256) {
        const float d = (float)((double)x[e] * ix - (double)y[e] * iy);
        P[(e / n) * PITCH + (e % n)] = d;
        mx = fmaxf(mx, fabsf(d));
    }
    mx = block_max_f(mx, red);
    if (mx == 0.f) {
        if (threadIdx.x == 0) out[(size_t)i * gridDim.x + j] = 0.f;
        return;
    }
    const float inv = 1.f / mx;
    for (int e = threadIdx.x; e < n * n; e += 256) P[(e / n) * PITCH + (e % n)] *= inv;
    __syncthreads();
    ata(P, G, n, 1.f);                                   // G = D^T D
    __syncthreads();
    const float* src = G;
    float* dst = P;
    for (int s = 0; s < SQUARINGS; ++s) {
        float tr = 0.f;
        for (int d = threadIdx.x; d < n; d += 256) tr += src[d * PITCH + d];
        tr = block_sum_f(tr, red);                       // trace >= lambda_max: entries of src / tr are <= 1
        const float sc = 1.f / tr;
        ata(src, dst, n, sc * sc);
        __syncthreads();
        src = dst;
        dst = (dst == P) ? Q : P;
    }
    // v = the column of the power matrix with the largest diagonal entry, polished by two more products
    float* v = dst;                                      // scratch rows of the free buffer
    float* u = dst + PITCH;
    float best = -1.f;
    int arg = 0;
    for (int d = threadIdx.x; d < n; d += 256) {
        const float t = src[d * PITCH + d];
        if (t > best) { best = t; arg = d; }
    }
    const float bmax = block_max_f(best, red);
    __shared__ int s_arg;
    if (threadIdx.x == 0) s_arg = n;
    __syncthreads();
    if (best == bmax) atomicMin(&s_arg, arg);
    __syncthreads();
    const int col = s_arg;
    for (int d = threadIdx.x; d < n; d += 256) v[d] = src[d * PITCH + col];
    __syncthreads();
    for (int it = 0; it < 2; ++it) {
        float nrm = 0.f;
        for (int d = threadIdx.x; d < n; d += 256) nrm = fmaxf(nrm, fabsf(v[d]));
        nrm = block_max_f(nrm, red);
        const float sc = 1.f / nrm;
        for (int r = threadIdx.x; r < n; r += 256) {
            float acc = 0.f;
            for (int k = 0; k < n; ++k) acc = fmaf(src[r * PITCH + k], v[k] * sc, acc);
            u[r] = acc;
        }
        __syncthreads();
        float* t = v; v = u; u = t;
    }
    // Rayleigh quotient on G
    float num = 0.f, den = 0.f;
    for (int r = threadIdx.x; r < n; r += 256) {
        float acc = 0.f;
        for (int k = 0; k < n; ++k) acc = fmaf(G[r * PITCH + k], v[k], acc);
        num += acc * v[r];
        den += v[r] * v[r];
    }
    num = block_sum_f(num, red);
    den = block_sum_f(den, red);
    if (threadIdx.x == 0) out[(size_t)i * gridDim.x + j] = (num / den) * mx * mx;
}

__global__ __launch_bounds__(256) void hist_inv_total_kernel(const unsigned* __restrict__ h, int nb, double* __restrict__ inv) {
    __shared__ double sh[4];
    const unsigned* p = h + (size_t)blockIdx.x * nb;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) acc += (double)p[i];
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) inv[blockIdx.x] = 1.0 / acc;
}

// mean of k = exp(-lambda / (2 sigma^2)) over an (nx, ny) table of squared distances; symmetric tables hold j > i only.
// out[0] = mean k, out[1] = mean (1 - k) (free of the cancellation against 1)
__global__ __launch_bounds__(1024) void mmd_reduce_kernel(const float* __restrict__ lam, int nx, int ny, int symmetric,
                                                          double inv_two_sigma2, double* __restrict__ out) {
    __shared__ double sh[16];
    double acc = 0.0;
    const long long total = (long long)nx * ny;
    for (long long e = threadIdx.x; e < total; e += 1024) {
        const int i = (int)(e / ny), j = (int)(e % ny);
        if (symmetric && j <= i) continue;
        const double one_minus_k = -expm1(-(double)lam[e] * inv_two_sigma2);
        acc += symmetric ? 2.0 * one_minus_k : one_minus_k;
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) {
        out[1] = acc / (double)total;
        out[0] = 1.0 - out[1];
    }
}

}  // namespace

extern "C" {

int rldm_bev_histogram(const float* points, const int32_t* offsets, int num_samples, int stride, float field_size, int bins,
                       float min_depth, float max_depth, uint32_t* hist, void* stream) {
    RLDM_REQUIRE(points && offsets && hist, "null argument");
    RLDM_REQUIRE(num_samples > 0 && stride >= 3, "bad shape");
    RLDM_REQUIRE(bins > 0 && bins % 2 == 0, "bins must be even (metrics/metrics/histogram/histogram.py:10-14)");
    hipStream_t st = (hipStream_t)stream;
    RLDM_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)num_samples * bins * bins * sizeof(uint32_t), st));
    const double square = (double)field_size / (double)bins;          // python floats: float64
    const double half = ((double)bins / 2.0) * square;
    const double step = (half - (-half)) / (double)bins;              // np.linspace(-half, half, bins + 1)
    bev_histogram_kernel<<<dim3(64, num_samples), 256, 0, st>>>(points, offsets, num_samples, stride, -half, half, step, bins,
                                                                min_depth, max_depth, hist);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int rldm_hist_jsd(const uint32_t* hx, int nx, const uint32_t* hy, int ny, int bins, double* jsd, void* stream) {
    RLDM_REQUIRE(hx && hy && jsd, "null argument");
    RLDM_REQUIRE(nx > 0 && ny > 0 && bins > 0, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int nb = bins * bins;
    unsigned long long* sums = nullptr;                 // [2][nb] column sums, then one double for the result
    RLDM_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&sums), (size_t)(2 * nb + 1) * sizeof(unsigned long long), st));
    double* res = reinterpret_cast<double*>(sums + 2 * nb);
    hist_colsum_kernel<<<(nb + 255) / 256, 256, 0, st>>>(hx, nx, nb, sums);
    hist_colsum_kernel<<<(nb + 255) / 256, 256, 0, st>>>(hy, ny, nb, sums + nb);
    jsd_kernel<<<1, 1024, 0, st>>>(sums, sums + nb, nb, res);
    RLDM_HIP_CHECK(hipGetLastError());
    RLDM_HIP_CHECK(hipMemcpyAsync(jsd, res, sizeof(double), hipMemcpyDeviceToHost, st));
    RLDM_HIP_CHECK(hipStreamSynchronize(st));
    RLDM_HIP_CHECK(hipFreeAsync(sums, st));
    return 0;
}

int rldm_hist_spectral_sq(const uint32_t* hx, int nx, const uint32_t* hy, int ny, int bins, int symmetric, float* lambda,
                          void* stream) {
    RLDM_REQUIRE(hx && hy && lambda, "null argument");
    RLDM_REQUIRE(nx > 0 && ny > 0, "bad shape");
    RLDM_REQUIRE(bins >= 4 && bins <= NBMAX && bins % 4 == 0, "bins must be a multiple of 4, at most 104");
    RLDM_REQUIRE(!symmetric || (hx == hy && nx == ny), "symmetric needs the same histogram set on both sides");
    hipStream_t st = (hipStream_t)stream;
    const int nb = bins * bins;
    double* inv = nullptr;
    RLDM_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&inv), (size_t)(nx + ny) * sizeof(double), st));
    hist_inv_total_kernel<<<nx, 256, 0, st>>>(hx, nb, inv);
    hist_inv_total_kernel<<<ny, 256, 0, st>>>(hy, nb, inv + nx);
    RLDM_HIP_CHECK(hipMemsetAsync(lambda, 0, (size_t)nx * ny * sizeof(float), st));
    constexpr int S = 12;
    auto kern = spectral_sq_kernel<S>;
    const size_t lds = (size_t)3 * NBMAX * PITCH * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        RLDM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    kern<<<dim3(ny, nx), 256, lds, st>>>(hx, hy, inv, inv + nx, bins, symmetric, lambda);
    RLDM_HIP_CHECK(hipGetLastError());
    RLDM_HIP_CHECK(hipFreeAsync(inv, st));
    return 0;
}

int rldm_hist_mmd(const uint32_t* hx, int nx, const uint32_t* hy, int ny, int bins, float sigma, double* out4, void* stream) {
    RLDM_REQUIRE(hx && hy && out4, "null argument");
    RLDM_REQUIRE(sigma > 0.f, "sigma must be positive");
    hipStream_t st = (hipStream_t)stream;
    const size_t nmax = (size_t)std::max(nx, ny) * std::max(nx, ny);
    float* lam = nullptr;
    double* part = nullptr;
    RLDM_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&lam), nmax * sizeof(float), st));
    RLDM_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&part), 6 * sizeof(double), st));
    const double i2s = 1.0 / (2.0 * (double)sigma * (double)sigma);
    int rc = 0;
    rc = rc || rldm_hist_spectral_sq(hx, nx, hx, nx, bins, 1, lam, stream);
    if (!rc) mmd_reduce_kernel<<<1, 1024, 0, st>>>(lam, nx, nx, 1, i2s, part);
    rc = rc || rldm_hist_spectral_sq(hy, ny, hy, ny, bins, 1, lam, stream);
    if (!rc) mmd_reduce_kernel<<<1, 1024, 0, st>>>(lam, ny, ny, 1, i2s, part + 2);
    rc = rc || rldm_hist_spectral_sq(hx, nx, hy, ny, bins, 0, lam, stream);
    if (!rc) mmd_reduce_kernel<<<1, 1024, 0, st>>>(lam, nx, ny, 0, i2s, part + 4);
    double h[6] = {};
    if (!rc) {
        RLDM_HIP_CHECK(hipMemcpyAsync(h, part, sizeof(h), hipMemcpyDeviceToHost, st));
        RLDM_HIP_CHECK(hipStreamSynchronize(st));
    }
    (void)hipFreeAsync(lam, st);
    (void)hipFreeAsync(part, st);
    if (rc) return 1;
    out4[0] = h[0];                                   // s1    = mean k(x, x')
    out4[1] = h[2];                                   // s2    = mean k(y, y')
    out4[2] = h[4];                                   // cross = mean k(x, y)
    out4[3] = 2.0 * h[5] - h[1] - h[3];               // s1 + s2 - 2 cross, from the (1 - k) means: no cancellation against 1
    return 0;
}

}  // extern "C"
