// Per-channel statistics for gfx950: deterministic partial (sum, sum of squares) per (image, pixel-chunk, channel) of a
// channels-last bf16 tensor, in the exact [B][P][C] float2 layout the conv epilogue emits (conv_igemm.hip), for
// tensors that did not come out of a conv launch (external inputs of the kernel-level entry points).  The consuming
// conv's prologue folds them into the GroupNorm affine.
// Reference op: torch.nn.GroupNorm(32, C, eps) in ResnetBlock2D / Attention [3P diffusers] and
// `Normalize` (vae/sgm/modules/diffusionmodules/model.py:59-62).
//
// HBM-bound streaming read: every lane loads 16 B (8 channels) of one pixel; a wave covers floor(64/(C/8)) pixels per
// pass, fully coalesced.  The reduction order is fixed (no atomics) so results are bit-reproducible run to run.
#include "kernels.h"

namespace rldm {

__global__ void __launch_bounds__(256) gn_stats_kernel(const GnStatsParams p) {
    __shared__ float sSum[256 * 8];
    __shared__ float sSq[256 * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = p.C;
    const int nch8 = C >> 3;                 // 16-byte pieces per pixel (<= 64)
    const int ppw = 64 / nch8;               // pixels per wave pass
    const int c8 = lane % nch8, sub = lane / nch8;
    const bool active = sub < ppw;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int per = (p.npix + p.P - 1) / p.P;
    const int px0 = chunk * per;
    const int px1 = min(p.npix, px0 + per);

    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    if (active) {
        const bf16_t* base = p.x + c8 * 8;
        for (int px = px0 + wave * ppw + sub; px < px1; px += 4 * ppw) {
            const uint4 v = *reinterpret_cast<const uint4*>(base + ((size_t)b * p.npix + px) * C);
            const float f[8] = {bf16lo(v.x), bf16hi(v.x), bf16lo(v.y), bf16hi(v.y),
                                bf16lo(v.z), bf16hi(v.z), bf16lo(v.w), bf16hi(v.w)};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s[e] += f[e];
                q[e] += f[e] * f[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sSum[tid * 8 + e] = s[e];
        sSq[tid * 8 + e] = q[e];
    }
    __syncthreads();
    // per-channel totals, fixed order over (wave, sub-pixel lane)
    for (int c = tid; c < C; c += 256) {
        const int cc8 = c >> 3, e = c & 7;
        float a = 0.f, d = 0.f;
        for (int w = 0; w < 4; ++w)
            for (int j = 0; j < ppw; ++j) {
                const int t = w * 64 + j * nch8 + cc8;
                a += sSum[t * 8 + e];
                d += sSq[t * 8 + e];
            }
        p.part[((size_t)b * p.P + chunk) * C + c] = make_float2(a, d);
    }
}

int launch_gn_stats(const GnStatsParams& p, hipStream_t stream) {
    RLDM_REQUIRE(p.C % 8 == 0 && p.C <= 512, "gn_stats: channels must be a multiple of 8, <= 512");
    hipLaunchKernelGGL(gn_stats_kernel, dim3(p.P, p.B), dim3(256), 0, stream, p);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

// Fold of the [B][P][C] partials of a tensor with many pixel tiles per image (the 512x32 / 1024x64 levels of the VAE and of
// the pixel-space UNet: P = 64 / 256) into [B][2][C] -- the per-image total in double precision, stored as its float head
// and tail, so that a consumer that adds its "partials" in double gets the same 48 bits back.  Without it every workgroup
// of the consuming conv re-reads P * Cin * 8 bytes of statistics (131 KB for a 43 KB activation tile at the 64-channel
// level) in a serial chain of P / 16 round trips.  A block = (image, 16 channels): 16 slices of P summed in a fixed order.
__global__ void __launch_bounds__(256) gn_fold_kernel(const GnFoldParams p) {
    __shared__ double sS[16][17], sQ[16][17];
    const int tid = threadIdx.x, cl = tid & 15, sl = tid >> 4;
    const int b = blockIdx.y, c = blockIdx.x * 16 + cl;
    const float2* src = p.part + (size_t)b * p.P * p.C + c;
    double S = 0.0, Q = 0.0;
    if (c < p.C) {
        int q = sl;
        // (round 5) 16 rows per thread in flight -- a 256-tile image is ONE memory round trip instead of four dependent ones (the launch
        // is all latency: 36 of them were 12 % of a RangeDM forward at batch 1); same summation order as before
        for (; q + 240 < p.P; q += 256) {
            float2 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = src[(size_t)(q + 16 * j) * p.C];
#pragma unroll
            for (int j = 0; j < 16; ++j) { S += (double)v[j].x; Q += (double)v[j].y; }
        }
        for (; q + 48 < p.P; q += 64) {
            float2 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = src[(size_t)(q + 16 * j) * p.C];
#pragma unroll
            for (int j = 0; j < 4; ++j) { S += (double)v[j].x; Q += (double)v[j].y; }
        }
        for (; q < p.P; q += 16) {
            const float2 v = src[(size_t)q * p.C];
            S += (double)v.x;
            Q += (double)v.y;
        }
    }
    sS[sl][cl] = S;
    sQ[sl][cl] = Q;
    __syncthreads();
    if (sl == 0 && c < p.C) {
        S = 0.0; Q = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) { S += sS[j][cl]; Q += sQ[j][cl]; }
        const float Sh = (float)S, Qh = (float)Q;
        float2* dst = p.out + (size_t)b * 2 * p.C + c;
        dst[0] = make_float2(Sh, Qh);
        dst[p.C] = make_float2((float)(S - (double)Sh), (float)(Q - (double)Qh));
    }
}

int launch_gn_fold(const GnFoldParams& p, hipStream_t stream) {
    RLDM_REQUIRE(p.P > 0 && p.C > 0, "gn_fold: empty statistics");
    hipLaunchKernelGGL(gn_fold_kernel, dim3((p.C + 15) / 16, p.B), dim3(256), 0, stream, p);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

// GroupNorm (+SiLU) of cat[x0, x1] from the producers' per-channel partials -> one activated bf16 tensor.
// A block owns GSL consecutive groups (1/8 of the channels) of `ppb` pixels of one image: it folds just those channels'
// partials into the affine (a few hundred floats, double accumulation as in the conv prologue of conv_igemm.hip) and then
// streams its slice: 16 bytes per lane in, 16 bytes out.
__global__ void __launch_bounds__(256) gn_apply_kernel(const GnApplyParams p, int ppb) {
    __shared__ double sD[2 * 128];
    __shared__ __attribute__((aligned(16))) float sGa[128];
    __shared__ __attribute__((aligned(16))) float sGs[128];
    const int tid = threadIdx.x;
    const int Cin = p.C0 + p.C1;
    const int b = blockIdx.z;
    const int cpg = Cin / p.groups;
    const int gsl = (p.groups + gridDim.y - 1) / gridDim.y;        // groups per block
    const int g0 = blockIdx.y * gsl;
    const int ng = min(gsl, p.groups - g0);
    const int c0 = g0 * cpg, nc = ng * cpg;                         // channel slice (<= 128, multiple of 8)
    const bf16_t* gx0 = p.x0;               // (locals: selecting between fields of `p` by address would copy it to scratch)
    const bf16_t* gx1 = p.x1;
    const float2* gs0 = p.st0;
    const float2* gs1 = p.st1;
    const int nC0 = p.C0, nC1 = p.C1, nP0 = p.P0, nP1 = p.P1;

    // everything the block reads is requested up front (one memory round trip): its channels' statistics, gamma / beta,
    // and the first pieces of its slice
    double S = 0.0, SS = 0.0;
    float gam = 0.f, bet = 0.f;
    const int n8 = nc >> 3;                  // 16-byte pieces per pixel of the slice
    const int px0 = blockIdx.x * ppb;
    const int total = min(ppb, p.npix - px0) * n8;
    constexpr int NB = 4;
    uint4 v[NB];
    int cl[NB];
    bf16_t* dst[NB];
    bf16_t* const gy0 = p.y;
    bf16_t* const gy1 = p.y1;
    const int ysplit = p.ysplit;
    auto load_batch = [&](int q0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int q = q0 + j * 256;
            const int pl = q / n8;
            cl[j] = (q - pl * n8) * 8;
            const int c = c0 + cl[j];
            const size_t pix = (size_t)b * p.npix + px0 + pl;
            dst[j] = ysplit == 0 ? gy0 + pix * Cin + c
                                 : (c < ysplit ? gy0 + pix * ysplit + c : gy1 + pix * (Cin - ysplit) + (c - ysplit));
            v[j] = make_uint4(0u, 0u, 0u, 0u);
            if (q < total) {
                const bool first = c < nC0;
                v[j] = *reinterpret_cast<const uint4*>(first ? gx0 + pix * nC0 + c : gx1 + pix * nC1 + (c - nC0));
            }
        }
    };
    if (tid < nc) {
        const int t = c0 + tid;
        const bool first = t < nC0;
        const int c = first ? t : t - nC0;
        const int C = first ? nC0 : nC1;
        const int P = first ? nP0 : nP1;
        const float2* src = (first ? gs0 : gs1) + (size_t)b * P * C + c;
        int q = 0;
        for (; q + 4 <= P; q += 4) {
            float2 u[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j] = src[(size_t)(q + j) * C];
#pragma unroll
            for (int j = 0; j < 4; ++j) { S += (double)u[j].x; SS += (double)u[j].y; }
        }
        for (; q < P; ++q) {
            const float2 u = src[(size_t)q * C];
            S += (double)u.x;
            SS += (double)u.y;
        }
        gam = p.gamma[t];
        bet = p.beta[t];
    }
    load_batch(tid);
    if (tid < nc) {
        sD[tid] = S;
        sD[128 + tid] = SS;
    }
    __syncthreads();
    if (tid < nc) {                          // every channel's thread folds its own group (no serial phase)
        const int gb = (tid / cpg) * cpg;
        double GS = 0.0, GSS = 0.0;
        for (int i = 0; i < cpg; ++i) {
            GS += sD[gb + i];
            GSS += sD[128 + gb + i];
        }
        const double inv_n = (double)p.inv_n;
        const double mean = GS * inv_n;
        double var = GSS * inv_n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const float a = gam * __builtin_amdgcn_rsqf((float)var + p.eps);
        sGa[tid] = a;
        sGs[tid] = bet - (float)mean * a;
    }
    __syncthreads();
    for (int q0 = tid; q0 < total; q0 += 256 * NB) {
        if (q0 != tid) load_batch(q0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (q0 + j * 256 >= total) continue;
            const float4 a0 = *reinterpret_cast<const float4*>(sGa + cl[j]), a1 = *reinterpret_cast<const float4*>(sGa + cl[j] + 4);
            const float4 s0 = *reinterpret_cast<const float4*>(sGs + cl[j]), s1 = *reinterpret_cast<const float4*>(sGs + cl[j] + 4);
            float f0 = bf16lo(v[j].x) * a0.x + s0.x, f1 = bf16hi(v[j].x) * a0.y + s0.y;
            float f2 = bf16lo(v[j].y) * a0.z + s0.z, f3 = bf16hi(v[j].y) * a0.w + s0.w;
            float f4 = bf16lo(v[j].z) * a1.x + s1.x, f5 = bf16hi(v[j].z) * a1.y + s1.y;
            float f6 = bf16lo(v[j].w) * a1.z + s1.z, f7 = bf16hi(v[j].w) * a1.w + s1.w;
            if (p.silu) {
                silu_x8(f0, f1, f2, f3, f4, f5, f6, f7);
            }
            uint4 o;
            o.x = pack_bf16x2(f0, f1); o.y = pack_bf16x2(f2, f3);
            o.z = pack_bf16x2(f4, f5); o.w = pack_bf16x2(f6, f7);
            *reinterpret_cast<uint4*>(dst[j]) = o;
        }
    }
}

int launch_gn_apply(const GnApplyParams& p, hipStream_t stream) {
    const int Cin = p.C0 + p.C1;
    RLDM_REQUIRE(Cin % p.groups == 0 && p.C0 % 8 == 0 && p.C1 % 8 == 0, "gn_apply: unsupported channels");
    RLDM_REQUIRE(p.ysplit == 0 || (p.ysplit % 8 == 0 && p.ysplit < Cin && p.y1 != nullptr), "gn_apply: bad output split");
    const int cpg = Cin / p.groups;
    // groups per block: ~1/8 of them, widened until the slice is whole 16-byte pieces (<= 128 channels)
    int gsl = (p.groups + 7) / 8;
    while ((gsl * cpg) % 8 != 0 && gsl < p.groups) ++gsl;
    RLDM_REQUIRE((gsl * cpg) % 8 == 0 && gsl * cpg <= 128 && gsl <= 32, "gn_apply: channel slices must be whole 16-byte pieces");
    const int ny = (p.groups + gsl - 1) / gsl;
    const int n8 = gsl * cpg / 8;
    int ppb = std::max(1, 1024 / n8);                           // ~4 pieces per thread
    while (ppb > 16 && (long long)p.B * ny * ((p.npix + ppb - 1) / ppb) < 256) ppb >>= 1;
    GnApplyParams q = p;
    q.inv_n = (float)(1.0 / ((double)p.npix * cpg));
    hipLaunchKernelGGL(gn_apply_kernel, dim3((p.npix + ppb - 1) / ppb, ny, p.B), dim3(256), 0, stream, q, ppb);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace rldm
