// lidar.hip -- range image <-> point cloud on gfx950: the step right after the sampler in
// every reference driver (ldm/inference.py:171-183) and the one right before the training path (ldm/dataset.py:159-226).
//
// Everything here is HBM / atomic bound, fp32, one thread per range-image pixel or per LiDAR return:
//   range_to_points_kernel   point_cloud_to_range_image.to_pc_torch          ldm/dataset.py:228-278
//   bev_splat_kernel         to_voxel + _splat_points_to_volumes (votes)     ldm/dataset.py:280-288, 13-124
//   bev_finalize_kernel      feature / clamp(density), log(density + 1)      ldm/dataset.py:126-130, 289-293
//   filter_count / filter_scatter   `pc[norm(pc[:, :3]) < 90]` in order      ldm/inference.py:177-179
//   render_u8_kernel         `(x.permute(2,1,0).clip(0,1)*255).astype(u8)`   ldm/inference.py:180-183
//   project_* kernels        point_cloud_to_range_image.__call__ + process_miss_value + normalize
//                                                                            ldm/dataset.py:159-226
// The arithmetic keeps the reference's fp32 operation order (no FMA contraction) so cell indices and the ordered
// compaction agree with the torch/numpy result wherever the inputs do.
#include "common.h"
#include "../../include/rangeldm_hip.h"

#include <cmath>
#include <map>
#include <vector>

#pragma clang fp contract(off)      // and -ffp-contract=off for this file in the Makefile (covers the header inlines)

namespace {

// One IEEE operation each, correctly rounded: plain operators under -ffp-contract=off, and sqrtf / `/` under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt.  (HIP's __fsqrt_rn / __fdiv_rn intrinsics are the ~1 ulp native
// instructions: measured 12 % of ranges off by one ulp against numpy.)
__device__ inline float f_mul(float a, float b) { return a * b; }
__device__ inline float f_add(float a, float b) { return a + b; }
__device__ inline float f_sub(float a, float b) { return a - b; }
__device__ inline float f_div(float a, float b) { return a / b; }
__device__ inline float f_sqrt(float a) { return sqrtf(a); }

struct LidarDev {
    const float* cos_incl;
    const float* sin_incl;
    const float* incl;
    const float* height;
    int H;
    int mode;          // 0 linear, 1 log, 2 inverse
    float mean, std, range_fill, intensity_fill;
};

__device__ inline float decode_range(float v, const LidarDev& L) {
    float r;
    if (L.mode == 1) r = f_sub(exp2f(f_mul(v, 6.f)), 1.f);
    else if (L.mode == 2) r = f_div(1.f, fmaxf(v, 0.0001f));
    else r = f_add(f_mul(v, L.std), L.mean);
    return r < 0.f ? L.range_fill : r;
}

// pixel (w, h) with decoded range r -> sensor-frame xyz (ldm/dataset.py:251-272)
__device__ inline void pixel_to_xyz(float r, int w, int h, const LidarDev& L, const float* cos_azi, const float* sin_azi,
                                    float* x, float* y, float* z) {
    const float xy = f_mul(r, L.cos_incl[h]);
    *z = f_sub(L.height[h], f_mul(r, L.sin_incl[h]));
    *x = f_mul(xy, cos_azi[w]);
    *y = f_mul(xy, sin_azi[w]);
}

// ---- to_pc_torch ------------------------------------------------------------------------------------------------
// img (B, C, W, H): flat pixel index n = w * H + h is also the point index of the reference's reshape(B, -1).
__global__ __launch_bounds__(256) void range_to_points_kernel(const float* __restrict__ img, int C, int W, int H,
                                                              LidarDev L, const float* __restrict__ cos_azi,
                                                              const float* __restrict__ sin_azi, float* __restrict__ pts) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    const int N = W * H;
    if (n >= N) return;
    const float* im = img + (size_t)b * C * N;
    const float r = decode_range(im[n], L);
    float x, y, z;
    pixel_to_xyz(r, n / H, n % H, L, cos_azi, sin_azi, &x, &y, &z);
    if (C > 1) {
        reinterpret_cast<float4*>(pts)[(size_t)b * N + n] = make_float4(x, y, z, im[N + n]);
    } else {
        float* p = pts + ((size_t)b * N + n) * 3;
        p[0] = x; p[1] = y; p[2] = z;
    }
}

// ---- to_voxel ---------------------------------------------------------------------------------------------------
struct GridDev {
    int gx, gy, gz;            // grid_sizes = [gz, gy, gx]
    float cx, cy, cz;          // (hi + lo) / 2
    float hx, hy, hz;          // (hi - lo) / 2
};

__device__ inline float grid_coord(float v, float c, float h, int g) {
    const float p = f_div(f_sub(v, c), h);
    return f_mul(f_mul(f_add(p, 1.f), 0.5f), (float)(g - 1));
}

// One thread per pixel: decode, project, cast its 8 trilinear votes with hardware fp32 atomics straight into the
// (B, 2*gz, gy, gx) output (density planes first, then feature planes).  Out-of-volume votes carry weight 0 in the
// reference (added to a random voxel): they are dropped.
__global__ __launch_bounds__(256) void bev_splat_kernel(const float* __restrict__ img, int C, int W, int H, LidarDev L,
                                                        const float* __restrict__ cos_azi, const float* __restrict__ sin_azi,
                                                        GridDev G, float* __restrict__ vox) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    const int N = W * H;
    if (n >= N) return;
    const float* im = img + (size_t)b * C * N;
    const float r = decode_range(im[n], L);
    const float f = im[N + n];
    float x, y, z;
    pixel_to_xyz(r, n / H, n % H, L, cos_azi, sin_azi, &x, &y, &z);
    const float px = grid_coord(x, G.cx, G.hx, G.gx), py = grid_coord(y, G.cy, G.hy, G.gy), pz = grid_coord(z, G.cz, G.hz, G.gz);
    // anything that cannot touch the volume (also NaN) is dropped before the float -> int conversion
    if (!(px > -1.f && px < (float)G.gx && py > -1.f && py < (float)G.gy && pz > -1.f && pz < (float)G.gz)) return;
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const int X = (int)fx, Y = (int)fy, Z = (int)fz;
    const float rx = f_sub(px, fx), ry = f_sub(py, fy), rz = f_sub(pz, fz);
    const size_t nvox = (size_t)G.gz * G.gy * G.gx;
    float* dens = vox + (size_t)b * 2 * nvox;
    float* feat = dens + nvox;
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
        const int X_ = X + dx;
        const float wx = dx ? rx : f_sub(1.f, rx);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int Y_ = Y + dy;
            const float wy = dy ? ry : f_sub(1.f, ry);
#pragma unroll
            for (int dz = 0; dz < 2; ++dz) {
                const int Z_ = Z + dz;
                const float wz = dz ? rz : f_sub(1.f, rz);
                const float w = f_mul(f_mul(wx, wy), wz);
                const bool ok = X_ >= 0 && X_ < G.gx && Y_ >= 0 && Y_ < G.gy && Z_ >= 0 && Z_ < G.gz;
                if (ok && w != 0.f) {
                    const size_t idx = ((size_t)Z_ * G.gy + Y_) * G.gx + X_;
                    unsafeAtomicAdd(dens + idx, w);
                    unsafeAtomicAdd(feat + idx, f_mul(w, f));
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void bev_finalize_kernel(float* __restrict__ vox, size_t nvox, int log_density,
                                                           float min_weight) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= nvox) return;
    float* dens = vox + (size_t)b * 2 * nvox;
    float* feat = dens + nvox;
    const float d = dens[i];
    if (d == 0.f) return;               // no votes: feature 0 / min_weight = 0 and log(0 + 1) = 0 are already there
    feat[i] = f_div(feat[i], fmaxf(d, min_weight));
    if (log_density) dens[i] = logf(f_add(d, 1.f));
}

// ---- ordered depth filter ---------------------------------------------------------------------------------------
// pass 1: kept points per 256-point chunk; pass 2: chunk base = sum of the preceding counts, wave ballots give the
// rank inside the chunk.  The kept points keep their order (what `pc[mask]` does), no atomics.
__device__ inline bool keep_point(const float* __restrict__ p, float max_depth) {
    const float d2 = f_add(f_add(f_mul(p[0], p[0]), f_mul(p[1], p[1])), f_mul(p[2], p[2]));
    return f_sqrt(d2) < max_depth;
}

__global__ __launch_bounds__(256) void filter_count_kernel(const float* __restrict__ pts, int N, int cols, float max_depth,
                                                           int* __restrict__ chunk_counts) {
    __shared__ int wave_cnt[4];
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    const bool keep = n < N && keep_point(pts + ((size_t)b * N + n) * cols, max_depth);
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) chunk_counts[(size_t)b * gridDim.x + blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ __launch_bounds__(256) void filter_scatter_kernel(const float* __restrict__ pts, int N, int cols, float max_depth,
                                                             const int* __restrict__ chunk_counts, float* __restrict__ out,
                                                             int* __restrict__ counts) {
    __shared__ int red[4];
    __shared__ int wave_cnt[4];
    const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int* cc = chunk_counts + (size_t)b * nchunks;
    int part = 0;
    for (int i = threadIdx.x; i < chunk; i += 256) part += cc[i];
    for (int o = 32; o; o >>= 1) part += __shfl_xor(part, o);
    const int n = chunk * 256 + threadIdx.x;
    const float* p = pts + ((size_t)b * N + n) * cols;
    const bool keep = n < N && keep_point(p, max_depth);
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[wv] = part; wave_cnt[wv] = __popcll(m); }
    __syncthreads();
    int base = red[0] + red[1] + red[2] + red[3];
    for (int i = 0; i < wv; ++i) base += wave_cnt[i];
    if (keep) {
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        float* q = out + ((size_t)b * N + pos) * cols;
        if (cols == 4) *reinterpret_cast<float4*>(q) = *reinterpret_cast<const float4*>(p);
        else { q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; }
    }
    if (chunk == nchunks - 1 && threadIdx.x == 0)
        counts[b] = red[0] + red[1] + red[2] + red[3] + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// ---- 8-bit rendering --------------------------------------------------------------------------------------------
// src (B, C, W, H) fp32, channel c -> dst [B][H][W] bytes: a 64 x 64 tile is read along H and written along W.
__global__ __launch_bounds__(256) void render_u8_kernel(const float* __restrict__ src, int C, int W, int H, int c,
                                                        unsigned char* __restrict__ dst) {
    __shared__ unsigned char tile[64][65];
    const int b = blockIdx.z, w0 = blockIdx.x * 64, h0 = blockIdx.y * 64;
    const float* s = src + ((size_t)b * C + c) * W * H;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int w = w0 + i, h = h0 + tx;
        if (w < W && h < H) {
            const float v = fminf(fmaxf(s[(size_t)w * H + h], 0.f), 1.f);
            tile[i][tx] = (unsigned char)(int)f_mul(v, 255.f);
        }
    }
    __syncthreads();
    unsigned char* d = dst + (size_t)b * W * H;
    for (int i = ty; i < 64; i += 4) {
        const int h = h0 + i, w = w0 + tx;
        if (w < W && h < H) d[(size_t)h * W + w] = tile[tx][i];
    }
}

// ---- projection (dataset side) ----------------------------------------------------------------------------------
// A LiDAR return competes for its pixel with a 64-bit key (range bits << 32 | point index): the reference writes the
// points sorted farthest-first so the NEAREST return of a pixel survives (ldm/dataset.py:173-185); atomicMin on the key
// yields the same winner without a sort (positive floats order like their bit patterns; equal ranges: the highest point
// index wins, which is what a stable farthest-first sort followed by in-order writes gives).
__global__ __launch_bounds__(256) void project_keys_kernel(const float* __restrict__ pts, int n_pts, int stride,
                                                           const int* __restrict__ rows, LidarDev L, int W, float min_depth,
                                                           unsigned long long* __restrict__ keys) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pts) return;
    const float* p = pts + (size_t)i * stride;
    const float x = p[0], y = p[1], z = p[2];
    if (min_depth > 0.f) {      // nuScenes: drop returns closer than 2 m (ldm/nuscenes_range_image.py:37-41)
        const float d = f_sqrt(f_add(f_add(f_mul(x, x), f_mul(y, y)), f_mul(z, z)));
        if (!(d > min_depth)) return;
    }
    int row;
    if (rows) {
        row = rows[i];
    } else {                    // ldm/kitti360_range_image.py:51-61: beam with the closest inclination
        const float xy = f_sqrt(f_add(f_mul(x, x), f_mul(y, y)));
        float best = INFINITY;
        row = 0;
        for (int h = 0; h < L.H; ++h) {
            const float e = fabsf(f_sub(L.incl[h], atan2f(f_sub(L.height[h], z), xy)));
            if (e < best) { best = e; row = h; }
        }
    }
    if (row < 0 || row >= L.H) return;
    // column (ldm/dataset.py:162-166): fp32 throughout (python scalars do not promote a float32 array), round half to even
    const float turn = f_div(f_add(atan2f(y, x), (float)M_PI), (float)(2.0 * M_PI));
    const float colf = f_sub((float)((double)W - 1.0 + 0.5), f_mul(turn, (float)W));
    int col = (int)rintf(colf);
    if (col == W) col = W - 1;
    if (col < 0) col = 0;
    const float zz = f_sub(z, L.height[row]);
    float rng = f_sqrt(f_add(f_add(f_mul(x, x), f_mul(y, y)), f_mul(zz, zz)));
    if (rng > L.range_fill) rng = L.range_fill;
    const unsigned long long key = ((unsigned long long)__float_as_uint(rng) << 32) | (unsigned int)~i;
    atomicMin(keys + (size_t)row * W + col, key);
}

// keys -> (H, W, 2) raw range image with -1 where no return landed
__global__ __launch_bounds__(256) void project_gather_kernel(const unsigned long long* __restrict__ keys,
                                                             const float* __restrict__ pts, int stride, LidarDev L, int W,
                                                             float2* __restrict__ raw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L.H * W) return;
    const unsigned long long k = keys[i];
    float2 v = make_float2(-1.f, -1.f);
    if (k != ~0ull) {
        const float rng = __uint_as_float((unsigned int)(k >> 32));
        float val = rng;
        if (L.mode == 1) val = f_div(log2f(f_add(rng, 1.f)), 6.f);
        else if (L.mode == 2) val = f_div(1.f, rng);
        v = make_float2(val, pts[(size_t)(~(unsigned int)k) * stride + 3]);
    }
    raw[i] = v;
}

// process_miss_value + normalize + the (2, 1, 0) permute: raw (H, W, 2) -> image (2, W, H), mask (W, H), car (W, H)
__global__ __launch_bounds__(256) void project_finish_kernel(const float2* __restrict__ raw, LidarDev L, int W,
                                                             float fill0, float fill1, float* __restrict__ img,
                                                             unsigned char* __restrict__ mask, unsigned char* __restrict__ car) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int H = L.H;
    if (i >= H * W) return;
    const int w = i / H, h = i % H;         // output-major indexing: writes are contiguous along H
    auto filled = [&](int hh, int ww) -> float2 {      // after fill_noise: a missing pixel takes its right neighbour
        float2 v = raw[(size_t)hh * W + ww];
        if (v.x == -1.f) v = raw[(size_t)hh * W + (ww + 1 == W ? 0 : ww + 1)];
        return v;
    };
    float2 v = filled(h, w);
    // mask = (range > 0) of the pixel itself, or of the right neighbour where the pixel was missing
    const bool m = v.x > 0.f;
    const bool still = v.x == -1.f;
    bool cw = false;
    if (still) {
        const int hd = (h - 2 + H) % H, hu = (h + 2) % H, wr = (w - 2 + W) % W, wl = (w + 2) % W;
        cw = filled(hd, w).x != -1.f || filled(hu, w).x != -1.f || filled(h, wr).x != -1.f || filled(h, wl).x != -1.f;
        v = make_float2(fill0, fill1);
    }
    if (L.mode == 0) v.x = f_div(f_sub(v.x, L.mean), L.std);
    img[(size_t)w * H + h] = v.x;
    img[(size_t)W * H + (size_t)w * H + h] = v.y;
    mask[(size_t)w * H + h] = m;
    car[(size_t)w * H + h] = cw;
}

}  // namespace

// ---- handle -----------------------------------------------------------------------------------------------------
struct rldm_lidar {
    rldm_lidar_config cfg;
    float* tables = nullptr;                  // [4][H]: cos(incl), sin(incl), incl, height
    std::map<int, float*> azimuth;            // W -> [2][W]: cos(azi), sin(azi)
    int* chunk_counts = nullptr;
    size_t chunk_cap = 0;
    unsigned long long* keys = nullptr;       // projection scratch [H][W] + raw (H, W, 2)
    float2* raw = nullptr;
    size_t proj_cap = 0;
    LidarDev dev() const {
        LidarDev L;
        L.cos_incl = tables;
        L.sin_incl = tables + cfg.beams;
        L.incl = tables + 2 * cfg.beams;
        L.height = tables + 3 * cfg.beams;
        L.H = cfg.beams;
        L.mode = cfg.mode;
        L.mean = cfg.mean;
        L.std = cfg.std;
        L.range_fill = cfg.range_fill;
        L.intensity_fill = cfg.intensity_fill;
        return L;
    }
};

static int lidar_azimuth(rldm_lidar* l, int W, const float** cos_azi, const float** sin_azi) {
    auto it = l->azimuth.find(W);
    if (it == l->azimuth.end()) {
        // ldm/dataset.py:266-267 in fp32 steps like the torch expression, then cos / sin of that fp32 angle
        std::vector<float> t(2 * (size_t)W);
        const float pi_f = (float)M_PI;
        for (int w = 0; w < W; ++w) {
            volatile float a = ((float)W - 0.5f) - (float)w;
            a = a / (float)W;
            a = a * 2.f;
            a = a * pi_f;
            a = a - pi_f;
            t[w] = (float)cos((double)a);
            t[W + w] = (float)sin((double)a);
        }
        float* d = nullptr;
        RLDM_HIP_CHECK(hipMalloc(&d, t.size() * sizeof(float)));
        RLDM_HIP_CHECK(hipMemcpy(d, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
        it = l->azimuth.emplace(W, d).first;
    }
    *cos_azi = it->second;
    *sin_azi = it->second + W;
    return 0;
}

extern "C" {

int rldm_lidar_create(const rldm_lidar_config* cfg, const float* incl, const float* height, rldm_lidar** out) {
    RLDM_REQUIRE(cfg && incl && height && out, "null argument");
    RLDM_REQUIRE(cfg->beams > 0 && cfg->beams <= 4096, "beams out of range");
    RLDM_REQUIRE(cfg->mode >= 0 && cfg->mode <= 2, "mode must be 0 (linear), 1 (log) or 2 (inverse)");
    RLDM_REQUIRE(cfg->grid[0] > 0 && cfg->grid[1] > 0 && cfg->grid[2] > 0, "grid_sizes must be positive");
    auto* l = new rldm_lidar();
    l->cfg = *cfg;
    const int H = cfg->beams;
    std::vector<float> t(4 * (size_t)H);
    for (int h = 0; h < H; ++h) {
        t[h] = (float)cos((double)incl[h]);
        t[H + h] = (float)sin((double)incl[h]);
        t[2 * H + h] = incl[h];
        t[3 * H + h] = height[h];
    }
    if (hipMalloc(&l->tables, t.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(l->tables, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        rldm::set_error("rldm_lidar_create: device allocation failed");
        delete l;
        return 1;
    }
    *out = l;
    return 0;
}

void rldm_lidar_destroy(rldm_lidar* l) {
    if (!l) return;
    if (l->tables) (void)hipFree(l->tables);
    for (auto& kv : l->azimuth) (void)hipFree(kv.second);
    if (l->chunk_counts) (void)hipFree(l->chunk_counts);
    if (l->keys) (void)hipFree(l->keys);
    if (l->raw) (void)hipFree(l->raw);
    delete l;
}

int rldm_lidar_to_points(rldm_lidar* l, const float* range_images, int B, int C, int W, float* points, void* stream) {
    RLDM_REQUIRE(l && range_images && points, "null argument");
    RLDM_REQUIRE(B > 0 && C >= 1 && W > 0, "bad shape");
    const float *ca, *sa;
    if (lidar_azimuth(l, W, &ca, &sa)) return 1;
    const int N = W * l->cfg.beams;
    range_to_points_kernel<<<dim3((N + 255) / 256, B), 256, 0, (hipStream_t)stream>>>(range_images, C, W, l->cfg.beams,
                                                                                    l->dev(), ca, sa, points);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int rldm_lidar_to_voxel(rldm_lidar* l, const float* range_images, int B, int C, int W, float* voxel, void* stream) {
    RLDM_REQUIRE(l && range_images && voxel, "null argument");
    RLDM_REQUIRE(B > 0 && W > 0, "bad shape");
    RLDM_REQUIRE(C >= 2, "to_voxel needs the remission channel (ldm/dataset.py:286 splats pc[:, :, 3:])");
    const float *ca, *sa;
    if (lidar_azimuth(l, W, &ca, &sa)) return 1;
    const rldm_lidar_config& c = l->cfg;
    GridDev G;
    G.gz = c.grid[0]; G.gy = c.grid[1]; G.gx = c.grid[2];
    // fp32 like the torch expressions (ldm/dataset.py:283-284)
    G.cx = (c.pc_range[3] + c.pc_range[0]) / 2.f; G.hx = (c.pc_range[3] - c.pc_range[0]) / 2.f;
    G.cy = (c.pc_range[4] + c.pc_range[1]) / 2.f; G.hy = (c.pc_range[4] - c.pc_range[1]) / 2.f;
    G.cz = (c.pc_range[5] + c.pc_range[2]) / 2.f; G.hz = (c.pc_range[5] - c.pc_range[2]) / 2.f;
    const size_t nvox = (size_t)G.gz * G.gy * G.gx;
    hipStream_t st = (hipStream_t)stream;
    RLDM_HIP_CHECK(hipMemsetAsync(voxel, 0, (size_t)B * 2 * nvox * sizeof(float), st));
    const int N = W * c.beams;
    bev_splat_kernel<<<dim3((N + 255) / 256, B), 256, 0, st>>>(range_images, C, W, c.beams, l->dev(), ca, sa, G, voxel);
    RLDM_HIP_CHECK(hipGetLastError());
    bev_finalize_kernel<<<dim3((unsigned)((nvox + 255) / 256), B), 256, 0, st>>>(voxel, nvox, c.normalize_volume_densities, 1e-4f);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int rldm_lidar_filter_points(rldm_lidar* l, const float* points, int B, int N, int cols, float max_depth, float* out,
                             int32_t* counts, void* stream) {
    RLDM_REQUIRE(l && points && out && counts, "null argument");
    RLDM_REQUIRE(B > 0 && N > 0 && (cols == 3 || cols == 4), "bad shape (cols must be 3 or 4)");
    const int nchunks = (N + 255) / 256;
    const size_t need = (size_t)B * nchunks;
    if (need > l->chunk_cap) {
        if (l->chunk_counts) RLDM_HIP_CHECK(hipFree(l->chunk_counts));
        l->chunk_counts = nullptr;
        RLDM_HIP_CHECK(hipMalloc(&l->chunk_counts, need * sizeof(int)));
        l->chunk_cap = need;
    }
    hipStream_t st = (hipStream_t)stream;
    filter_count_kernel<<<dim3(nchunks, B), 256, 0, st>>>(points, N, cols, max_depth, l->chunk_counts);
    RLDM_HIP_CHECK(hipGetLastError());
    filter_scatter_kernel<<<dim3(nchunks, B), 256, 0, st>>>(points, N, cols, max_depth, l->chunk_counts, out, counts);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int rldm_render_u8(const float* src, int B, int C, int W, int H, int channel, uint8_t* dst, void* stream) {
    RLDM_REQUIRE(src && dst, "null argument");
    RLDM_REQUIRE(B > 0 && C > 0 && W > 0 && H > 0 && channel >= 0 && channel < C, "bad shape");
    render_u8_kernel<<<dim3((W + 63) / 64, (H + 63) / 64, B), 256, 0, (hipStream_t)stream>>>(src, C, W, H, channel, dst);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int rldm_lidar_project(rldm_lidar* l, const float* points, int n_points, int stride, const int32_t* rows, float min_depth,
                       float* image, uint8_t* mask, uint8_t* car_window_mask, void* stream) {
    RLDM_REQUIRE(l && (points || n_points == 0) && image && mask && car_window_mask, "null argument");
    RLDM_REQUIRE(n_points >= 0 && stride >= 4, "points must carry at least x, y, z, intensity");
    const rldm_lidar_config& c = l->cfg;
    const int W = c.width, H = c.beams;
    RLDM_REQUIRE(W > 0, "config.width must be positive");
    const size_t px = (size_t)W * H;
    if (px > l->proj_cap) {
        if (l->keys) RLDM_HIP_CHECK(hipFree(l->keys));
        if (l->raw) RLDM_HIP_CHECK(hipFree(l->raw));
        l->keys = nullptr; l->raw = nullptr;
        RLDM_HIP_CHECK(hipMalloc(&l->keys, px * sizeof(unsigned long long)));
        RLDM_HIP_CHECK(hipMalloc(&l->raw, px * sizeof(float2)));
        l->proj_cap = px;
    }
    hipStream_t st = (hipStream_t)stream;
    RLDM_HIP_CHECK(hipMemsetAsync(l->keys, 0xff, px * sizeof(unsigned long long), st));
    const LidarDev L = l->dev();
    if (n_points > 0) {
        project_keys_kernel<<<(n_points + 255) / 256, 256, 0, st>>>(points, n_points, stride, rows, L, W, min_depth, l->keys);
        RLDM_HIP_CHECK(hipGetLastError());
    }
    project_gather_kernel<<<(unsigned)((px + 255) / 256), 256, 0, st>>>(l->keys, points, stride, L, W, l->raw);
    RLDM_HIP_CHECK(hipGetLastError());
    // ldm/dataset.py:212-217: what a still-missing pixel is filled with
    float fill0 = c.range_fill, fill1 = c.intensity_fill;
    if (c.mode == 1) { fill0 = (float)(log2((double)c.range_fill + 1.0) / 6.0); fill1 = (float)(log2((double)c.intensity_fill + 1.0) / 6.0); }
    else if (c.mode == 2) fill0 = (float)(1.0 / (double)c.range_fill);
    project_finish_kernel<<<(unsigned)((px + 255) / 256), 256, 0, st>>>(l->raw, L, W, fill0, fill1, image, mask, car_window_mask);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"
