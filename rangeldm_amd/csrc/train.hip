// train.hip -- kernels of the UNet training step (ldm/train_unconditional.py:466-558) for gfx950.
//
// First correct version: self-contained and deliberately simple.  Activations and gradients are fp32 channels-last
// [B][W][H][C] (W = azimuth wraps, H = beams zero-padded); GEMM operands are rounded to bf16 on their way into
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (what `mixed_precision: bf16` autocast does to conv / linear), everything
// else is fp32.  Master weights stay in the torch layout [N][Cin][3][3] (fp32, flat buffer shared with AdamW / EMA); the
// convolutions read bf16 copies [N][tap][Cin] (forward) and [Cin][8 - tap][N] (data gradient) refreshed after every
// optimizer step.
//
//   tr_conv_kernel        conv 3x3 / 1x1 (circular W, zero H; stride 1 | 2; nearest-x2 or zero-insertion of the input)
//                         + bias + per-sample row (time embedding) + residual.  Forward, data gradient (flipped /
//                         transposed weights; stride-2 convs through zero insertion) and every Linear (W = H = 1).
//   tr_wgrad_kernel       dW[n][c][tap] = sum_p dy[p][n] * x[src(p, tap)][c]: pixels are the contraction index, split over
//                         workgroups and waves, fp32 atomics into the zeroed gradient.
//   tr_colsum_kernel      bias / time-embedding-row gradients: per-image column sums of dy.
//   tr_gn_*               GroupNorm(32) statistics, forward (+ SiLU), backward (reduce + apply, d gamma / d beta).
//   tr_attn_*             head_dim 8 softmax attention, forward (+ log-sum-exp) and backward (dq ; dk, dv): one thread per
//                         query / key against the head's K, V (or Q, dO) staged in LDS; no atomics, deterministic.
//   elementwise           add, channel copy (concat / split), 2x2 sum (nearest-x2 backward), SiLU, sinusoidal timestep
//                         embedding, input packing, MSE loss + gradient, sum of squares, AdamW (+ clip scale + EMA),
//                         weight repacking.
#include "common.h"
#include "../../include/rangeldm_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace rldm {
// train_attn.hip: the same three passes on the matrix cores (bf16 operands, fp32 statistics / accumulation)
int tr_attention_forward_mfma(const float* q, const float* k, const float* v, int ld, int B, int L, int C, float* o, float* lse,
                              hipStream_t st);
int tr_attention_backward_mfma(const float* q, const float* k, const float* v, int ld, const float* o, const float* dO,
                               const float* lse, int B, int L, int C, float* delta, float* dq, float* dk, float* dv, hipStream_t st);
}  // namespace rldm

namespace {

inline bool gn_vec_reduce_env() {
    static const bool v = getenv("RLDM_TR_GN_VEC_REDUCE") != nullptr;
    return v;
}

// RLDM_TR_ATTN=scalar: the fp32 one-thread-per-query kernels of this file (A/B runs)
inline bool attention_scalar() {
    static const bool v = getenv("RLDM_TR_ATTN") && std::string(getenv("RLDM_TR_ATTN")) == "scalar";
    return v;
}

// dw[n][c][t] += sum over slices of part[t][slice][n][c], four channels per thread (Cin % 4 == 0), `nb` workgroups of `nt` threads
// striding over the (tap, channel quad) items: the body of tr_wgrad_reduce_vec_kernel, also run as a rider of the next conv launch.
__device__ inline void tr_wgrad_reduce_items(const float* __restrict__ part, int slices, int N, int Cin, int taps, float* __restrict__ dw,
                                             int block, int nb, int nt) {
    const size_t nc = (size_t)N * Cin, nc4 = nc >> 2, items = nc4 * taps;
    for (size_t i = (size_t)block * nt + threadIdx.x; i < items; i += (size_t)nb * nt) {
        const int t = (int)(i / nc4);
        const size_t e = (i - (size_t)t * nc4) * 4;
        const float* src = part + (size_t)t * slices * nc + e;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int sidx = 0;
        for (; sidx + 8 <= slices; sidx += 8) {
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(src + (size_t)(sidx + j) * nc);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
        for (; sidx < slices; ++sidx) acc += *reinterpret_cast<const f32x4*>(src + (size_t)sidx * nc);
#pragma unroll
        for (int q = 0; q < 4; ++q) dw[(e + q) * taps + t] += acc[q];
    }
}

// source pixel of output pixel (b, wo, ho) under tap (dw, dh): index into the input's pixel array, or -1 for a zero.
// mode 0: plain; 1: nearest x2 (virtual input is twice as large); 2: zero insertion (virtual odd coordinates are zeros)
__device__ inline int src_pixel(int b, int wo, int ho, int dw, int dh, int stride, int mode, int Win, int Hin) {
    const int sh = mode ? 1 : 0;
    const int Wv = Win << sh, Hv = Hin << sh;
    int vw = wo * stride + dw, vh = ho * stride + dh;
    if (vh < 0 || vh >= Hv) return -1;
    vw = vw < 0 ? vw + Wv : (vw >= Wv ? vw - Wv : vw);
    if (mode == 2 && ((vw | vh) & 1)) return -1;
    return (b * Win + (vw >> sh)) * Hin + (vh >> sh);
}

// 8 consecutive fp32 -> bf16x8; `valid` elements exist (the rest are zeros); vec: the row is 16-byte aligned
__device__ inline bf16x8 load_bf16x8_from_f32(const float* p, int valid, bool vec) {
    float f[8];
    if (vec && valid >= 8) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = e < valid ? p[e] : 0.f;
    }
    uint4 u;
    u.x = rldm::pack_bf16x2(f[0], f[1]); u.y = rldm::pack_bf16x2(f[2], f[3]);
    u.z = rldm::pack_bf16x2(f[4], f[5]); u.w = rldm::pack_bf16x2(f[6], f[7]);
    return __builtin_bit_cast(bf16x8, u);
}

// ---- convolution / linear (forward and data gradient) ---------------------------------------------------------------
struct TrConv {
    const float* x; const bf16_t* w; const float* bias; const float* rowadd; const float* res; float* y;
    int B, Win, Hin, Cin, Cin_pad, Wout, Hout, N, taps, stride, mode, rowadd_ld, accumulate;
};

// (round 5) what the fused launches fold into a conv / weight-gradient kernel (include/rangeldm_hip.h: rldm_train_fuse).  A tensor's
// GroupNorm statistics travel as per-(image, channel) (sum, sum of squares) pairs "cs" [B][C][2], accumulated by the epilogue of the
// conv that produced the tensor; a consumer derives (mean, rstd) of its groups from them (a concatenated input = two sources with their
// own pairs: groups may straddle the seam).  The GroupNorm backward sums travel the same way: "gs" [B][C][2] = (sum dz, sum dz xhat).
struct TrFuse {
    const float* x1; int C0;                       // second source of the input: channels [C0, Cin) (null: one source)
    const float* cs0; const float* cs1;            // input = act(GroupNorm(cat(x, x1))) built while staging (cs0 null: plain input)
    const float* gamma; const float* beta; int silu, groups; float eps;
    float* cs_out;                                 // += (sum, sumsq) of y per (image, channel)
    const float* g0; const float* g1; int G0;      // data-gradient epilogue: y = d act(GroupNorm(cat(g0, g1))) -> dz = y act'(z) stored,
    const float* gcs0; const float* gcs1;          //   gs_out += (sum dz, sum dz xhat)
    const float* ggamma; const float* gbeta; int gsilu, ggroups; float geps;
    float* gs_out;
    unsigned* tickets;                             // split-K launches: a zeroed arrival counter per output tile (left zeroed)
    // a rider: the reduction of the PREVIOUS weight-gradient launch's partial tiles (it does not depend on this conv, and as a launch of
    // its own it cost ~10 us 47 times per step): run by extra z-planes of this launch's grid (rz0 = the first of them; rblocks = 0: none)
    const float* rpart; float* rdw; int rslices, rN, rCin, rtaps, rz0, rblocks;
};

// (v_exp_f32 + v_rcp_f32: the IEEE division of 1.f / x costs ten more instructions per element of every staged tile)
#ifndef RLDM_TR_LAST_ACQUIRE
#define RLDM_TR_LAST_ACQUIRE 1  /* the last arriver of a fused split-K tile acquires (agent scope) before it re-reads the tile */
#endif
#ifndef RLDM_TR_ABL
#define RLDM_TR_ABL 0          /* timing experiments (wrong results): 1 no statistics atomics, 2 no sigmoid in the staging transforms, 4 no split-K output atomics, 8 plain stores instead of them; weight gradient: 16 no LDS stash, 32 no MFMA loop, 64 no global fetch of the next chunk */
#endif
#if RLDM_TR_ABL & 2
__device__ inline float tr_sigmoid(float z) { return z; }
#else
__device__ inline float tr_sigmoid(float z) { return __builtin_amdgcn_rcpf(1.f + __expf(-z)); }
#endif

// y = x * a + b = GroupNorm(x) for the channels [c_lo, c_hi) of image `img` of a (possibly two-source) tensor with npix pixels per
// image: sc[c] = (a, b).  Every thread sums its own channel's group from global memory (<= 24 pairs, L2 hits): no barrier inside, the
// caller synchronises before sc is read.
__device__ inline void tr_gn_coeffs(float2* sc, int c_lo, int c_hi, const float* cs0, const float* cs1, int C0, int C, int img, int groups,
                                    float eps, int npix, const float* gamma, const float* beta) {
    const int C1 = C - C0, cpg = C / groups;
    for (int ch = c_lo + (int)threadIdx.x; ch < c_hi; ch += blockDim.x) {
        const int g = ch / cpg;
        double s = 0.0, ss = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            const float2 v = c < C0 ? reinterpret_cast<const float2*>(cs0)[(size_t)img * C0 + c] : reinterpret_cast<const float2*>(cs1)[(size_t)img * C1 + c - C0];
            s += (double)v.x; ss += (double)v.y;
        }
        const double n = (double)npix * cpg, mean = s / n;
        double var = ss / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const float a = rsqrtf((float)var + eps) * gamma[ch];
        sc[ch] = make_float2(a, beta[ch] - (float)mean * a);
    }
}

// the same for the BN channels [n0, n0 + BN) only, with (mean, rstd) kept: ce[cl] = (a, b, mean, rstd); every thread sums its own group
// from global memory (<= 24 pairs, L2 hits).  No barrier inside: the caller synchronises before ce is read.
__device__ inline void tr_gn_coeffs_tile(float4* ce, int BN, int n0, const float* cs0, const float* cs1, int C0, int C, int img, int groups,
                                         float eps, int npix, const float* gamma, const float* beta) {
    const int C1 = C - C0, cpg = C / groups;
    for (int cl = threadIdx.x; cl < BN; cl += blockDim.x) {
        const int ch = n0 + cl;
        if (ch >= C) { ce[cl] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
        const int g = ch / cpg;
        double s = 0.0, ss = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            const float2 v = c < C0 ? reinterpret_cast<const float2*>(cs0)[(size_t)img * C0 + c] : reinterpret_cast<const float2*>(cs1)[(size_t)img * C1 + c - C0];
            s += (double)v.x; ss += (double)v.y;
        }
        const double n = (double)npix * cpg, mean = s / n;
        double var = ss / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const float rstd = rsqrtf((float)var + eps), a = rstd * gamma[ch];
        ce[cl] = make_float4(a, beta[ch] - (float)mean * a, (float)mean, rstd);
    }
}

// Epilogue of a fused conv over the finished fp32 tile in LDS (tile[pl * (BN + 1) + cl], PT pixels of ONE image x BN channels):
// thread (channel cl = tid % BN, pixel group tid / BN) walks its pixels: + bias / row / residual (add_terms), the GroupNorm-backward
// transform where asked, the store (coalesced along the channels), and the per-channel sums, which reach cs_out / gs_out as one atomic
// per (workgroup, channel, component).  colacc: [2 * BN] floats of LDS.
template <int BN, int PT, int NT>
__device__ inline void tr_tile_epilogue(const float* tile, float* colacc, const float4* ce, const TrConv& p, const TrFuse& f, int px0,
                                        int n0, int img, bool add_terms, bool store_y) {
    // thread (channel quad cq = tid % (BN / 4), pixel group tid / (BN / 4)): 16-byte global accesses, all of a thread's loads in flight
    // at once (the 4-byte form of this pass -- a channel per thread, 32 pixels in two batches -- was ~5 us of every fused launch)
    constexpr int NQ = BN / 4, NPG = NT / NQ, IT = PT / NPG;        // BN = 128, 256 threads: 32 quads x 8 groups, 8 pixels per thread
    static_assert(PT % NPG == 0 && IT <= 16, "tile epilogue shape");
    const int tid = threadIdx.x, cq = tid % NQ, pg = tid / NQ, cl = 4 * cq, ch = n0 + cl, N = p.N;
    for (int e = tid; e < 2 * BN; e += NT) colacc[e] = 0.f;
    __syncthreads();
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (ch < N) {                                                   // (N % 4 == 0: a quad is inside or outside)
        f32x4 addc = {0.f, 0.f, 0.f, 0.f};
        if (add_terms) {
            if (p.bias) addc += *reinterpret_cast<const f32x4*>(p.bias + ch);
            if (p.rowadd) addc += *reinterpret_cast<const f32x4*>(p.rowadd + (size_t)img * p.rowadd_ld + ch);
        }
        const bool gn = f.gs_out != nullptr, act = gn && f.gsilu != 0;
        float4 c4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) c4[q] = gn ? ce[cl + q] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* __restrict__ gsrc = nullptr;
        int gld = 0;
        if (gn) {                                                   // (G0 % 4 == 0: the quad lies in one source)
            if (ch < f.G0) { gsrc = f.g0 + ch; gld = f.G0; }
            else { gsrc = f.g1 + (ch - f.G0); gld = N - f.G0; }
        }
        const float* __restrict__ rsrc = (add_terms && p.res) ? p.res + ch : nullptr;
        float* __restrict__ ydst = p.y + ch;
        f32x4 rv[IT], gv[IT];
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            const size_t gp = (size_t)px0 + pg + u * NPG;
            rv[u] = rsrc ? *reinterpret_cast<const f32x4*>(rsrc + gp * N) : f32x4{0.f, 0.f, 0.f, 0.f};
            gv[u] = gn ? *reinterpret_cast<const f32x4*>(gsrc + gp * gld) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            const int pl = pg + u * NPG;
            const size_t gp = (size_t)px0 + pl;
            const float* tr_ = tile + pl * (BN + 1) + cl;
            f32x4 v = {tr_[0], tr_[1], tr_[2], tr_[3]};
            v += addc + rv[u];
            if (gn) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float g = gv[u][q], xh = (g - c4[q].z) * c4[q].w;
                    if (act) {
                        const float z = g * c4[q].x + c4[q].y, sg = tr_sigmoid(z);
                        v[q] *= sg * (1.f + z * (1.f - sg));
                    }
                    s1[q] += v[q];
                    s2[q] += v[q] * xh;
                }
                *reinterpret_cast<f32x4*>(ydst + gp * N) = v;
            } else {
                s1 += v;
                s2 += v * v;
                if (store_y) *reinterpret_cast<f32x4*>(ydst + gp * N) = v;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        atomicAdd(&colacc[2 * (cl + q)], s1[q]);
        atomicAdd(&colacc[2 * (cl + q) + 1], s2[q]);
    }
    __syncthreads();
    float* out = f.gs_out ? f.gs_out : f.cs_out;
    for (int e = tid; e < 2 * BN; e += NT) {
        const int c2 = n0 + (e >> 1);
#if !(RLDM_TR_ABL & 1)
        if (c2 < N) unsafeAtomicAdd(out + ((size_t)img * N + c2) * 2 + (e & 1), colacc[e]);
#endif
    }
}

// D[channel][pixel]: lane (pixel l & 31, half l >> 5) holds channels (r & 3) + 8 (r >> 2) + 4 half of its pixel.
// A wave owns 32 pixels x 64 channels (one pixel fragment feeds two MFMAs); a workgroup 64 pixels x 128 channels.
template <bool FAST>
__global__ __launch_bounds__(256) void tr_conv_kernel(const TrConv p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kg = lane >> 5;
    const int P = p.B * p.Wout * p.Hout;
    const int px = blockIdx.x * 64 + (wave & 1) * 32 + l31;
    const int n0 = blockIdx.y * 128 + (wave >> 1) * 64;
    if (n0 >= p.N) return;
    const bool pxok = px < P;
    const int pc = pxok ? px : 0;
    const int ho = pc % p.Hout, t1 = pc / p.Hout, wo = t1 % p.Wout, b = t1 / p.Wout;
    const int r0 = n0 + l31, r1 = r0 + 32;                            // this lane's weight rows (A operands)
    const bool ok0 = r0 < p.N, ok1 = r1 < p.N;
    const size_t rstride = (size_t)p.taps * p.Cin_pad;
    const bf16_t* w0 = p.w + (size_t)(ok0 ? r0 : 0) * rstride + 8 * kg;
    const bf16_t* w1 = p.w + (size_t)(ok1 ? r1 : 0) * rstride + 8 * kg;
    const bool vec = (p.Cin & 3) == 0;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    for (int t = 0; t < p.taps; ++t) {
        const int dw = p.taps == 9 ? t / 3 - 1 : 0, dh = p.taps == 9 ? t % 3 - 1 : 0;
        const int sp = pxok ? src_pixel(b, wo, ho, dw, dh, p.stride, p.mode, p.Win, p.Hin) : -1;
        const float* xrow = p.x + (size_t)(sp < 0 ? 0 : sp) * p.Cin + 8 * kg;
        const size_t toff = (size_t)t * p.Cin_pad;
        if (FAST) {
            // Cin % 16 == 0: branch-free body, four k-steps of loads in flight (rows past N are computed on row 0's
            // weights and never stored; a zero-padded tap multiplies by 0)
            const float live = sp < 0 ? 0.f : 1.f;
            const bf16_t* wa = w0 + toff;
            const bf16_t* wb = w1 + toff;
#pragma unroll 4
            for (int c0 = 0; c0 < p.Cin_pad; c0 += 16) {
                const uint4 a0 = *reinterpret_cast<const uint4*>(wa + c0), a1 = *reinterpret_cast<const uint4*>(wb + c0);
                const float4 x0 = *reinterpret_cast<const float4*>(xrow + c0), x1 = *reinterpret_cast<const float4*>(xrow + c0 + 4);
                uint4 u;
                u.x = rldm::pack_bf16x2(x0.x * live, x0.y * live); u.y = rldm::pack_bf16x2(x0.z * live, x0.w * live);
                u.z = rldm::pack_bf16x2(x1.x * live, x1.y * live); u.w = rldm::pack_bf16x2(x1.z * live, x1.w * live);
                const bf16x8 bv = __builtin_bit_cast(bf16x8, u);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), bv, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), bv, acc1, 0, 0, 0);
            }
        } else {
            for (int c0 = 0; c0 < p.Cin_pad; c0 += 16) {
                uint4 a0 = make_uint4(0u, 0u, 0u, 0u), a1 = a0;
                if (ok0) a0 = *reinterpret_cast<const uint4*>(w0 + toff + c0);
                if (ok1) a1 = *reinterpret_cast<const uint4*>(w1 + toff + c0);
                const int valid = sp < 0 ? 0 : p.Cin - (c0 + 8 * kg);
                const bf16x8 bv = load_bf16x8_from_f32(xrow + c0, valid, vec);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), bv, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), bv, acc1, 0, 0, 0);
            }
        }
    }
    if (!pxok) return;
    float* yrow = p.y + (size_t)px * p.N;
    const float* rrow = p.res ? p.res + (size_t)px * p.N : nullptr;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = n0 + 32 * h + (r & 3) + 8 * (r >> 2) + 4 * kg;
            if (ch >= p.N) continue;
            float v = h ? acc1[r] : acc0[r];
            if (p.bias) v += p.bias[ch];
            if (p.rowadd) v += p.rowadd[(size_t)b * p.rowadd_ld + ch];
            if (rrow) v += rrow[ch];
            if (p.accumulate) v += yrow[ch];
            yrow[ch] = v;
        }
}

// LDS-staged variant for Cin % CK == 0 (CK = 32 | 64 channels per stage): the workgroup (64 pixels x 128 channels) stages the
// pixel tile [64][CK] (fp32 -> bf16 on the way) and the weight tile [128][CK] with coalesced pieces per thread.  A stage
// costs one memory latency (~1.5 us measured: activations of a level live in the Infinity Cache, not in L2) against 0.1 us
// of MFMAs, so the global loads of the next D stages are kept in flight in registers (static ring, loop unrolled by D).
// Row pitch CK * 2 + 16 bytes: an odd number of 16-byte slots, conflict-free ds_read_b128.
// (round 5) FU: the fused form (TrFuse) -- GroupNorm (+ SiLU) of a one- or two-source input while staging, and the tile epilogue
// (tr_tile_epilogue: output statistics or the GroupNorm-backward transform + sums), in split-K launches run by each tile's last arriver.
// (round 5) D: stages of global loads in flight.  D = 3 (the first three stages requested at entry, a slot refilled as soon as it has
// been copied to LDS) was measured on the low-resolution launches (2 - 9 stages per workgroup) and changes nothing: a 4-stage 1x1 conv of
// 16 workgroups takes 9.6 us against 9.4 us, the 3x3 convs of the 64x4 level 19.0 against 21.6 us, the step 798 against 796 samples/s --
// those launches are bound by launch + first-touch + drain latency, not by the per-stage round trip.  Not instantiated.
template <int CK, int BN, bool FU = false, int D = 1>
__global__ __launch_bounds__(256) void tr_conv_lds_kernel(const TrConv p, const TrFuse f) {
    static_assert(BN == 64 || BN == 128, "channel tile");
    constexpr int TPR = 256 / BN, NR = BN / 64;     // threads per weight row of a stage; 32-channel MFMA rows per wave
    constexpr int PITCH = CK + 8;                   // bf16 elements
    constexpr int XV = CK / 32;                     // float4 pairs per thread for x (4 threads per pixel row, CK / 4 channels each)
    constexpr int WV = CK / (8 * TPR);              // uint4 per thread for w
    // one buffer: the two stage tiles during the K loop, the fp32 output tile [64][BN + 1] of the split-K epilogue afterwards
    constexpr int STAGE_BYTES = (64 + BN) * PITCH * 2, TILE_BYTES = 64 * (BN + 1) * 4;
    __shared__ __attribute__((aligned(16))) unsigned char raw[STAGE_BYTES > TILE_BYTES ? STAGE_BYTES : TILE_BYTES];
    bf16_t* const sX = reinterpret_cast<bf16_t*>(raw);
    bf16_t* const sW = sX + 64 * PITCH;
    __shared__ float2 sSc[FU ? 768 : 1];                            // (FU) input GroupNorm: (a, b) per input channel
    __shared__ float4 sCe[FU ? BN : 1];                             // (FU) epilogue GroupNorm: (a, b, mean, rstd) per tile channel
    __shared__ float sCol[FU ? 2 * BN : 1];
    __shared__ int sLast;
    // (native vector types: arrays of HIP's uint4 / float4 structs are not split into registers and went through scratch;
    //  no lambda may capture the by-value argument `p` by reference either: that copies the struct to scratch)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int ksplit = f.rblocks ? f.rz0 : (int)gridDim.z;          // (z-planes from rz0 on carry the rider)
    if ((int)blockIdx.z >= ksplit) {
        const int rb = ((int)blockIdx.z - ksplit) * (int)(gridDim.x * gridDim.y) + (int)(blockIdx.y * gridDim.x + blockIdx.x);
        if (rb < f.rblocks) tr_wgrad_reduce_items(f.rpart, f.rslices, f.rN, f.rCin, f.rtaps, f.rdw, rb, f.rblocks, 256);
        return;
    }
    const int Cin = p.Cin, Cin_pad = p.Cin_pad, taps = p.taps, Wout = p.Wout, Hout = p.Hout, N = p.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kg = lane >> 5;
    const int P = p.B * Wout * Hout;
    const int px0 = blockIdx.x * 64, n0 = blockIdx.y * BN;
    const int img = FU ? px0 / (Wout * Hout) : 0;                   // (FU: a tile lies inside one image)
    const bool in_gn = FU && f.cs0 != nullptr;
    const int fC0 = FU ? f.C0 : Cin;
    // staging roles
    const int spx = tid >> 2, spart = tid & 3;      // x: pixel row of the tile, quarter of the CK channels
    const int swr = tid / TPR, shalf = tid % TPR;   // w: weight row of the tile, 1 / TPR of the CK channels
    const int gpx = px0 + spx;
    const bool gpx_ok = gpx < P;
    const int pc = gpx_ok ? gpx : 0;
    const int sho = pc % Hout, st1 = pc / Hout, swo = st1 % Wout, sb = st1 / Wout;
    const int wrow = n0 + swr;
    const bf16_t* wptr = p.w + (size_t)(wrow < N ? wrow : 0) * taps * Cin_pad + shalf * (CK / TPR);     // walks [tap][chunk]
    const float* const xbase = p.x + spart * (CK / 4);
    const int nck = Cin / CK, niter = taps * nck;
    const int wskip = Cin_pad - Cin;                // weight rows are padded to 16 channels per tap
    // stage walker: (tap, chunk) advance without divisions; the tap's source pixel is looked up when the tap changes
    int tap = 0, cc = 0;
    const float* xs = xbase;
    float live = 0.f;
    int sp_cur = 0, scc = 0;                        // (FU) source pixel of the current tap; chunk of the data in the staging registers
    const float* const x1p = FU ? f.x1 : nullptr;
    auto new_tap = [&]() __attribute__((always_inline)) {
        const int dw = taps == 9 ? tap / 3 - 1 : 0, dh = taps == 9 ? tap % 3 - 1 : 0;
        const int sp = gpx_ok ? src_pixel(sb, swo, sho, dw, dh, p.stride, p.mode, p.Win, p.Hin) : -1;
        live = sp < 0 ? 0.f : 1.f;
        sp_cur = sp < 0 ? 0 : sp;
        xs = xbase + (size_t)sp_cur * Cin;
    };
    // split K: workgroup z of gridDim.z contracts stages [it0, it1) and adds its partial tile to y atomically (y zeroed by the
    // launcher; z == 0 carries bias / row / residual).  The low-resolution levels have 32 - 128 output tiles for 36 - 72
    // serial stages of ~0.7 us each: latency, not work, was their whole cost.
    const int it0 = (int)((long long)niter * blockIdx.z / ksplit), it1 = (int)((long long)niter * (blockIdx.z + 1) / ksplit);
    tap = it0 / nck;
    cc = it0 - tap * nck;
    wptr += (size_t)tap * Cin_pad + cc * CK;
    new_tap();
    xs += cc * CK;
    f32x4 xr[D][XV][2];                             // (ring slots: every index below is a compile-time constant after unrolling)
    u32x4 wr[D][WV];
    float lvs[D];
    int sccs[D];
    auto fetch = [&](const int slot) __attribute__((always_inline)) {
        lvs[slot] = live;
        const float* src = xs;
        if constexpr (FU) {                         // two sources: the chunk lies in one of them (C0 % CK == 0)
            const int c = cc * CK + spart * (CK / 4);
            src = c < fC0 ? p.x + (size_t)sp_cur * fC0 + c : x1p + (size_t)sp_cur * (Cin - fC0) + (c - fC0);
            sccs[slot] = cc;
        }
#pragma unroll
        for (int q = 0; q < XV; ++q) {
            xr[slot][q][0] = reinterpret_cast<const f32x4*>(src)[2 * q];
            xr[slot][q][1] = reinterpret_cast<const f32x4*>(src)[2 * q + 1];
        }
#pragma unroll
        for (int q = 0; q < WV; ++q) wr[slot][q] = reinterpret_cast<const u32x4*>(wptr)[q];
        xs += CK;
        wptr += CK;
        if (++cc == nck) {
            cc = 0;
            ++tap;
            wptr += wskip;
            new_tap();
        }
    };
    auto stash = [&](const int slot) __attribute__((always_inline)) {
        const float lv = lvs[slot];
#pragma unroll
        for (int q = 0; q < XV; ++q) {
            f32x4 a = xr[slot][q][0], b = xr[slot][q][1];
            if constexpr (FU) {
                if (in_gn) {                        // GroupNorm (+ SiLU) of the staged values (zero padding applies to the result: lv below)
                    const float2* co = sSc + sccs[slot] * CK + spart * (CK / 4) + 8 * q;
                    const bool act = f.silu != 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float2 c0_ = co[e], c1_ = co[4 + e];
                        float z0 = a[e] * c0_.x + c0_.y, z1 = b[e] * c1_.x + c1_.y;
                        if (act) { z0 *= tr_sigmoid(z0); z1 *= tr_sigmoid(z1); }
                        a[e] = z0; b[e] = z1;
                    }
                }
            }
            u32x4 u;
            u.x = rldm::pack_bf16x2(a.x * lv, a.y * lv); u.y = rldm::pack_bf16x2(a.z * lv, a.w * lv);
            u.z = rldm::pack_bf16x2(b.x * lv, b.y * lv); u.w = rldm::pack_bf16x2(b.z * lv, b.w * lv);
            *reinterpret_cast<u32x4*>(sX + spx * PITCH + spart * (CK / 4) + 8 * q) = u;
        }
#pragma unroll
        for (int q = 0; q < WV; ++q) *reinterpret_cast<u32x4*>(sW + swr * PITCH + shalf * (CK / TPR) + 8 * q) = wr[slot][q];
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const bf16_t* bx = sX + ((wave & 1) * 32 + l31) * PITCH + 8 * kg;
    const bf16_t* aw0 = sW + ((wave >> 1) * (BN / 2) + l31) * PITCH + 8 * kg;
    const bf16_t* aw1 = aw0 + (NR == 2 ? 32 : 0) * PITCH;
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (it0 + d < it1) fetch(d);
    if constexpr (FU) {
        // (behind the first stage's loads; a K split only needs the chunks of its own stages; visible after the loop's first barrier)
        if (in_gn) {
            const bool all = it1 - it0 >= nck;
            const int c_lo = all ? 0 : (it0 % nck) * CK, c_hi = all ? Cin : c_lo + (it1 - it0) * CK;
            tr_gn_coeffs(sSc, c_lo, min(c_hi, Cin), f.cs0, f.cs1, fC0, Cin, img, f.groups, f.eps, p.Win * p.Hin, f.gamma, f.beta);
            if (c_hi > Cin) tr_gn_coeffs(sSc, 0, c_hi - Cin, f.cs0, f.cs1, fC0, Cin, img, f.groups, f.eps, p.Win * p.Hin, f.gamma, f.beta);
        }
        if (f.gs_out && ksplit == 1)
            tr_gn_coeffs_tile(sCe, BN, n0, f.gcs0, f.gcs1, f.G0, N, img, f.ggroups, f.geps, Wout * Hout, f.ggamma, f.gbeta);
    }
    for (int it = it0; it < it1; it += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (D > 1 && it + d >= it1) break;
            __syncthreads();                        // everyone is done reading the previous stage
            stash(d);
            __syncthreads();
            if (it + d + D < it1) fetch(d);         // the slot just copied out is requested again: in flight during D stages of MFMAs
#pragma unroll
            for (int ks = 0; ks < CK / 16; ++ks) {
                const bf16x8 bv = *reinterpret_cast<const bf16x8*>(bx + 16 * ks);
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(aw0 + 16 * ks);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bv, acc0, 0, 0, 0);
                if (NR == 2) {
                    const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(aw1 + 16 * ks);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bv, acc1, 0, 0, 0);
                }
            }
        }
    }
    if (ksplit > 1) {
        // through LDS so that the atomics run along the channels (one cache line per 32 lanes; lane = pixel would touch 64 lines
        // per instruction: measured 4x slower than not splitting at all)
        float* tile = reinterpret_cast<float*>(raw);
        __syncthreads();
        {
            float* trow = tile + ((wave & 1) * 32 + l31) * (BN + 1) + (wave >> 1) * (BN / 2);
#pragma unroll
            for (int h = 0; h < NR; ++h)
#pragma unroll
                for (int r = 0; r < 16; ++r) trow[32 * h + 8 * (r >> 2) + 4 * kg + (r & 3)] = h ? acc1[r] : acc0[r];
        }
        __syncthreads();
        const bool first = blockIdx.z == 0;
        const int HW = p.Wout * p.Hout;
        for (int e = tid; e < 64 * BN; e += 256) {
            const int pl = e / BN, cl = e % BN;
            const int gp = px0 + pl, ch = n0 + cl;
            if (gp >= P || ch >= p.N) continue;
            float u = tile[pl * (BN + 1) + cl];
            if (first) {
                if (p.bias) u += p.bias[ch];
                if (p.rowadd) u += p.rowadd[(size_t)(gp / HW) * p.rowadd_ld + ch];
                if (p.res) u += p.res[(size_t)gp * p.N + ch];
            }
#if RLDM_TR_ABL & 4
            if (u == 12345.678f) p.y[(size_t)gp * p.N + ch] = u;      // (timing experiment: no split-K atomics)
#elif RLDM_TR_ABL & 8
            p.y[(size_t)gp * p.N + ch] = u;                          // (timing experiment: plain stores instead of atomics)
#else
            unsafeAtomicAdd(p.y + (size_t)gp * p.N + ch, u);
#endif
        }
        if constexpr (FU) {
            if (f.cs_out || f.gs_out) {
                // the tile's last arriver runs the epilogue over the finished sums.  The partial tiles travel as device-scope atomics --
                // performed at the coherence point, never dirty in an L2 -- so "published" = every wave has its atomics acknowledged
                // (s_waitcnt vmcnt(0)) before one lane draws the ticket: no write-back fence (cdna_hip_programming.md prices it at 1.7 -
                // 6.5 us per workgroup).  The last arriver added to every line of the tile itself (an atomic drops the line from its L2)
                // and never read one (nothing in its L1): it re-reads the tile with L1-bypassing loads, no invalidate.
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    unsigned* tk = f.tickets + blockIdx.y * gridDim.x + blockIdx.x;
                    const unsigned t = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    const int last = (int)t == ksplit - 1;
                    if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sLast = last;
                }
                __syncthreads();
                if (!sLast) return;
#if RLDM_TR_LAST_ACQUIRE
                // (round 6) the argument above rests on what an atomic does to a line of the issuing XCD's L2, which nothing documents:
                // the ONE workgroup per tile that re-reads the sums pairs the release on the ticket with an agent-scope acquire (an
                // invalidate of what its caches may hold of the tile) before it does.  -DRLDM_TR_LAST_ACQUIRE=0: the round-5 form.
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
                if (f.gs_out) tr_gn_coeffs_tile(sCe, BN, n0, f.gcs0, f.gcs1, f.G0, N, img, f.ggroups, f.geps, Wout * Hout, f.ggamma, f.gbeta);
                if ((N & 3) == 0) {
                    for (int e = tid; e < 64 * (BN / 4); e += 256) {
                        const int pl = e / (BN / 4), cl = (e % (BN / 4)) * 4;
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        if (n0 + cl < N) v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.y + (size_t)(px0 + pl) * N + n0 + cl));
#pragma unroll
                        for (int q = 0; q < 4; ++q) tile[pl * (BN + 1) + cl + q] = v[q];
                    }
                } else {
                    for (int e = tid; e < 64 * BN; e += 256) {
                        const int pl = e / BN, cl = e % BN;
                        const int ch = n0 + cl;
                        tile[pl * (BN + 1) + cl] = ch < N ? __hip_atomic_load(p.y + (size_t)(px0 + pl) * N + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
                    }
                }
                __syncthreads();
                tr_tile_epilogue<BN, 64, 256>(tile, sCol, sCe, p, f, px0, n0, img, false, false);
            }
        }
        return;
    }
    if constexpr (FU) {
        if (f.cs_out || f.gs_out) {
            float* tile = reinterpret_cast<float*>(raw);
            __syncthreads();
            float* trow = tile + ((wave & 1) * 32 + l31) * (BN + 1) + (wave >> 1) * (BN / 2);
#pragma unroll
            for (int h = 0; h < NR; ++h)
#pragma unroll
                for (int r = 0; r < 16; ++r) trow[32 * h + 8 * (r >> 2) + 4 * kg + (r & 3)] = h ? acc1[r] : acc0[r];
            __syncthreads();
            tr_tile_epilogue<BN, 64, 256>(tile, sCol, sCe, p, f, px0, n0, img, true, true);
            return;
        }
    }
    const int px = px0 + (wave & 1) * 32 + l31;
    if (px >= P) return;
    const int b = px / (p.Wout * p.Hout);
    const int nb = n0 + (wave >> 1) * (BN / 2);
    float* yrow = p.y + (size_t)px * p.N;
    const float* rrow = p.res ? p.res + (size_t)px * p.N : nullptr;
#pragma unroll
    for (int h = 0; h < NR; ++h)
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
            const int ch = nb + 32 * h + 8 * (r >> 2) + 4 * kg;       // 4 consecutive channels per register quad
            if (ch >= p.N) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = h ? acc1[r + e] : acc0[r + e];
            if (ch + 3 < p.N && (p.N & 3) == 0) {
                if (p.bias) { const float4 t = *reinterpret_cast<const float4*>(p.bias + ch); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
                if (p.rowadd) { const float4 t = *reinterpret_cast<const float4*>(p.rowadd + (size_t)b * p.rowadd_ld + ch); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
                if (rrow) { const float4 t = *reinterpret_cast<const float4*>(rrow + ch); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
                if (p.accumulate) { const float4 t = *reinterpret_cast<const float4*>(yrow + ch); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
                *reinterpret_cast<float4*>(yrow + ch) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (ch + e >= p.N) continue;
                    float u = v[e];
                    if (p.bias) u += p.bias[ch + e];
                    if (p.rowadd) u += p.rowadd[(size_t)b * p.rowadd_ld + ch + e];
                    if (rrow) u += rrow[ch + e];
                    if (p.accumulate) u += yrow[ch + e];
                    yrow[ch + e] = u;
                }
            }
        }
}

// Halo variant for the stride-1, mode-0 3x3 convs whose launch fills the chip (levels 0 / 1: forward and data gradient): the
// kernel above re-stages its 64-pixel tile for every tap (9 x 16 KB of fp32 per 64-channel chunk and workgroup -- at level 0
// 302 MB through L2 in 26 us, which is what bounds it).  Here a 64-channel chunk's tile is staged ONCE with its halo --
// (64 / H + 2) azimuth columns x (H + 2) beams, circular in azimuth, zero rows above and below -- and the nine taps read
// their B fragments from it at a uniform row offset dw * (H + 2) + dh; only the weight tile is staged per tap.  x traffic
// and fp32->bf16 conversions drop 5x.  Stage order: chunk-major, tap fastest.
template <int BN, int PT, bool FU = false>
__global__ __launch_bounds__(PT * 4) void tr_conv_halo_kernel(const TrConv p, const TrFuse f) {
    static_assert(BN == 64 || BN == 128, "channel tile");
    constexpr int CK = 64;
    static_assert(PT == 64 || PT == 128, "pixel tile");
    constexpr int NT = PT * 4;                      // threads: PT / 32 pixel groups x 2 channel halves of waves
    constexpr int TPR = NT / BN, NR = BN / 64;
    constexpr int PITCH = CK + 8;
    constexpr int WV = CK / (8 * TPR);
    constexpr int MAXHP = PT == 64 ? 136 : 264;     // (PT / H + 2) * (H + 2) for H = 2 .. 32
    // (FU: one buffer, the fp32 output tile [PT][BN + 1] of the fused epilogue afterwards)
    constexpr int STAGE_BYTES = (MAXHP + BN) * PITCH * 2, TILE_BYTES = FU ? PT * (BN + 1) * 4 : 0;
    __shared__ __attribute__((aligned(16))) unsigned char hraw[STAGE_BYTES > TILE_BYTES ? STAGE_BYTES : TILE_BYTES];
    bf16_t* const sXh = reinterpret_cast<bf16_t*>(hraw);
    bf16_t* const sW = sXh + MAXHP * PITCH;
    __shared__ float2 sSc[FU ? 768 : 1];
    __shared__ float4 sCe[FU ? BN : 1];
    __shared__ float sCol[FU ? 2 * BN : 1];
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    if (f.rblocks && (int)blockIdx.z >= f.rz0) {                    // the rider's z-planes
        const int rb = ((int)blockIdx.z - f.rz0) * (int)(gridDim.x * gridDim.y) + (int)(blockIdx.y * gridDim.x + blockIdx.x);
        if (rb < f.rblocks) tr_wgrad_reduce_items(f.rpart, f.rslices, f.rN, f.rCin, f.rtaps, f.rdw, rb, f.rblocks, NT);
        return;
    }
    const int Cin = p.Cin, Cin_pad = p.Cin_pad, W = p.Wout, H = p.Hout, N = p.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kg = lane >> 5;
    const int px0 = blockIdx.x * PT, n0 = blockIdx.y * BN;
    const int WCt = PT / H, HP2 = H + 2, HP = (WCt + 2) * HP2;
    const int b = px0 / (W * H), w0 = (px0 / H) % W;                 // the tile = WCt whole columns of image b
    const bool in_gn = FU && f.cs0 != nullptr;
    const int fC0 = FU ? f.C0 : Cin;
    const float* const x1p = FU ? f.x1 : nullptr;
    // halo staging roles: pass j: halo pixel hp = PT j + (tid >> 2), quarter (tid & 3) of the chunk's 64 channels
    const int sq = tid & 3;
    const float* hsrc[3];
    size_t hpix[3];
    bool hlive[3], hin[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int hp = PT * j + (tid >> 2);
        hin[j] = hp < HP;
        const int wc = hp / HP2, hr = hp - wc * HP2;
        int w = w0 - 1 + wc;
        w = w < 0 ? w + W : (w >= W ? w - W : w);
        const int h = hr - 1;
        hlive[j] = hin[j] && h >= 0 && h < H;
        hpix[j] = (size_t)(b * W + w) * H + (hlive[j] ? h : 0);
        hsrc[j] = p.x + hpix[j] * Cin + sq * 16;
    }
    const int npass = (HP + PT - 1) / PT;                            // 2, or 3 (H = 2)
    const int swr = tid / TPR, shalf = tid % TPR;
    const int wrow = n0 + swr;
    const bf16_t* wptr = p.w + (size_t)(wrow < N ? wrow : 0) * 9 * Cin_pad + shalf * (CK / TPR);
    const int nck = Cin / CK;
    constexpr int D = 3;                            // weight stages in flight (one L2 latency = about three stage times)
    f32x4 xr[3][4];
    u32x4 wr[D][WV];
    auto fetch_x = [&](int cc) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < npass && hin[j]) {
                const float* src = hsrc[j] + cc * CK;
                if constexpr (FU) {                 // two sources: the chunk lies in one of them (C0 % 64 == 0)
                    const int c = cc * CK + sq * 16;
                    src = c < fC0 ? p.x + hpix[j] * fC0 + c : x1p + hpix[j] * (Cin - fC0) + (c - fC0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) xr[j][q] = reinterpret_cast<const f32x4*>(src)[q];
            }
    };
    // (FU) GroupNorm (+ SiLU) of halo pass j's staged values, in their registers.  For every chunk but the first it runs during the previous
    // chunk's last three taps, between their MFMAs (the values have been in flight since tap 2) -- as part of stash_x the whole workgroup
    // did VALU work between two barriers while the matrix pipe idled: + 9 us per 256x16 conv.
    auto xform = [&](int cc, int j) __attribute__((always_inline)) {
        if (j < npass && hin[j]) {
            const float2* co = sSc + cc * CK + sq * 16;
            const bool act = f.silu != 0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float2 c_ = co[4 * q + e];
                    float z = xr[j][q][e] * c_.x + c_.y;
                    if (act) z *= tr_sigmoid(z);
                    xr[j][q][e] = z;
                }
        }
    };
    auto stash_x = [&](int cc, bool xformed) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < npass && hin[j]) {
                const float lv = hlive[j] ? 1.f : 0.f;
                bf16_t* dst = sXh + (PT * j + (tid >> 2)) * PITCH + sq * 16;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f32x4 a = xr[j][2 * q], c = xr[j][2 * q + 1];
                    if constexpr (FU) {
                        if (in_gn && !xformed) {    // GroupNorm (+ SiLU) once per staged element; the halo's zero rows stay zeros (lv)
                            const float2* co = sSc + cc * CK + sq * 16 + 8 * q;
                            const bool act = f.silu != 0;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float2 c0_ = co[e], c1_ = co[4 + e];
                                float z0 = a[e] * c0_.x + c0_.y, z1 = c[e] * c1_.x + c1_.y;
                                if (act) { z0 *= tr_sigmoid(z0); z1 *= tr_sigmoid(z1); }
                                a[e] = z0; c[e] = z1;
                            }
                        }
                    }
                    u32x4 u;
                    u.x = rldm::pack_bf16x2(a.x * lv, a.y * lv); u.y = rldm::pack_bf16x2(a.z * lv, a.w * lv);
                    u.z = rldm::pack_bf16x2(c.x * lv, c.y * lv); u.w = rldm::pack_bf16x2(c.z * lv, c.w * lv);
                    *reinterpret_cast<u32x4*>(dst + 8 * q) = u;
                }
            }
    };
#define HALO_FETCH_W(SLOT, TAP, CC)                                                                      \
    {                                                                                                    \
        const bf16_t* src_ = wptr + (size_t)(TAP) * Cin_pad + (CC) * CK;                                 \
        _Pragma("unroll") for (int q = 0; q < WV; ++q) wr[SLOT][q] = reinterpret_cast<const u32x4*>(src_)[q]; \
    }
#define HALO_STASH_W(SLOT)                                                                               \
    _Pragma("unroll") for (int q = 0; q < WV; ++q)                                                       \
        *reinterpret_cast<u32x4*>(sW + swr * PITCH + shalf * (CK / TPR) + 8 * q) = wr[SLOT][q];
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    // this lane's pixel of the tile -> its row in the halo image
    constexpr int PG = PT / 32;                     // pixel groups of 32
    const int lp = (wave % PG) * 32 + l31;
    const int r0 = (lp / H + 1) * HP2 + (lp % H) + 1;
    const bf16_t* bx0 = sXh + r0 * PITCH + 8 * kg;
    const bf16_t* aw0 = sW + ((wave / PG) * (BN / 2) + l31) * PITCH + 8 * kg;
    const bf16_t* aw1 = aw0 + (NR == 2 ? 32 : 0) * PITCH;
    fetch_x(0);
    HALO_FETCH_W(0, 0, 0)
    HALO_FETCH_W(1, 1, 0)
    HALO_FETCH_W(2, 2, 0)
    if constexpr (FU) {                             // (behind the first loads; visible after the loop's first barrier)
        if (in_gn) tr_gn_coeffs(sSc, 0, Cin, f.cs0, f.cs1, fC0, Cin, b, f.groups, f.eps, W * H, f.gamma, f.beta);
        if (f.gs_out) tr_gn_coeffs_tile(sCe, BN, n0, f.gcs0, f.gcs1, f.G0, N, b, f.ggroups, f.geps, W * H, f.ggamma, f.gbeta);
    }
    for (int cc = 0; cc < nck; ++cc) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {         // (unrolled: the ring slot tap % D is a compile-time register set)
            __syncthreads();                        // everyone is done reading the previous stage (and, at tap 0, the halo tile)
            if (tap == 0) stash_x(cc, cc > 0);
            HALO_STASH_W(tap % D)
            __syncthreads();
            if (tap + D < 9) HALO_FETCH_W(tap % D, tap + D, cc)
            else if (cc + 1 < nck) HALO_FETCH_W(tap % D, tap + D - 9, cc + 1)
            if (tap == 2 && cc + 1 < nck) fetch_x(cc + 1);          // the next chunk's halo rides in registers for six taps
            const int toff = ((tap / 3) - 1) * HP2 + (tap % 3) - 1;
            const bf16_t* bx = bx0 + toff * PITCH;
#pragma unroll
            for (int ks = 0; ks < CK / 16; ++ks) {
                const bf16x8 bv = *reinterpret_cast<const bf16x8*>(bx + 16 * ks);
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(aw0 + 16 * ks);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bv, acc0, 0, 0, 0);
                if (NR == 2) {
                    const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(aw1 + 16 * ks);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bv, acc1, 0, 0, 0);
                }
            }
            if constexpr (FU) {
                if (in_gn && tap >= 6 && cc + 1 < nck) xform(cc + 1, tap - 6);
            }
        }
    }
    if constexpr (FU) {
        if (f.cs_out || f.gs_out) {
            float* tile = reinterpret_cast<float*>(hraw);
            __syncthreads();
            float* trow = tile + lp * (BN + 1) + (wave / PG) * (BN / 2);
#pragma unroll
            for (int h = 0; h < NR; ++h)
#pragma unroll
                for (int r = 0; r < 16; ++r) trow[32 * h + 8 * (r >> 2) + 4 * kg + (r & 3)] = h ? acc1[r] : acc0[r];
            __syncthreads();
            tr_tile_epilogue<BN, PT, NT>(tile, sCol, sCe, p, f, px0, n0, b, true, true);
            return;
        }
    }
    const int px = px0 + lp;
    const int nb = n0 + (wave / PG) * (BN / 2);
    float* yrow = p.y + (size_t)px * N;
    const float* rrow = p.res ? p.res + (size_t)px * N : nullptr;
#pragma unroll
    for (int h = 0; h < NR; ++h)
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
            const int ch = nb + 32 * h + 8 * (r >> 2) + 4 * kg;       // 4 consecutive channels per register quad (N % 4 == 0 here)
            if (ch >= N) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = h ? acc1[r + e] : acc0[r + e];
            if (p.bias) { const float4 t = *reinterpret_cast<const float4*>(p.bias + ch); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
            if (p.rowadd) { const float4 t = *reinterpret_cast<const float4*>(p.rowadd + (size_t)b * p.rowadd_ld + ch); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
            if (rrow) { const float4 t = *reinterpret_cast<const float4*>(rrow + ch); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
            if (p.accumulate) { const float4 t = *reinterpret_cast<const float4*>(yrow + ch); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
            *reinterpret_cast<float4*>(yrow + ch) = make_float4(v[0], v[1], v[2], v[3]);
        }
#undef HALO_FETCH_W
#undef HALO_STASH_W
}

// ---- weight gradient --------------------------------------------------------------------------------------------------
struct TrWgrad {
    const float* dy; const float* x; float* dw;   // dw: partial sums [taps][slices][N][Cin], slices = grid.z * 4 waves
    int B, Win, Hin, Cin, Wout, Hout, N, taps, stride, mode, chunk;      // chunk: pixels per wave (multiple of 16)
};

// 16 consecutive channels of one pixel row: `valid` of them exist; vec: the row pointer is 16-byte aligned
__device__ inline void load_row16(const float* p, int valid, bool vec, float* out) {
    if (vec && valid >= 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = reinterpret_cast<const float4*>(p)[q];
            out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) out[e] = e < valid ? p[e] : 0.f;
    }
}

// grid (n tiles * c tiles, taps, K splits), 4 waves: wave v contracts pixels [((z * 4 + v) * chunk), + chunk) for a 64 x 64
// tile of dW (2 x 2 MFMA tiles).  D[n][c] += A[n][k] B[k][c] with k = pixel, while memory has channels contiguous: every
// k-step the wave loads 16 pixel rows x 64 channels of dy and of x (tap-shifted) with coalesced 16-byte loads into its
// private LDS tile and reads the fragments back transposed (8 pixels of one channel per lane).
constexpr int WG_PITCH = 68;                     // floats: 16-byte aligned rows, conflict-free column reads
__global__ __launch_bounds__(256) void tr_wgrad_kernel(const TrWgrad p) {
    __shared__ __attribute__((aligned(16))) float sm[4][2][16][WG_PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kg = lane >> 5;
    const int ct = (p.Cin + 63) / 64;
    const int n0 = (blockIdx.x / ct) * 64, c0 = (blockIdx.x % ct) * 64;
    const int t = blockIdx.y;
    const int dw = p.taps == 9 ? t / 3 - 1 : 0, dh = p.taps == 9 ? t % 3 - 1 : 0;
    const int P = p.B * p.Wout * p.Hout;
    const int kbeg = (blockIdx.z * 4 + wave) * p.chunk;
    const int kend = min(kbeg + p.chunk, P);
    float (*sA)[WG_PITCH] = sm[wave][0];
    float (*sB)[WG_PITCH] = sm[wave][1];
    const int row = lane >> 2, seg = (lane & 3) * 16;          // staging role: pixel row of the k-step, 16-channel segment
    const bool vecA = (p.N & 3) == 0, vecB = (p.Cin & 3) == 0;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        const int px = k0 + row;
        float va[16], vb[16];
        int validA = 0, validB = 0;
        const float* pa = p.dy;
        const float* pb = p.x;
        if (px < kend) {
            validA = p.N - (n0 + seg);
            pa = p.dy + (size_t)px * p.N + n0 + seg;
            const int ho = px % p.Hout, t1 = px / p.Hout, wo = t1 % p.Wout, b = t1 / p.Wout;
            const int sp = src_pixel(b, wo, ho, dw, dh, p.stride, p.mode, p.Win, p.Hin);
            if (sp >= 0) {
                validB = p.Cin - (c0 + seg);
                pb = p.x + (size_t)sp * p.Cin + c0 + seg;
            }
        }
        load_row16(pa, validA, vecA, va);
        load_row16(pb, validB, vecB, vb);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<float4*>(&sA[row][seg + 4 * q]) = make_float4(va[4 * q], va[4 * q + 1], va[4 * q + 2], va[4 * q + 3]);
            *reinterpret_cast<float4*>(&sB[row][seg + 4 * q]) = make_float4(vb[4 * q], vb[4 * q + 1], vb[4 * q + 2], vb[4 * q + 3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float a0[8], a1[8], b0[8], b1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a0[j] = sA[8 * kg + j][l31]; a1[j] = sA[8 * kg + j][32 + l31];
            b0[j] = sB[8 * kg + j][l31]; b1[j] = sB[8 * kg + j][32 + l31];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        auto pack8 = [](const float* f) {
            uint4 u;
            u.x = rldm::pack_bf16x2(f[0], f[1]); u.y = rldm::pack_bf16x2(f[2], f[3]);
            u.z = rldm::pack_bf16x2(f[4], f[5]); u.w = rldm::pack_bf16x2(f[6], f[7]);
            return __builtin_bit_cast(bf16x8, u);
        };
        const bf16x8 A0 = pack8(a0), A1 = pack8(a1), B0 = pack8(b0), B1 = pack8(b1);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc[1][1], 0, 0, 0);
    }
    // every wave stores its partial tile (coalesced along c); slices that had no pixels store zeros.
    // tile (i, j): lane (column c0 + 32 j + l31, half kg) holds rows n0 + 32 i + (r & 3) + 8 (r >> 2) + 4 kg
    const int slices = gridDim.z * 4, slice = blockIdx.z * 4 + wave;
    float* part = p.dw + ((size_t)t * slices + slice) * p.N * p.Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + 32 * j + l31;
            if (c >= p.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = n0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (nn < p.N) part[(size_t)nn * p.Cin + c] = acc[i][j][r];
            }
        }
}

// Weight gradient, all taps in one workgroup (stride 1; every 3x3 / 1x1 conv of the UNet body).  The kernel above gives a
// workgroup ONE tap, so dy and x are read from L2 / HBM nine times per tile pair (604 MB for a 33 MB level-0 problem: it ran at
// the memory system's speed, 85 TFLOP/s).  Here a workgroup owns a 64 x 64 (n, c) tile for ALL taps over a range of pixel
// chunks.  A chunk = WC azimuth columns x all H beams of one image, staged once, transposed, as bf16:
//   sA[n][k]            k = h * WC + wl  (beam-major inside the chunk, so 8 consecutive k = 8 azimuth neighbours of a beam)
//   sB[d][c][(h + 1) * WC + wl] = x[w0 + wl + d - 1][h]   three copies pre-shifted by the azimuth tap (circular halo), beam
//                       rows -1 and H are zeros written once: the beam tap is a shift by WC elements = 16-byte aligned
// so every MFMA fragment of every tap is one aligned 16-byte LDS read: a wave does 9 MFMAs (one per tap, 32 x 32 x 16, its
// own quarter of the tile) per 10 fragment reads.  Staging lanes = 4 channel quads x 16 pixel pairs: 64-byte global segments
// and conflict-free 4-byte transposed LDS stores (pitch = odd number of 16-byte slots).  mode 1 (nearest x2 in front of the
// conv) is an index map at staging.  part: [taps][grid.y][N][Cin] partial sums (summed by tr_wgrad_reduce_kernel).
struct TrWgrad2 {
    const float* dy; const float* x; float* part;
    int B, W, H, Win, Hin, Cin, N, mode, lwc, cpw, nchunks, pitchA, pitchB;
    float* dw;                 // non-null: add the tile straight into the fp32 gradient [N][Cin][taps] (coalesced atomics), no partials
    float* rows; int rows_ld;  // bias / time-embedding-row gradients from the staged dy tile (workgroups of channel tile 0):
    float* total;              //   rows[b][n] += sum over image b's pixels, total[n] += sum over all pixels; either may be null
};

// (round 5) FU: x = act(GroupNorm(cat(x, x1))) rebuilt while staging (TrFuse's input side; the 64-channel tile lies in one source)
// (round 6) the kernel's body as a function of its place in the launch -- tile bx of the layer, K slice z of Z -- so that ONE launch can
// run the weight gradients of many layers (tr_wgrad2_group_kernel below).  GROUP: the slices of a tile meet through an arrival ticket
// and the LAST arriver sums the partial tiles in slice order into dw (deterministic; no reduction launch); Z == 1: the tile goes
// straight into dw.
// (round 6) V3: the staging rebuilt for images of 8 / 16 beams (the two high-resolution levels).  A LANE owns a channel of the tile and a
// wave every fourth 8-beam SEGMENT of the chunk: a wave-load is one 256-byte line, the contraction index runs beam-fastest
// (k = column * H + beam), so a segment is 16 contiguous bytes of its channel's LDS row (one ds_write_b128 instead of eight transposing
// ds_write_b32), GroupNorm + SiLU are applied once per element (the round-5 staging transformed every input element for each of its four
// azimuth-shifted roles) with the channel's coefficients in two registers, and the three copies a 3x3 needs are BEAM-shifted -- built
// from the segment's own ten registers -- while the azimuth taps are a shift by H elements (16-byte aligned for H >= 8) into a row
// staged with one halo column each side.  84 KB instead of 98 KB of fp32 per chunk and a fifth of the staging instructions -- and the
// grouped 3x3 launch of the two high-resolution levels takes 684 instead of 701 us (1x1: 174 / 180): what bounds these launches is not
// the staging but the L2-MISS TRAFFIC of fp32 activations (every (layer, slice) is read by (N / 64)(C / 64) tiles spread over the eight
// XCDs: 1.5 GB per launch at ~2.3 TB/s; profiles/round6_wgrad_ablation.txt).  bf16 activations would halve it; the tape stores fp32.
template <int TAPS, bool FU, bool GROUP, bool V3 = false>
__device__ __forceinline__ void tr_wgrad2_body(const TrWgrad2& p, const TrFuse& f, const int bx, const int z, const int Z, unsigned* tickets) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
    constexpr int NC = TAPS == 9 ? 3 : 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kg = lane >> 5;
    const int WC = 1 << p.lwc, H = p.H, W = p.W, N = p.N, Cin = p.Cin;
    const int KP = WC * H;
    const int pitchA = p.pitchA, pitchB = V3 ? (WC + (TAPS == 9 ? 2 : 0)) * H + 8 : p.pitchB;
    bf16_t* sA = reinterpret_cast<bf16_t*>(wg_smem);                 // [64][pitchA]
    bf16_t* sB = sA + 64 * pitchA;                                   // [NC][64][pitchB]
    const int ct = Cin >> 6;
    const int n0 = (bx / ct) * 64, c0 = (bx % ct) * 64;
    const int wi = wave >> 1, wj = wave & 1;                         // this wave's 32 x 32 quarter of the tile
    if (TAPS == 9 && !V3) {                                          // zero beam rows -1 and H of the three copies
        for (int e = tid; e < NC * 64 * 2 * WC; e += 256) {
            const int wl = e & (WC - 1), r = (e >> p.lwc) & 1, rc = e >> (p.lwc + 1);
            sB[rc * pitchB + (r ? (H + 1) * WC : 0) + wl] = (bf16_t)0;
        }
    }
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // staging role: 4 channel quads x 16 pixel pairs per wave (8 quads x 8 pairs = whole 128-byte lines measured the same)
    const int cq = 4 * wave + (lane & 3), ppl = lane >> 2;
    const int nwc = W >> p.lwc;
    const int sh = p.mode ? 1 : 0;
    struct WgStage { f32x4 d0, d1, xm, x0, x1, x2; };
    const float* const dyb = p.dy + n0 + 4 * cq;
    // (FU, two sources: the tile's 64 channels lie in one of them -- C0 % 64 == 0; xld = that source's channels per pixel)
    const bool second = FU && f.x1 != nullptr && c0 >= f.C0;
    const float* const xb = second ? f.x1 + (c0 - f.C0) + 4 * cq : p.x + c0 + 4 * cq;
    const int xld = FU ? (second ? Cin - f.C0 : f.C0) : Cin;
    const int Win = p.Win, Hin = p.Hin, lwc = p.lwc;
    // (FU) GroupNorm coefficients of the tile's channels for every image: sCo[b * 64 + c] = (a, b); after sB in the dynamic LDS
    float2* const sCo = reinterpret_cast<float2*>(sB + NC * 64 * pitchB);
    const bool in_gn = FU && f.cs0 != nullptr;
    if constexpr (FU) {
        if (in_gn) {
            const int cpg = Cin / f.groups, C1 = Cin - f.C0, npix = Win * Hin;
            for (int e = tid; e < p.B * 64; e += 256) {
                const int bi = e >> 6, ch = c0 + (e & 63), g = ch / cpg;
                double s_ = 0.0, ss_ = 0.0;
                for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
                    const float2 v = c < f.C0 ? reinterpret_cast<const float2*>(f.cs0)[(size_t)bi * f.C0 + c]
                                              : reinterpret_cast<const float2*>(f.cs1)[(size_t)bi * C1 + c - f.C0];
                    s_ += (double)v.x; ss_ += (double)v.y;
                }
                const double n = (double)npix * cpg, mean = s_ / n;
                double var = ss_ / n - mean * mean;
                var = var < 0.0 ? 0.0 : var;
                const float a = (float)(1.0 / sqrt(var + (double)f.eps)) * f.gamma[ch];
                sCo[e] = make_float2(a, f.beta[ch] - (float)mean * a);
            }
            // (visible after the first __syncthreads() of the chunk loop, before the first stash)
        }
    }
    auto fetch = [&](int pp, int b, int w0, WgStage& r) __attribute__((always_inline)) {
        const int k = 2 * pp, h = k >> lwc, wl = k & (WC - 1), hs = h >> sh;
        const float* src = dyb + ((size_t)(b * W + w0 + wl) * H + h) * N;
        r.d0 = *reinterpret_cast<const f32x4*>(src);
        r.d1 = *reinterpret_cast<const f32x4*>(src + (size_t)H * N);
        r.x0 = *reinterpret_cast<const f32x4*>(xb + ((size_t)(b * Win + ((w0 + wl) >> sh)) * Hin + hs) * xld);
        r.x1 = *reinterpret_cast<const f32x4*>(xb + ((size_t)(b * Win + ((w0 + wl + 1) >> sh)) * Hin + hs) * xld);
        if (TAPS == 9) {
            int wm = w0 + wl - 1, w2 = w0 + wl + 2;
            wm = wm < 0 ? wm + W : wm;
            w2 = w2 >= W ? w2 - W : w2;
            r.xm = *reinterpret_cast<const f32x4*>(xb + ((size_t)(b * Win + (wm >> sh)) * Hin + hs) * xld);
            r.x2 = *reinterpret_cast<const f32x4*>(xb + ((size_t)(b * Win + (w2 >> sh)) * Hin + hs) * xld);
        }
    };
    auto stash = [&](int pp, WgStage& r, int b) __attribute__((always_inline)) {
        const int k = 2 * pp, h = k >> lwc, wl = k & (WC - 1);
        if constexpr (FU) {
            if (in_gn) {
                const float2* co = sCo + b * 64 + 4 * cq;
                const bool act = f.silu != 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float2 cf = co[e];
                    float z0 = r.x0[e] * cf.x + cf.y, z1 = r.x1[e] * cf.x + cf.y;
                    if (act) { z0 *= tr_sigmoid(z0); z1 *= tr_sigmoid(z1); }
                    r.x0[e] = z0; r.x1[e] = z1;
                    if (TAPS == 9) {
                        float zm = r.xm[e] * cf.x + cf.y, z2 = r.x2[e] * cf.x + cf.y;
                        if (act) { zm *= tr_sigmoid(zm); z2 *= tr_sigmoid(z2); }
                        r.xm[e] = zm; r.x2[e] = z2;
                    }
                }
            }
        }
        uint32_t* da = reinterpret_cast<uint32_t*>(sA + (4 * cq) * pitchA + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) da[e * (pitchA >> 1)] = rldm::pack_bf16x2(r.d0[e], r.d1[e]);
        if (TAPS == 9) {
            uint32_t* dst = reinterpret_cast<uint32_t*>(sB + (4 * cq) * pitchB + (h + 1) * WC + wl);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dst[e * (pitchB >> 1)] = rldm::pack_bf16x2(r.xm[e], r.x0[e]);
                dst[(64 + e) * (pitchB >> 1)] = rldm::pack_bf16x2(r.x0[e], r.x1[e]);
                dst[(128 + e) * (pitchB >> 1)] = rldm::pack_bf16x2(r.x1[e], r.x2[e]);
            }
        } else {
            uint32_t* dst = reinterpret_cast<uint32_t*>(sB + (4 * cq) * pitchB + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e * (pitchB >> 1)] = rldm::pack_bf16x2(r.x0[e], r.x1[e]);
        }
    };
    const int chunk_end = min((z + 1) * p.cpw, p.nchunks);
    const bool do_sums = (p.rows || p.total) && c0 == 0;
    int sum_b = -1;
    float sum_img = 0.f, sum_all = 0.f;
    if constexpr (!V3) {
    WgStage R[4];                                                    // one chunk of staging data (1 - 4 iterations of 6 loads)
    if (z * p.cpw < chunk_end) {
        const int c0_ = z * p.cpw, b0_ = c0_ / nwc, w00 = (c0_ - b0_ * nwc) << p.lwc;
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (2 * (ppl + 16 * it) < KP) fetch(ppl + 16 * it, b0_, w00, R[it]);
    }
    for (int chunk = z * p.cpw; chunk < chunk_end; ++chunk) {
        const int b = chunk / nwc, w0 = (chunk - b * nwc) << p.lwc;
        __syncthreads();                                             // the previous chunk's fragments have been read
#if !(RLDM_TR_ABL & 16)
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (2 * (ppl + 16 * it) < KP) stash(ppl + 16 * it, R[it], b);
#endif
        __syncthreads();
        if (do_sums) {                                               // thread (row n = tid >> 2, quarter of the chunk's pixels)
            const bf16_t* r = sA + (tid >> 2) * pitchA + (tid & 3) * (KP >> 2);
            float sacc = 0.f;
            for (int k = 0; k < (KP >> 2); k += 2) {
                const uint32_t u = *reinterpret_cast<const uint32_t*>(r + k);
                sacc += rldm::bf16lo(u) + rldm::bf16hi(u);
            }
            sacc += __shfl_xor(sacc, 1);
            sacc += __shfl_xor(sacc, 2);
            if (b != sum_b) {
                if (sum_b >= 0 && p.rows && (tid & 3) == 0) unsafeAtomicAdd(p.rows + (size_t)sum_b * p.rows_ld + n0 + (tid >> 2), sum_img);
                sum_b = b;
                sum_img = 0.f;
            }
            sum_img += sacc;
            sum_all += sacc;
        }
#if !(RLDM_TR_ABL & 64)
        if (chunk + 1 < chunk_end) {                                 // the next chunk's loads fly during this chunk's MFMAs
            const int nb_ = (chunk + 1) / nwc, nw0 = ((chunk + 1) - nb_ * nwc) << p.lwc;
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (2 * (ppl + 16 * it) < KP) fetch(ppl + 16 * it, nb_, nw0, R[it]);
        }
#endif
#if RLDM_TR_ABL & 32
        if (chunk >= 0) continue;
#endif
        const bf16_t* fa = sA + (32 * wi + l31) * pitchA + 8 * kg;
        const bf16_t* fb = sB + (32 * wj + l31) * pitchB + 8 * kg;
        // fragments of k-step k + 1 are requested before the MFMAs of k-step k (one wave per SIMD: nothing else hides the LDS latency)
        if (TAPS == 9) {
            uint4 Ac = *reinterpret_cast<const uint4*>(fa), Bc[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) Bc[t] = *reinterpret_cast<const uint4*>(fb + (t / 3) * 64 * pitchB + (t % 3) * WC);
            for (int k = 0; k < KP; k += 16) {
                uint4 An = Ac, Bn[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) Bn[t] = Bc[t];
                if (k + 16 < KP) {
                    An = *reinterpret_cast<const uint4*>(fa + k + 16);
#pragma unroll
                    for (int t = 0; t < 9; ++t) Bn[t] = *reinterpret_cast<const uint4*>(fb + (t / 3) * 64 * pitchB + k + 16 + (t % 3) * WC);
                }
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Ac), __builtin_bit_cast(bf16x8, Bc[t]), acc[t], 0, 0, 0);
                Ac = An;
#pragma unroll
                for (int t = 0; t < 9; ++t) Bc[t] = Bn[t];
            }
        } else {
#pragma unroll 2
            for (int k = 0; k < KP; k += 16) {
                const bf16x8 A = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(fa + k));
                const bf16x8 Bf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(fb + k));
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bf, acc[0], 0, 0, 0);
            }
        }
    }
    } else {
        // ---- V3 staging (see the comment above the function) ----
        constexpr int HALO = TAPS == 9 ? 1 : 0, NE = TAPS == 9 ? 10 : 8;
        const int hsh = H == 16 ? 1 : 0, hmask = (1 << hsh) - 1;       // 8-beam segments per column: 1 << hsh
        const int nsx = (WC + 2 * HALO) << hsh, nsd = WC << hsh;         // segments of the (halo'd) input chunk / of dy (KP / 8 <= 16)
        const float* const dyc = p.dy + n0 + lane;
        const float* const xc = (second ? f.x1 + (c0 - f.C0) : p.x + c0) + lane;
        const size_t sN = (size_t)N, sX = (size_t)xld;
        float dv[4][8], xv[5][NE];
        auto fetch3 = [&](const int b, const int w0) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // (a chunk of fewer than 16 segments -- 64 pixels -- : the surplus slots re-load the last segment and are not stored; a
                //  branch around the loads instead cost the common 16-segment case 20 %: 840 against 684 us for the grouped 3x3 launch)
                const int sg = min(wave + 4 * i, nsd - 1), wl = sg >> hsh, h0 = (sg & hmask) * 8;
                const float* src = dyc + ((size_t)(b * W + w0 + wl) * H + h0) * sN;
#pragma unroll
                for (int e = 0; e < 8; ++e) dv[i][e] = src[(size_t)e * sN];
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int sg = wave + 4 * i;
                if (sg < nsx) {
                    const int j = sg >> hsh, h0 = (sg & hmask) * 8;
                    int w = w0 + j - HALO;
                    w = w < 0 ? w + W : (w >= W ? w - W : w);
                    const float* src = xc + ((size_t)(b * W + w) * H + h0) * sX;
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
                        const int h = h0 + e - HALO;
                        xv[i][e] = (h >= 0 && h < H) ? src[((ptrdiff_t)e - HALO) * (ptrdiff_t)sX] : 0.f;
                    }
                }
            }
        };
        auto stash3 = [&](const int b) __attribute__((always_inline)) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sg = wave + 4 * i;
                if (sg < nsd) {
                    u32x4 u;
                    u.x = rldm::pack_bf16x2(dv[i][0], dv[i][1]); u.y = rldm::pack_bf16x2(dv[i][2], dv[i][3]);
                    u.z = rldm::pack_bf16x2(dv[i][4], dv[i][5]); u.w = rldm::pack_bf16x2(dv[i][6], dv[i][7]);
                    *reinterpret_cast<u32x4*>(sA + lane * pitchA + 8 * sg) = u;
                }
            }
            float2 cf = make_float2(1.f, 0.f);
            if constexpr (FU) {
                if (in_gn) cf = sCo[b * 64 + lane];
            }
            const bool act = FU && in_gn && f.silu != 0;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int sg = wave + 4 * i;
                if (sg < nsx) {
                    const int h0 = (sg & hmask) * 8;
                    float v[NE];
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
                        float zz = xv[i][e];
                        if constexpr (FU) {
                            if (in_gn) {
                                zz = zz * cf.x + cf.y;
                                if (act) zz *= tr_sigmoid(zz);
                            }
                        }
                        v[e] = zz;
                    }
                    bf16_t* dst = sB + lane * pitchB + 8 * sg;
                    if constexpr (TAPS == 9) {
                        // (zero padding applies to the ACTIVATED tensor: the beams above / below the image)
                        if (h0 == 0) v[0] = 0.f;
                        if (h0 + 8 == H) v[9] = 0.f;
                        const unsigned p01 = rldm::pack_bf16x2(v[0], v[1]), p23 = rldm::pack_bf16x2(v[2], v[3]), p45 = rldm::pack_bf16x2(v[4], v[5]),
                                       p67 = rldm::pack_bf16x2(v[6], v[7]), p89 = rldm::pack_bf16x2(v[8], v[9]);
                        u32x4 m, lo, hi;                             // copies dh = 0 | -1 | +1: position h holds a[h + dh]
                        m.x = rldm::pack_bf16x2(v[1], v[2]); m.y = rldm::pack_bf16x2(v[3], v[4]); m.z = rldm::pack_bf16x2(v[5], v[6]); m.w = rldm::pack_bf16x2(v[7], v[8]);
                        lo.x = p01; lo.y = p23; lo.z = p45; lo.w = p67;
                        hi.x = p23; hi.y = p45; hi.z = p67; hi.w = p89;
                        *reinterpret_cast<u32x4*>(dst) = lo;
                        *reinterpret_cast<u32x4*>(dst + 64 * pitchB) = m;
                        *reinterpret_cast<u32x4*>(dst + 128 * pitchB) = hi;
                    } else {
                        u32x4 m;
                        m.x = rldm::pack_bf16x2(v[0], v[1]); m.y = rldm::pack_bf16x2(v[2], v[3]); m.z = rldm::pack_bf16x2(v[4], v[5]); m.w = rldm::pack_bf16x2(v[6], v[7]);
                        *reinterpret_cast<u32x4*>(dst) = m;
                    }
                }
            }
        };
        if (z * p.cpw < chunk_end) {
            const int c0_ = z * p.cpw, b0_ = c0_ / nwc, w00 = (c0_ - b0_ * nwc) << p.lwc;
            fetch3(b0_, w00);
        }
        for (int chunk = z * p.cpw; chunk < chunk_end; ++chunk) {
            const int b = chunk / nwc;
            __syncthreads();                                         // the previous chunk's fragments have been read
            stash3(b);
            __syncthreads();
            if (do_sums) {                                           // thread (row n = tid >> 2, quarter of the chunk's pixels)
                const bf16_t* r = sA + (tid >> 2) * pitchA + (tid & 3) * (KP >> 2);
                float sacc = 0.f;
                for (int k = 0; k < (KP >> 2); k += 2) {
                    const uint32_t u = *reinterpret_cast<const uint32_t*>(r + k);
                    sacc += rldm::bf16lo(u) + rldm::bf16hi(u);
                }
                sacc += __shfl_xor(sacc, 1);
                sacc += __shfl_xor(sacc, 2);
                if (b != sum_b) {
                    if (sum_b >= 0 && p.rows && (tid & 3) == 0) unsafeAtomicAdd(p.rows + (size_t)sum_b * p.rows_ld + n0 + (tid >> 2), sum_img);
                    sum_b = b;
                    sum_img = 0.f;
                }
                sum_img += sacc;
                sum_all += sacc;
            }
            if (chunk + 1 < chunk_end) {                             // the next chunk's loads fly during this chunk's MFMAs
                const int nb_ = (chunk + 1) / nwc, nw0 = ((chunk + 1) - nb_ * nwc) << p.lwc;
                fetch3(nb_, nw0);
            }
            const bf16_t* fa = sA + (32 * wi + l31) * pitchA + 8 * kg;
            const bf16_t* fb = sB + (32 * wj + l31) * pitchB + 8 * kg;
            if (TAPS == 9) {
                // tap t = (dw + 1) * 3 + (dh + 1): copy t % 3 (beam shift), t / 3 columns of H elements into the halo'd row
                uint4 Ac = *reinterpret_cast<const uint4*>(fa), Bc[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) Bc[t] = *reinterpret_cast<const uint4*>(fb + (t % 3) * 64 * pitchB + (t / 3) * H);
                for (int k = 0; k < KP; k += 16) {
                    uint4 An = Ac, Bn[9];
#pragma unroll
                    for (int t = 0; t < 9; ++t) Bn[t] = Bc[t];
                    if (k + 16 < KP) {
                        An = *reinterpret_cast<const uint4*>(fa + k + 16);
#pragma unroll
                        for (int t = 0; t < 9; ++t) Bn[t] = *reinterpret_cast<const uint4*>(fb + (t % 3) * 64 * pitchB + k + 16 + (t / 3) * H);
                    }
#pragma unroll
                    for (int t = 0; t < 9; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Ac), __builtin_bit_cast(bf16x8, Bc[t]), acc[t], 0, 0, 0);
                    Ac = An;
#pragma unroll
                    for (int t = 0; t < 9; ++t) Bc[t] = Bn[t];
                }
            } else {
#pragma unroll 2
                for (int k = 0; k < KP; k += 16) {
                    const bf16x8 A = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(fa + k));
                    const bf16x8 Bf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(fb + k));
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bf, acc[0], 0, 0, 0);
                }
            }
        }
    }
    if (do_sums && (tid & 3) == 0) {
        if (sum_b >= 0 && p.rows) unsafeAtomicAdd(p.rows + (size_t)sum_b * p.rows_ld + n0 + (tid >> 2), sum_img);
        if (p.total) unsafeAtomicAdd(p.total + n0 + (tid >> 2), sum_all);
    }
    // lane (column c = c0 + 32 wj + l31, half kg), register r <-> row n0 + 32 wi + (r & 3) + 8 (r >> 2) + 4 kg
    if (GROUP ? Z == 1 : p.dw != nullptr) {
        // the tile in the gradient's own layout ([n][c][tap]: 64 * TAPS contiguous floats per row) through LDS, 32 rows at a
        // time, then along a row: one cache line per 32 lanes (atomics; GROUP with one slice: this workgroup is the tile's only writer)
        constexpr int TP = 64 * TAPS + 1;
        float* tile = reinterpret_cast<float*>(wg_smem);
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
            if (wi == half) {
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * kg) * TP + (32 * wj + l31) * TAPS + t] = acc[t][r];
            }
            __syncthreads();
            for (int e = tid; e < 32 * 64 * TAPS; e += 256) {
                const int row = e / (64 * TAPS), col = e - row * (64 * TAPS);
                float* dst = p.dw + ((size_t)(n0 + 32 * half + row) * Cin + c0) * TAPS + col;
                if (GROUP) *dst += tile[row * TP + col];
                else unsafeAtomicAdd(dst, tile[row * TP + col]);
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        float* part = p.part + (((size_t)t * Z + z) * N + n0 + 32 * wi + 4 * kg) * Cin + c0 + 32 * wj + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(size_t)((r & 3) + 8 * (r >> 2)) * Cin] = acc[t][r];
    }
    if constexpr (GROUP) {
        // the tile's Z slices meet here: every workgroup publishes its partial tile (release at agent scope: the stores are written back
        // past this XCD's L2) and takes a ticket; the last one invalidates its caches (acquire) and sums the Z partial tiles in slice
        // order -- the same order whatever the arrival order, so the gradient is bit-reproducible -- into dw, and re-arms the ticket.
        __shared__ unsigned s_last;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(tickets + bx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == (unsigned)(Z - 1) ? 1u : 0u;
            if (s_last) __hip_atomic_store(tickets + bx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!s_last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // 16 columns x 16 rows per pass (a thread = one (row, column quad): 16-byte loads along c), slices in flight eight at a time
        const size_t nc = (size_t)N * Cin;
        const int cq4 = (tid & 15) * 4, rr = tid >> 4;
        for (int t = 0; t < TAPS; ++t)
            for (int r0 = 0; r0 < 64; r0 += 16) {
                const size_t e = (size_t)(n0 + r0 + rr) * Cin + c0 + cq4;
                const float* src = p.part + (size_t)t * Z * nc + e;
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                int sidx = 0;
                for (; sidx + 8 <= Z; sidx += 8) {
                    f32x4 v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(src + (size_t)(sidx + j) * nc);
#pragma unroll
                    for (int j = 0; j < 8; ++j) a += v[j];
                }
                for (; sidx < Z; ++sidx) a += *reinterpret_cast<const f32x4*>(src + (size_t)sidx * nc);
#pragma unroll
                for (int q = 0; q < 4; ++q) p.dw[(e + q) * TAPS + t] += a[q];
            }
    }
}

template <int TAPS, bool FU = false>
__global__ __launch_bounds__(256) void tr_wgrad2_kernel(const TrWgrad2 p, const TrFuse f) {
    tr_wgrad2_body<TAPS, FU, false>(p, f, blockIdx.x, blockIdx.y, gridDim.y, nullptr);
}

// ---- (round 6) the weight gradients of MANY layers in one launch --------------------------------------------------------------------
// A weight gradient depends on nothing that comes after it in backward, and nothing in backward depends on it: launched where the tape
// reaches them, the ~85 all-taps launches of a step each ran alone on the chip at its launch + first-touch + drain floor (1.8 ms of a
// 9.3 ms step for 0.27 TFLOP).  The host queues them instead (rldm_train_wgrad_group) and hands the queue to this kernel in groups of
// up to kWgGroupMax layers of one instantiation (the kernel-argument block holds the records: capture-safe, no descriptor upload):
// block id -> (layer, tile, K slice).
constexpr int kWgGroupMax = 22;
struct WgItem {
    const float* dy; const float* x; const float* x1; float* dw; float* rows; float* total; float* part;
    const float* cs0; const float* cs1; const float* gamma; const float* beta;
    int B, W, H, Win, Hin, Cin, N, mode, lwc, cpw, nchunks, Z, rows_ld, C0, silu, groups;
    float eps;
    unsigned ticket_off;
};
struct WgGroup {
    int n;
    unsigned* tickets;
    int first[kWgGroupMax + 1];         // first block of layer i (first[n] = the grid)
    WgItem item[kWgGroupMax];
};
static_assert(sizeof(WgGroup) <= 4096, "the group record is the kernel's argument block");

template <int TAPS, bool FU, bool V3>
__global__ __launch_bounds__(256) void tr_wgrad2_group_kernel(const WgGroup g) {
    const int bid = blockIdx.x;
    int l = 0;
    for (int i = 1; i < g.n; ++i) l = bid >= g.first[i] ? i : l;
    const WgItem& it = g.item[l];
    TrWgrad2 p;
    p.dy = it.dy; p.x = it.x; p.part = it.part; p.dw = it.dw; p.rows = it.rows; p.rows_ld = it.rows_ld; p.total = it.total;
    p.B = it.B; p.W = it.W; p.H = it.H; p.Win = it.Win; p.Hin = it.Hin; p.Cin = it.Cin; p.N = it.N; p.mode = it.mode; p.lwc = it.lwc;
    p.cpw = it.cpw; p.nchunks = it.nchunks;
    p.pitchA = (it.H << it.lwc) + 8;
    p.pitchB = ((TAPS == 9 ? it.H + 2 : it.H) << it.lwc) + 8;
    TrFuse f;
    f.x1 = it.x1; f.C0 = it.C0; f.cs0 = it.cs0; f.cs1 = it.cs1; f.gamma = it.gamma; f.beta = it.beta; f.silu = it.silu;
    f.groups = it.groups; f.eps = it.eps;
    // block -> (K slice, tile): tile fastest.  (Measured and dropped: a layout [Z / 8][tile][8] that puts the tiles of a slice -- which
    // read the same pixels of dy and x -- on one XCD behind one L2: 873 against 780 us for the 3x3 launch of the two high-resolution
    // levels, and the one-slice layers of the low levels, each on a single XCD, 529 against 307 us.  The kernel is bound by its staging
    // pipeline per CU -- one chunk of loads in flight, a transposing LDS write, two barriers per chunk -- not by L2 misses.)
    const int local = bid - g.first[l];
    const int tiles = (it.N >> 6) * (it.Cin >> 6);
    tr_wgrad2_body<TAPS, FU, true, V3>(p, f, local % tiles, local / tiles, it.Z, g.tickets + it.ticket_off);
}

// four channels per thread, eight slices in flight (Cin % 4 == 0): the scalar kernel below ran at 2 TB/s over 38 MB of partials
__global__ __launch_bounds__(256) void tr_wgrad_reduce_vec_kernel(const float* __restrict__ part, int slices, int N, int Cin, int taps,
                                                                  float* __restrict__ dw) {
    tr_wgrad_reduce_items(part, slices, N, Cin, taps, dw, blockIdx.x, gridDim.x, 256);
}

// dw[n][c][t] += sum over slices of part[t][slice][n][c]   (one thread per (t, n, c); reads coalesced along c)
__global__ __launch_bounds__(256) void tr_wgrad_reduce_kernel(const float* __restrict__ part, int slices, int N, int Cin, int taps,
                                                              float* __restrict__ dw) {
    const size_t nc = (size_t)N * Cin;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nc * taps) return;
    const int t = (int)(i / nc);
    const size_t e = i % nc;
    const float* src = part + (size_t)t * slices * nc + e;
    float acc = 0.f;
    for (int s = 0; s < slices; ++s) acc += src[(size_t)s * nc];
    dw[e * taps + t] += acc;
}

// rows[b][n] += sum over the pixels of image b of dy[p][n]; total[n] += the same over all images.  grid (64-channel tiles,
// images, 256-pixel slabs), 256 threads = 16 pixel lanes x 16 channel quads (16-byte loads when N % 4 == 0): fp32 atomics
// into buffers the caller zeroed (rows) / accumulates into (total = bias gradient).
__global__ __launch_bounds__(256) void tr_colsum_kernel(const float* __restrict__ dy, int npix, int N, float* __restrict__ rows,
                                                        int rows_ld, float* __restrict__ total) {
    __shared__ float sh[16][68];
    const int b = blockIdx.y, cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int ch = blockIdx.x * 64 + cq * 4;
    const int p0 = blockIdx.z * 256, p1 = min(p0 + 256, npix);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const float* base = dy + (size_t)b * npix * N + ch;
    if ((N & 3) == 0 && ch + 3 < N) {
        for (int px = p0 + pl; px < p1; px += 16) {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)px * N);
            a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
        }
    } else {
        for (int px = p0 + pl; px < p1; px += 16) {
            const float* r = base + (size_t)px * N;
            if (ch < N) a0 += r[0];
            if (ch + 1 < N) a1 += r[1];
            if (ch + 2 < N) a2 += r[2];
            if (ch + 3 < N) a3 += r[3];
        }
    }
    sh[pl][cq * 4] = a0; sh[pl][cq * 4 + 1] = a1; sh[pl][cq * 4 + 2] = a2; sh[pl][cq * 4 + 3] = a3;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = blockIdx.x * 64 + threadIdx.x;
        if (c < N) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) v += sh[r][threadIdx.x];
            if (rows) unsafeAtomicAdd(rows + (size_t)b * rows_ld + c, v);
            if (total) unsafeAtomicAdd(total + c, v);
        }
    }
}

// ---- GroupNorm ----------------------------------------------------------------------------------------------------------
__device__ inline double block_sum_d(double v, double* sh) {
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    return t;
}

// grid (groups, B): stats[b][g] = (mean, rstd) over npix x cpg values
__global__ __launch_bounds__(256) void tr_gn_stats_kernel(const float* __restrict__ x, int npix, int C, int groups, float eps,
                                                          float2* __restrict__ stats) {
    __shared__ double sh[4];
    const int g = blockIdx.x, b = blockIdx.y, cpg = C / groups;
    const float* base = x + (size_t)b * npix * C + g * cpg;
    double s = 0.0, ss = 0.0;
    for (int e = threadIdx.x; e < npix * cpg; e += 256) {
        const float v = base[(size_t)(e / cpg) * C + (e % cpg)];
        s += v;
        ss += (double)v * v;
    }
    s = block_sum_d(s, sh);
    ss = block_sum_d(ss, sh);
    if (threadIdx.x == 0) {
        const double n = (double)npix * cpg, mean = s / n;
        double var = ss / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        stats[b * groups + g] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
    }
}

// 4 channels per thread for 4 | channels-per-group <= 16: thread (pixel lane t / Q, channel quad t % Q), Q = cpg / 4
__global__ __launch_bounds__(256) void tr_gn_stats_vec_kernel(const float* __restrict__ x, int npix, int C, int groups, float eps,
                                                              float2* __restrict__ stats) {
    __shared__ double sh[4];
    const int g = blockIdx.x, b = blockIdx.y, cpg = C / groups, Q = cpg >> 2;
    const int q = threadIdx.x % Q, pl = threadIdx.x / Q, step = 256 / Q;
    const float* base = x + (size_t)b * npix * C + g * cpg + 4 * q;
    double s = 0.0, ss = 0.0;
    for (int p = pl; p < npix; p += step) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)p * C);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s += v[e]; ss += (double)v[e] * v[e]; }
    }
    s = block_sum_d(s, sh);
    ss = block_sum_d(ss, sh);
    if (threadIdx.x == 0) {
        const double n = (double)npix * cpg, mean = s / n;
        double var = ss / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        stats[b * groups + g] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
    }
}

__device__ inline float sigmoid_f(float z) { return 1.f / (1.f + __expf(-z)); }

// Statistics AND apply in one launch for the small tensors (levels 1-3): the block of (image, group) re-reads its own slab
// (it is 2-32 KB: L1 / L2 hits) and writes y.  Saves a launch per GroupNorm where a launch costs more than the work.
__global__ __launch_bounds__(256) void tr_gn_fwd_fused_vec_kernel(const float* __restrict__ x, int npix, int C, int groups, float eps,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  int silu, float2* __restrict__ stats, float* __restrict__ y) {
    __shared__ double sh[4];
    __shared__ float2 sst;
    const int g = blockIdx.x, b = blockIdx.y, cpg = C / groups, Q = cpg >> 2;
    const int q = threadIdx.x % Q, pl = threadIdx.x / Q, step = 256 / Q;
    const size_t off = (size_t)b * npix * C + g * cpg + 4 * q;
    double s = 0.0, ss = 0.0;
    for (int p = pl; p < npix; p += step) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + off + (size_t)p * C);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s += v[e]; ss += (double)v[e] * v[e]; }
    }
    s = block_sum_d(s, sh);
    ss = block_sum_d(ss, sh);
    if (threadIdx.x == 0) {
        const double n = (double)npix * cpg, mean = s / n;
        double var = ss / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        sst = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
        stats[b * groups + g] = sst;
    }
    __syncthreads();
    const float2 st = sst;
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + g * cpg + 4 * q), be = *reinterpret_cast<const f32x4*>(beta + g * cpg + 4 * q);
    for (int p = pl; p < npix; p += step) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + off + (size_t)p * C);
        f32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float z = (v[e] - st.x) * st.y * ga[e] + be[e];
            out[e] = silu ? z * sigmoid_f(z) : z;
        }
        *reinterpret_cast<f32x4*>(y + off + (size_t)p * C) = out;
    }
}

// Large tensors: the one-block-per-(image, group) kernels above read 16-byte pieces 2 KB apart.  These variants give a
// block a slab of 64 pixels x all channels (fully coalesced float4 rows); a thread's channel quad is the same in every
// iteration when C divides 1024, so it accumulates in registers and ends with one LDS atomic per group it touched.
// acc [B][groups][2] doubles (zeroed by the caller): (sum, sum of squares).
__global__ __launch_bounds__(256) void tr_gn_stats_slab_kernel(const float* __restrict__ x, int npix, int C, int groups,
                                                               double* __restrict__ acc) {
    __shared__ double sS[64], sSS[64];
    const int b = blockIdx.y, cpg = C / groups;
    const int p0 = blockIdx.x * 64, p1 = min(p0 + 64, npix);
    if (threadIdx.x < groups) { sS[threadIdx.x] = 0.0; sSS[threadIdx.x] = 0.0; }
    __syncthreads();
    const float4* base = reinterpret_cast<const float4*>(x + ((size_t)b * npix + p0) * C);
    const int n4 = (p1 - p0) * C / 4;
    const int c0 = (threadIdx.x * 4) % C;                  // fixed channel quad (C divides 1024)
    float s[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 v = base[i];
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        ss[0] += v.x * v.x; ss[1] += v.y * v.y; ss[2] += v.z * v.z; ss[3] += v.w * v.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int g = (c0 + e) / cpg;
        atomicAdd(&sS[g], (double)s[e]);
        atomicAdd(&sSS[g], (double)ss[e]);
    }
    __syncthreads();
    if (threadIdx.x < groups) {
        unsafeAtomicAdd(acc + ((size_t)b * groups + threadIdx.x) * 2, sS[threadIdx.x]);
        unsafeAtomicAdd(acc + ((size_t)b * groups + threadIdx.x) * 2 + 1, sSS[threadIdx.x]);
    }
}

// (the finish kernels leave the accumulators zeroed for the next launch: no separate fill)
__global__ __launch_bounds__(256) void tr_gn_stats_finish_kernel(double* __restrict__ acc, int n, double count, float eps,
                                                                 float2* __restrict__ stats) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double s0 = acc[2 * i], s1 = acc[2 * i + 1];
    acc[2 * i] = 0.0;
    acc[2 * i + 1] = 0.0;
    const double mean = s0 / count;
    double var = s1 / count - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    stats[i] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
}


__global__ __launch_bounds__(256) void tr_gn_fwd_kernel(const float* __restrict__ x, const float2* __restrict__ stats,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, int npix,
                                                        int C, int groups, int silu, size_t total, float* __restrict__ y) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int b = (int)(i / ((size_t)npix * C));
    const float2 st = stats[b * groups + c / (C / groups)];
    const float z = (x[i] - st.x) * st.y * gamma[c] + beta[c];
    y[i] = silu ? z * sigmoid_f(z) : z;
}

// four channels per thread (channels per group a multiple of 4, fewer than 2^31 elements): 16-byte accesses, 32-bit index math
__global__ __launch_bounds__(256) void tr_gn_fwd_vec_kernel(const float* __restrict__ x, const float2* __restrict__ stats,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            unsigned per_image, unsigned C, unsigned cpg, int groups, int silu,
                                                            unsigned total4, float* __restrict__ y) {
    const unsigned i4 = blockIdx.x * 256 + threadIdx.x;
    if (i4 >= total4) return;
    const unsigned i = i4 * 4, c = i % C, b = i / per_image;
    const float2 st = stats[b * groups + c / cpg];
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + i);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 out;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float z = (xv[e] - st.x) * st.y * ga[e] + be[e];
        out[e] = silu ? z * sigmoid_f(z) : z;
    }
    *reinterpret_cast<f32x4*>(y + i) = out;
}

__global__ __launch_bounds__(256) void tr_gn_bwd_apply_vec_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  const float2* __restrict__ stats, const float2* __restrict__ sums,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  unsigned per_image, unsigned C, unsigned cpg, int groups, int silu,
                                                                  int accumulate, float inv_n, unsigned total4, float* __restrict__ dx) {
    const unsigned i4 = blockIdx.x * 256 + threadIdx.x;
    if (i4 >= total4) return;
    const unsigned i = i4 * 4, c = i % C, b = i / per_image;
    const int sg_i = b * groups + c / cpg;
    const float2 st = stats[sg_i], sm = sums[sg_i];
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + i), dv = *reinterpret_cast<const f32x4*>(dy + i);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 out;
    if (accumulate) out = *reinterpret_cast<const f32x4*>(dx + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xh = (xv[e] - st.x) * st.y;
        float dz = dv[e];
        if (silu) {
            const float z = xh * ga[e] + be[e], sg = sigmoid_f(z);
            dz *= sg * (1.f + z * (1.f - sg));
        }
        const float v = st.y * (dz * ga[e] - sm.x * inv_n - xh * sm.y * inv_n);
        out[e] = accumulate ? out[e] + v : v;
    }
    *reinterpret_cast<f32x4*>(dx + i) = out;
}

// grid (groups, B): sums[b][g] = (sum dz gamma, sum dz gamma xhat); dgamma[c] += sum dz xhat, dbeta[c] += sum dz
// (dz = dy * act'(z)); the first T = 256 - 256 % cpg threads stride by T, so thread t always meets channel t % cpg
__global__ __launch_bounds__(256) void tr_gn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               const float2* __restrict__ stats, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int npix, int C, int groups, int silu,
                                                               float2* __restrict__ sums, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta) {
    __shared__ double sh[4];
    __shared__ float shg[256], shb[256];
    const int g = blockIdx.x, b = blockIdx.y, cpg = C / groups;
    const float2 st = stats[b * groups + g];
    const size_t base = (size_t)b * npix * C + g * cpg;
    const int ci = threadIdx.x % cpg, c = g * cpg + ci;
    const int TT = 256 - 256 % cpg;
    const float ga = gamma[c], be = beta[c];
    double s1 = 0.0, s2 = 0.0;
    float dg = 0.f, db = 0.f;
    for (int e = threadIdx.x < TT ? threadIdx.x : npix * cpg; e < npix * cpg; e += TT) {
        const size_t idx = base + (size_t)(e / cpg) * C + ci;
        const float xh = (x[idx] - st.x) * st.y;
        float dz = dy[idx];
        if (silu) {
            const float z = xh * ga + be, sg = sigmoid_f(z);
            dz *= sg * (1.f + z * (1.f - sg));
        }
        s1 += (double)(dz * ga);
        s2 += (double)(dz * ga * xh);
        dg += dz * xh;
        db += dz;
    }
    s1 = block_sum_d(s1, sh);
    s2 = block_sum_d(s2, sh);
    shg[threadIdx.x] = dg;
    shb[threadIdx.x] = db;
    __syncthreads();
    if (threadIdx.x < cpg) {
        float tg = 0.f, tb = 0.f;
        for (int t = threadIdx.x; t < TT; t += cpg) { tg += shg[t]; tb += shb[t]; }
        unsafeAtomicAdd(dgamma + g * cpg + threadIdx.x, tg);
        unsafeAtomicAdd(dbeta + g * cpg + threadIdx.x, tb);
    }
    if (threadIdx.x == 0) sums[b * groups + g] = make_float2((float)s1, (float)s2);
}

__global__ __launch_bounds__(256) void tr_gn_bwd_reduce_vec_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   const float2* __restrict__ stats, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, int npix, int C, int groups, int silu,
                                                                   float2* __restrict__ sums, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta) {
    __shared__ double sh[4];
    __shared__ float shg[4][256], shb[4][256];
    const int g = blockIdx.x, b = blockIdx.y, cpg = C / groups, Q = cpg >> 2;
    const int q = threadIdx.x % Q, pl = threadIdx.x / Q, step = 256 / Q;
    const float2 st = stats[b * groups + g];
    const size_t base = (size_t)b * npix * C + g * cpg + 4 * q;
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + g * cpg + 4 * q), be = *reinterpret_cast<const f32x4*>(beta + g * cpg + 4 * q);
    double s1 = 0.0, s2 = 0.0;
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = pl; p < npix; p += step) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + base + (size_t)p * C);
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dy + base + (size_t)p * C);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xv[e] - st.x) * st.y;
            float dz = dv[e];
            if (silu) {
                const float z = xh * ga[e] + be[e], sg = sigmoid_f(z);
                dz *= sg * (1.f + z * (1.f - sg));
            }
            s1 += (double)(dz * ga[e]);
            s2 += (double)(dz * ga[e] * xh);
            dg[e] += dz * xh;
            db[e] += dz;
        }
    }
    s1 = block_sum_d(s1, sh);
    s2 = block_sum_d(s2, sh);
#pragma unroll
    for (int e = 0; e < 4; ++e) { shg[e][threadIdx.x] = dg[e]; shb[e][threadIdx.x] = db[e]; }
    __syncthreads();
    if (threadIdx.x < cpg) {                                          // channel ci = 4 q' + e: threads t = q' (mod Q)
        const int qq = threadIdx.x >> 2, e = threadIdx.x & 3;
        float tg = 0.f, tb = 0.f;
        for (int t = qq; t < 256; t += Q) { tg += shg[e][t]; tb += shb[e][t]; }
        unsafeAtomicAdd(dgamma + g * cpg + threadIdx.x, tg);
        unsafeAtomicAdd(dbeta + g * cpg + threadIdx.x, tb);
    }
    if (threadIdx.x == 0) sums[b * groups + g] = make_float2((float)s1, (float)s2);
}

// slab variant of the reduction (see tr_gn_stats_slab_kernel): acc [B][groups][2] doubles = (sum dz gamma, sum dz gamma xhat);
// dgamma / dbeta by one global atomic per channel and block
__global__ __launch_bounds__(256) void tr_gn_bwd_reduce_slab_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                    const float2* __restrict__ stats, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, int npix, int C, int groups,
                                                                    int silu, double* __restrict__ acc, float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta) {
    __shared__ double s1[64], s2[64];
    __shared__ float sg[1024], sb[1024];
    const int b = blockIdx.y, cpg = C / groups;
    const int SP = gridDim.z ? (npix + gridDim.x - 1) / gridDim.x : 64;   // pixels per block (the launcher's slab size)
    const int p0 = blockIdx.x * SP, p1 = min(p0 + SP, npix);
    if (threadIdx.x < groups) { s1[threadIdx.x] = 0.0; s2[threadIdx.x] = 0.0; }
    for (int c = threadIdx.x; c < C; c += 256) { sg[c] = 0.f; sb[c] = 0.f; }
    __syncthreads();
    const size_t off = ((size_t)b * npix + p0) * C;
    const float4* bx = reinterpret_cast<const float4*>(x + off);
    const float4* bd = reinterpret_cast<const float4*>(dy + off);
    const int n4 = (p1 - p0) * C / 4;
    const int c0 = (threadIdx.x * 4) % C;
    float ga[4], be[4], mean[4], rstd[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float2 st = stats[b * groups + (c0 + e) / cpg];
        ga[e] = gamma[c0 + e]; be[e] = beta[c0 + e]; mean[e] = st.x; rstd[e] = st.y;
    }
    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f}, dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 xv = bx[i], dv = bd[i];
        const float xe[4] = {xv.x, xv.y, xv.z, xv.w}, de[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xe[e] - mean[e]) * rstd[e];
            float dz = de[e];
            if (silu) {
                const float z = xh * ga[e] + be[e], sgm = sigmoid_f(z);
                dz *= sgm * (1.f + z * (1.f - sgm));
            }
            a1[e] += dz * ga[e];
            a2[e] += dz * ga[e] * xh;
            dg[e] += dz * xh;
            db[e] += dz;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int g = (c0 + e) / cpg;
        atomicAdd(&s1[g], (double)a1[e]);
        atomicAdd(&s2[g], (double)a2[e]);
        atomicAdd(&sg[c0 + e], dg[e]);
        atomicAdd(&sb[c0 + e], db[e]);
    }
    __syncthreads();
    if (threadIdx.x < groups) {
        unsafeAtomicAdd(acc + ((size_t)b * groups + threadIdx.x) * 2, s1[threadIdx.x]);
        unsafeAtomicAdd(acc + ((size_t)b * groups + threadIdx.x) * 2 + 1, s2[threadIdx.x]);
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        unsafeAtomicAdd(dgamma + c, sg[c]);
        unsafeAtomicAdd(dbeta + c, sb[c]);
    }
}

__global__ __launch_bounds__(256) void tr_gn_sums_finish_kernel(double* __restrict__ acc, int n, float2* __restrict__ sums) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    sums[i] = make_float2((float)acc[2 * i], (float)acc[2 * i + 1]);
    acc[2 * i] = 0.0;
    acc[2 * i + 1] = 0.0;
}

__global__ __launch_bounds__(256) void tr_gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              const float2* __restrict__ stats, const float2* __restrict__ sums,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int npix, int C, int groups, int silu, int accumulate, size_t total,
                                                              float* __restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C), cpg = C / groups;
    const int b = (int)(i / ((size_t)npix * C));
    const int sg_i = b * groups + c / cpg;
    const float2 st = stats[sg_i], sm = sums[sg_i];
    const float xh = (x[i] - st.x) * st.y;
    float dz = dy[i];
    if (silu) {
        const float z = xh * gamma[c] + beta[c], sg = sigmoid_f(z);
        dz *= sg * (1.f + z * (1.f - sg));
    }
    const float inv_n = 1.f / ((float)npix * cpg);
    const float v = st.y * (dz * gamma[c] - sm.x * inv_n - xh * sm.y * inv_n);
    dx[i] = accumulate ? dx[i] + v : v;
}

// ---- (round 5) per-channel statistics and the GroupNorm backward apply of the fused tape -------------------------------------
// cs[b][c] += (sum, sum of squares) over a slab of image b's pixels: for tensors no fused conv produced (conv_in's output).
// grid (slabs, B); thread (pixel lane t / Q, channel quad t % Q), Q = C / 4 <= 256.
__global__ __launch_bounds__(256) void tr_chan_stats_kernel(const float* __restrict__ x, int npix, int C, int slab, float* __restrict__ cs) {
    __shared__ float sacc[2 * 1024];
    const int b = blockIdx.y, Q = C >> 2, step = 256 / Q;
    const int q = threadIdx.x % Q, pl = threadIdx.x / Q;
    for (int e = threadIdx.x; e < 2 * C; e += 256) sacc[e] = 0.f;
    __syncthreads();
    if (pl < step) {
        const int p0 = blockIdx.x * slab, p1 = min(p0 + slab, npix);
        const float* base = x + (size_t)b * npix * C + 4 * q;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int px = p0 + pl; px < p1; px += step) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)px * C);
            s += v;
            ss += v * v;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(&sacc[(4 * q + e) * 2], s[e]);
            atomicAdd(&sacc[(4 * q + e) * 2 + 1], ss[e]);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += 256) unsafeAtomicAdd(cs + (size_t)b * C * 2 + e, sacc[e]);
}

// dx = rstd (gamma dz - s1 / n - xhat s2 / n) [+ res] of a GroupNorm over cat(x0, x1), from dz = dy act'(z) (stored by the data-gradient
// conv's epilogue) and gs[b][c] = (sum dz, sum dz xhat): s1 = sum over the group of gamma gs.x, s2 = ... gamma gs.y.  The result is
// written (or added) to the two sources' gradients separately; d gamma / d beta are finished by block (0, 0).  grid (slabs, B).
struct TrGnApply {
    const float* dz; const float* x0; const float* x1; const float* cs0; const float* cs1; const float* gs; const float* gamma;
    const float* res; float* dx0; float* dx1; float* dgamma; float* dbeta;
    int B, npix, C, C0, groups, slab, acc0, acc1; float eps;
};
__global__ __launch_bounds__(256) void tr_gn_bwd_apply2_kernel(const TrGnApply a) {
    __shared__ float4 sT[768];                  // per channel: (rstd gamma, rstd s1 / n, rstd s2 / n, -)
    __shared__ float2 sM[768];                  // per channel: (mean, rstd)
    __shared__ float4 sG[64];                   // per group: (mean, rstd, s1, s2)
    const int tid = threadIdx.x, b = blockIdx.y, C = a.C, C0 = a.C0, C1 = C - C0, cpg = C / a.groups;
    for (int c = tid; c < C; c += 256) {
        const float2 v = c < C0 ? reinterpret_cast<const float2*>(a.cs0)[(size_t)b * C0 + c] : reinterpret_cast<const float2*>(a.cs1)[(size_t)b * C1 + c - C0];
        const float2 g = reinterpret_cast<const float2*>(a.gs)[(size_t)b * C + c];
        const float ga = a.gamma[c];
        sM[c] = v;
        sT[c] = make_float4(ga, g.x * ga, g.y * ga, 0.f);
    }
    __syncthreads();
    const double n = (double)a.npix * cpg;
    for (int g = tid; g < a.groups; g += 256) {
        double s = 0.0, ss = 0.0, s1 = 0.0, s2 = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { s += (double)sM[c].x; ss += (double)sM[c].y; s1 += (double)sT[c].y; s2 += (double)sT[c].z; }
        const double mean = s / n;
        double var = ss / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        sG[g] = make_float4((float)mean, (float)(1.0 / sqrt(var + (double)a.eps)), (float)(s1 / n), (float)(s2 / n));
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const float4 g = sG[c / cpg];
        const float ga = sT[c].x;
        sM[c] = make_float2(g.x, g.y);
        sT[c] = make_float4(g.y * ga, g.y * g.z, g.y * g.w, 0.f);
    }
    __syncthreads();
    if (blockIdx.x == 0 && b == 0 && a.dgamma)
        for (int c = tid; c < C; c += 256) {
            float dg = 0.f, db = 0.f;
            for (int bi = 0; bi < a.B; ++bi) {
                const float2 g = reinterpret_cast<const float2*>(a.gs)[(size_t)bi * C + c];
                db += g.x; dg += g.y;
            }
            a.dgamma[c] += dg;
            a.dbeta[c] += db;
        }
    const int p0 = blockIdx.x * a.slab, p1 = min(p0 + a.slab, a.npix);
    const int Q = C >> 2, n4 = (p1 - p0) * Q;
    const size_t pix0 = (size_t)b * a.npix + p0;
    for (int i = tid; i < n4; i += 256) {
        const int pl = i / Q, c = (i - pl * Q) * 4;
        const size_t px = pix0 + pl;
        const f32x4 dz = *reinterpret_cast<const f32x4*>(a.dz + px * C + c);
        const bool first = c < C0;
        const size_t off = first ? px * C0 + c : px * C1 + (c - C0);
        const f32x4 xv = *reinterpret_cast<const f32x4*>((first ? a.x0 : a.x1) + off);
        f32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float4 t = sT[c + e];
            const float2 m = sM[c + e];
            out[e] = t.x * dz[e] - t.y - (xv[e] - m.x) * m.y * t.z;
        }
        if (a.res) out += *reinterpret_cast<const f32x4*>(a.res + px * C + c);
        float* dst = (first ? a.dx0 : a.dx1) + off;
        if (first ? a.acc0 : a.acc1) out += *reinterpret_cast<const f32x4*>(dst);
        *reinterpret_cast<f32x4*>(dst) = out;
    }
}

// ---- attention, head_dim 8 -----------------------------------------------------------------------------------------------
// q, k, v, o: [B][L][C] fp32, head h = channels 8h .. 8h + 8.  grid (ceil(L / 128), heads, B), 128 threads = 128 queries.
// The other side (keys / values, or queries / dO) passes through LDS in tiles of ATT_TK rows (16-18 KB: ten workgroups per
// CU instead of two with the whole head resident), every thread reads the same row at a time (LDS broadcast).
constexpr int ATT_TK = 256;

__global__ __launch_bounds__(128) void tr_attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, int L, int C, float scale,
                                                          float* __restrict__ o, float* __restrict__ lse) {
    __shared__ __attribute__((aligned(16))) float sK[ATT_TK * 8];
    __shared__ __attribute__((aligned(16))) float sV[ATT_TK * 8];
    const int h = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
    const size_t base = (size_t)b * L * C + h * 8;
    const int i = blockIdx.x * 128 + threadIdx.x;
    const bool live = i < L;
    float qi[8], acc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) { qi[d] = live ? q[base + (size_t)i * C + d] * scale : 0.f; acc[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    for (int j0 = 0; j0 < L; j0 += ATT_TK) {
        const int nt = min(ATT_TK, L - j0);
        __syncthreads();
        for (int e = threadIdx.x; e < nt * 8; e += 128) {
            sK[e] = k[base + (size_t)(j0 + (e >> 3)) * C + (e & 7)];
            sV[e] = v[base + (size_t)(j0 + (e >> 3)) * C + (e & 7)];
        }
        __syncthreads();
        // groups of 8 keys: one running-maximum update (one rescale of the accumulator) per group
        for (int jj = 0; jj < nt; jj += 8) {
            float sc[8];
            float gm = m;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float t = -INFINITY;
                if (jj + u < nt) {
                    t = 0.f;
#pragma unroll
                    for (int d = 0; d < 8; ++d) t += qi[d] * sK[(jj + u) * 8 + d];
                }
                sc[u] = t;
                gm = fmaxf(gm, t);
            }
            const float corr = __expf(m - gm);
            l *= corr;
#pragma unroll
            for (int d = 0; d < 8; ++d) acc[d] *= corr;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (jj + u < nt) {
                    const float pj = __expf(sc[u] - gm);
                    l += pj;
#pragma unroll
                    for (int d = 0; d < 8; ++d) acc[d] += pj * sV[(jj + u) * 8 + d];
                }
            }
            m = gm;
        }
    }
    if (!live) return;
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < 8; ++d) o[base + (size_t)i * C + d] = acc[d] * inv;
    lse[((size_t)b * heads + h) * L + i] = m + __logf(l);
}

// dq and delta_i = dO_i . O_i ; one thread per query
__global__ __launch_bounds__(128) void tr_attn_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, const float* __restrict__ o,
                                                             const float* __restrict__ dO, const float* __restrict__ lse, int L, int C,
                                                             float scale, float* __restrict__ dq, float* __restrict__ delta) {
    __shared__ __attribute__((aligned(16))) float sK[ATT_TK * 8];
    __shared__ __attribute__((aligned(16))) float sV[ATT_TK * 8];
    const int h = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
    const size_t base = (size_t)b * L * C + h * 8;
    const int i = blockIdx.x * 128 + threadIdx.x;
    const bool live = i < L;
    float qi[8], doi[8], acc[8];
    float D = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        qi[d] = live ? q[base + (size_t)i * C + d] * scale : 0.f;
        doi[d] = live ? dO[base + (size_t)i * C + d] : 0.f;
        D += live ? doi[d] * o[base + (size_t)i * C + d] : 0.f;
        acc[d] = 0.f;
    }
    const float li = live ? lse[((size_t)b * heads + h) * L + i] : 0.f;
    for (int j0 = 0; j0 < L; j0 += ATT_TK) {
        const int nt = min(ATT_TK, L - j0);
        __syncthreads();
        for (int e = threadIdx.x; e < nt * 8; e += 128) {
            sK[e] = k[base + (size_t)(j0 + (e >> 3)) * C + (e & 7)];
            sV[e] = v[base + (size_t)(j0 + (e >> 3)) * C + (e & 7)];
        }
        __syncthreads();
        for (int j = 0; j < nt; ++j) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) { s += qi[d] * sK[j * 8 + d]; dp += doi[d] * sV[j * 8 + d]; }
            const float ds = __expf(s - li) * (dp - D);
#pragma unroll
            for (int d = 0; d < 8; ++d) acc[d] += ds * sK[j * 8 + d];
        }
    }
    if (!live) return;
#pragma unroll
    for (int d = 0; d < 8; ++d) dq[base + (size_t)i * C + d] = acc[d] * scale;
    delta[((size_t)b * heads + h) * L + i] = D;
}

// dk, dv: one thread per key against all queries (Q * scale, dO, lse, delta pass through LDS in tiles)
__global__ __launch_bounds__(128) void tr_attn_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, const float* __restrict__ dO,
                                                              const float* __restrict__ lse, const float* __restrict__ delta, int L,
                                                              int C, float scale, float* __restrict__ dk, float* __restrict__ dv) {
    __shared__ __attribute__((aligned(16))) float sQ[ATT_TK * 8];
    __shared__ __attribute__((aligned(16))) float sDO[ATT_TK * 8];
    __shared__ float sL[ATT_TK];
    __shared__ float sD[ATT_TK];
    const int h = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
    const size_t base = (size_t)b * L * C + h * 8;
    const size_t rbase = ((size_t)b * heads + h) * L;
    const int j = blockIdx.x * 128 + threadIdx.x;
    const bool live = j < L;
    float kj[8], vj[8], ak[8], av[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        kj[d] = live ? k[base + (size_t)j * C + d] : 0.f;
        vj[d] = live ? v[base + (size_t)j * C + d] : 0.f;
        ak[d] = 0.f;
        av[d] = 0.f;
    }
    for (int i0 = 0; i0 < L; i0 += ATT_TK) {
        const int nt = min(ATT_TK, L - i0);
        __syncthreads();
        for (int e = threadIdx.x; e < nt * 8; e += 128) {
            sQ[e] = q[base + (size_t)(i0 + (e >> 3)) * C + (e & 7)] * scale;
            sDO[e] = dO[base + (size_t)(i0 + (e >> 3)) * C + (e & 7)];
        }
        for (int e = threadIdx.x; e < nt; e += 128) {
            sL[e] = lse[rbase + i0 + e];
            sD[e] = delta[rbase + i0 + e];
        }
        __syncthreads();
        for (int i = 0; i < nt; ++i) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) { s += sQ[i * 8 + d] * kj[d]; dp += sDO[i * 8 + d] * vj[d]; }
            const float pij = __expf(s - sL[i]);
            const float ds = pij * (dp - sD[i]);
#pragma unroll
            for (int d = 0; d < 8; ++d) { av[d] += pij * sDO[i * 8 + d]; ak[d] += ds * sQ[i * 8 + d]; }
        }
    }
    if (!live) return;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        dk[base + (size_t)j * C + d] = ak[d];          // sQ already carries the scale
        dv[base + (size_t)j * C + d] = av[d];
    }
}

// ---- elementwise ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tr_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                                     size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = a[i] + b[i];
}

// dst[p][dst_off + c] (+)= src[p][src_off + c], c < ncopy  (concat, split, accumulate a gradient slice)
__global__ __launch_bounds__(256) void tr_copy_channels_kernel(const float* __restrict__ src, int src_ld, int src_off,
                                                               float* __restrict__ dst, int dst_ld, int dst_off, int ncopy,
                                                               size_t npix, int accumulate) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix * ncopy) return;
    const size_t px = i / ncopy;
    const int c = (int)(i % ncopy);
    float* d = dst + px * dst_ld + dst_off + c;
    const float v = src[px * src_ld + src_off + c];
    *d = accumulate ? *d + v : v;
}

__global__ __launch_bounds__(256) void tr_copy_channels_vec_kernel(const float* __restrict__ src, unsigned src_ld, unsigned src_off,
                                                                   float* __restrict__ dst, unsigned dst_ld, unsigned dst_off,
                                                                   unsigned ncopy4, unsigned total4, int accumulate) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const unsigned px = i / ncopy4, c = (i % ncopy4) * 4;
    f32x4* d = reinterpret_cast<f32x4*>(dst + (size_t)px * dst_ld + dst_off + c);
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)px * src_ld + src_off + c);
    *d = accumulate ? *d + v : v;
}

// nearest-x2 backward: dx[b][w][h][c] = sum of the 2 x 2 block of du[b][2w..][2h..][c]
__global__ __launch_bounds__(256) void tr_sum2x2_kernel(const float* __restrict__ du, int B, int W, int H, int C,
                                                        float* __restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * W * H * C) return;
    const int c = (int)(i % C);
    size_t t = i / C;
    const int h = (int)(t % H);
    t /= H;
    const int w = (int)(t % W), b = (int)(t / W);
    const size_t r0 = (((size_t)b * 2 * W + 2 * w) * 2 * H + 2 * h) * C + c, r1 = r0 + (size_t)2 * H * C;
    dx[i] = (du[r0] + du[r0 + C]) + (du[r1] + du[r1 + C]);
}

__global__ __launch_bounds__(256) void tr_silu_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ y,
                                                      size_t n, int backward, int accumulate) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float z = x[i], sg = sigmoid_f(z);
    float v = backward ? dy[i] * sg * (1.f + z * (1.f - sg)) : z * sg;
    if (accumulate) v += y[i];
    y[i] = v;
}

// diffusers Timesteps(dim, flip_sin_to_cos=True, shift 0): e = [cos(t f_i), sin(t f_i)], f_i = exp(-ln(1e4) i / half)
__global__ __launch_bounds__(256) void tr_timestep_embed_kernel(const long long* __restrict__ t, int B, int dim, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * dim) return;
    const int b = i / dim, j = i % dim, half = dim / 2;
    const float f = expf(-9.210340371976184f * (float)(j % half) / (float)half);
    const float a = (float)t[b] * f;
    out[i] = j < half ? cosf(a) : sinf(a);
}

// NCHW latents (+ constant pos-encoding channel: 1 at azimuth 0) -> NHWC [B][W][H][C + pos]
__global__ __launch_bounds__(256) void tr_pack_input_kernel(const float* __restrict__ x, int B, int C, int W, int H, int pos,
                                                            float* __restrict__ y) {
    const int Co = C + pos;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * W * H * Co) return;
    const int c = (int)(i % Co);
    size_t t = i / Co;
    const int h = (int)(t % H);
    t /= H;
    const int w = (int)(t % W), b = (int)(t / W);
    y[i] = c < C ? x[(((size_t)b * C + c) * W + w) * H + h] : (w == 0 ? 1.f : 0.f);
}

// NHWC -> NCHW (model_output for the caller)
__global__ __launch_bounds__(256) void tr_unpack_kernel(const float* __restrict__ x, int B, int C, int W, int H, float* __restrict__ y) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * W * H * C) return;
    const int h = (int)(i % H);
    size_t t = i / H;
    const int w = (int)(t % W);
    t /= W;
    const int c = (int)(t % C), b = (int)(t / C);
    y[i] = x[(((size_t)b * W + w) * H + h) * C + c];
}

// loss = mean_b( weight[b] * mean_{c,w,h} (pred - target)^2 ); dpred = weight[b] * 2 (pred - target) / (B * C * W * H)
// pred NHWC, target NCHW; loss accumulated in fp64
__global__ __launch_bounds__(256) void tr_mse_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                     const float* __restrict__ weight, int B, int C, int W, int H,
                                                     float* __restrict__ dpred, double* __restrict__ loss) {
    __shared__ double sh[4];
    const size_t total = (size_t)B * W * H * C;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    double part = 0.0;
    if (i < total) {
        const int c = (int)(i % C);
        size_t t = i / C;
        const int h = (int)(t % H);
        t /= H;
        const int w = (int)(t % W), b = (int)(t / W);
        const float d = pred[i] - target[(((size_t)b * C + c) * W + w) * H + h];
        const float wb = weight ? weight[b] : 1.f;
        dpred[i] = wb * 2.f * d / (float)total;
        part = (double)wb * (double)d * (double)d / (double)total;
    }
    part = block_sum_d(part, sh);
    if (threadIdx.x == 0) unsafeAtomicAdd(loss, part);
}

__global__ __launch_bounds__(256) void tr_sqnorm_kernel(const float* __restrict__ g, size_t n, double* __restrict__ out) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += (double)g[i] * g[i];
    acc = block_sum_d(acc, sh);
    if (threadIdx.x == 0) unsafeAtomicAdd(out, acc);
}

// torch.optim.AdamW step on flat buffers with the clip_grad_norm_ coefficient folded in, then diffusers EMAModel.step
struct TrAdam {
    float* p; float* g; float* m; float* v; float* ema; const double* sqnorm;
    size_t n; float lr, b1, b2, eps, wd, bc1, bc2, max_norm, ema_decay;
    const float* dyn;          // device [4] = (lr, bias correction 1, bias correction 2, ema decay) or null: the fields above
    int zero_grads;            // also clear the gradient buffer (saves a separate fill pass)
};
__global__ __launch_bounds__(256) void tr_adamw_kernel(const TrAdam a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    float coef = 1.f;
    if (a.sqnorm && a.max_norm > 0.f) {
        const float tn = (float)sqrt(*a.sqnorm);
        coef = fminf(a.max_norm / (tn + 1e-6f), 1.f);
    }
    const float lr = a.dyn ? a.dyn[0] : a.lr, bc1 = a.dyn ? a.dyn[1] : a.bc1, bc2 = a.dyn ? a.dyn[2] : a.bc2;
    const float ema_decay = a.dyn ? a.dyn[3] : a.ema_decay;
    const float g = a.g[i] * coef;
    if (a.zero_grads) a.g[i] = 0.f;
    float w = a.p[i] * (1.f - lr * a.wd);
    const float m = a.b1 * a.m[i] + (1.f - a.b1) * g;
    const float v = a.b2 * a.v[i] + (1.f - a.b2) * g * g;
    a.m[i] = m;
    a.v[i] = v;
    const float denom = sqrtf(v) / sqrtf(bc2) + a.eps;
    w -= (lr / bc1) * (m / denom);
    a.p[i] = w;
    if (a.ema) a.ema[i] -= (1.f - ema_decay) * (a.ema[i] - w);
}

// the per-step scalars of the optimizer from a DEVICE step counter, so that a captured step graph advances by itself:
// step = ++*counter; lr = diffusers "cosine" schedule with warmup at (step - 1) scheduler steps; AdamW bias corrections;
// EMAModel.get_decay(step) with use_ema_warmup (ldm/train_unconditional.py:320-329,394-399,546-556)
__global__ void tr_hyper_kernel(long long* __restrict__ counter, const rldm_hyper_config c, float* __restrict__ dyn) {
    if (threadIdx.x || blockIdx.x) return;
    const long long step = *counter + 1;
    *counter = step;
    const long long k = step - 1;
    double lr;
    if (k < c.lr_warmup_steps) lr = (double)c.lr * (double)k / (double)(c.lr_warmup_steps > 1 ? c.lr_warmup_steps : 1);
    else {
        const long long span = c.total_steps - c.lr_warmup_steps;
        const double progress = (double)(k - c.lr_warmup_steps) / (double)(span > 1 ? span : 1);
        const double f = 0.5 * (1.0 + cos(3.14159265358979323846 * progress));
        lr = (double)c.lr * (f > 0.0 ? f : 0.0);
    }
    dyn[0] = (float)lr;
    dyn[1] = (float)(1.0 - pow((double)c.beta1, (double)step));
    dyn[2] = (float)(1.0 - pow((double)c.beta2, (double)step));
    double dec = 0.0;
    const long long es = step - 1 > 0 ? step - 1 : 0;
    if (es > 0) {
        dec = 1.0 - pow(1.0 + (double)es / (double)c.ema_inv_gamma, -(double)c.ema_power);
        dec = dec < (double)c.ema_max_decay ? dec : (double)c.ema_max_decay;
        dec = dec > 0.0 ? dec : 0.0;
    }
    dyn[3] = (float)dec;
}

// master fp32 [N][Cin][taps] -> forward copy bf16 [N][taps][Cin_pad] and data-gradient copy bf16 [Cin][taps][N_pad]
// (tap flipped: 8 - t; 1x1: plain transpose); pads are zero
__global__ __launch_bounds__(256) void tr_pack_weights_kernel(const float* __restrict__ w, int N, int Cin, int taps, int Cin_pad,
                                                              int N_pad, bf16_t* __restrict__ wf, bf16_t* __restrict__ wt) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t nf = (size_t)N * taps * Cin_pad, nt = (size_t)Cin * taps * N_pad;
    if (i < nf) {
        const int c = (int)(i % Cin_pad);
        const int t = (int)((i / Cin_pad) % taps), n = (int)(i / ((size_t)Cin_pad * taps));
        wf[i] = c < Cin ? rldm::f32_to_bf16(w[((size_t)n * Cin + c) * taps + t]) : (bf16_t)0;
    }
    if (wt && i < nt) {
        const int n = (int)(i % N_pad);
        const int t = (int)((i / N_pad) % taps), c = (int)(i / ((size_t)N_pad * taps));
        wt[i] = n < N ? rldm::f32_to_bf16(w[((size_t)n * Cin + c) * taps + (taps - 1 - t)]) : (bf16_t)0;
    }
}

// every conv / linear weight of the model in ONE launch: thread i finds its layer by bisection over the cumulative element
// counts (elements of a layer = max(forward copy, transposed copy))
__global__ __launch_bounds__(256) void tr_pack_all_kernel(const float* __restrict__ params, const rldm_pack_desc* __restrict__ d,
                                                          int nlayers, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int lo = 0, hi = nlayers - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (d[mid].first <= i) lo = mid; else hi = mid - 1;
    }
    const rldm_pack_desc L = d[lo];
    const long long e = i - L.first;
    const float* w = params + L.param_offset;
    const int Cin_pad = (L.Cin + 15) / 16 * 16, N_pad = (L.N + 15) / 16 * 16;
    const long long nf = (long long)L.N * L.taps * Cin_pad, nt = (long long)L.Cin * L.taps * N_pad;
    if (e < nf) {
        const int c = (int)(e % Cin_pad);
        const int t = (int)((e / Cin_pad) % L.taps), n = (int)(e / ((long long)Cin_pad * L.taps));
        static_cast<bf16_t*>(L.w_forward)[e] = c < L.Cin ? rldm::f32_to_bf16(w[((size_t)n * L.Cin + c) * L.taps + t]) : (bf16_t)0;
    }
    if (L.w_transposed && e < nt) {
        const int n = (int)(e % N_pad);
        const int t = (int)((e / N_pad) % L.taps), c = (int)(e / ((long long)N_pad * L.taps));
        static_cast<bf16_t*>(L.w_transposed)[e] =
            n < L.N ? rldm::f32_to_bf16(w[((size_t)n * L.Cin + c) * L.taps + (L.taps - 1 - t)]) : (bf16_t)0;
    }
}

// zero fill as a kernel: inside a captured graph, hipMemsetAsync nodes over these (large, pool-allocated) outputs replayed
// wrongly from the second replay on (garbage gradients; tools/determinism_probe.py), a kernel node does not
__global__ __launch_bounds__(256) void tr_zero_kernel(float* __restrict__ y, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) *reinterpret_cast<float4*>(y + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    else
        for (size_t e = i; e < n; ++e) y[e] = 0.f;
}

__global__ __launch_bounds__(256) void tr_zero2d_kernel(float* __restrict__ y, int ld, int N, int B) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)B * N) y[(i / N) * ld + i % N] = 0.f;
}

// ---- Linear layers on a handful of rows (time embedding MLP, the resnets' time_emb_proj: B <= 16 rows) -------------------------
// The MFMA conv kernel gives such a layer 4 workgroups that walk K serially (19 us for K = 512).  Here one wave owns one output
// feature: lanes stride over K in 16-byte pieces of the bf16 weight row, fp32 FMAs against the rows of x, butterfly at the end.
constexpr int LIN_MAXB = 16;
__global__ __launch_bounds__(256) void tr_linear_rows_kernel(const float* __restrict__ x, int ldx, const bf16_t* __restrict__ w, int Kp,
                                                             int K, const float* __restrict__ bias, float* __restrict__ y, int ldy,
                                                             int B, int N, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[LIN_MAXB];
#pragma unroll
    for (int b = 0; b < LIN_MAXB; ++b) acc[b] = 0.f;
    const bf16_t* wr = w + (size_t)n * Kp;
    for (int k0 = lane * 8; k0 < K; k0 += 512) {
        const uint4 wv = *reinterpret_cast<const uint4*>(wr + k0);
        const float wf[8] = {rldm::bf16lo(wv.x), rldm::bf16hi(wv.x), rldm::bf16lo(wv.y), rldm::bf16hi(wv.y),
                             rldm::bf16lo(wv.z), rldm::bf16hi(wv.z), rldm::bf16lo(wv.w), rldm::bf16hi(wv.w)};
#pragma unroll
        for (int b = 0; b < LIN_MAXB; ++b) {
            if (b < B) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(x + (size_t)b * ldx + k0);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(x + (size_t)b * ldx + k0 + 4);
                acc[b] += a0[0] * wf[0] + a0[1] * wf[1] + a0[2] * wf[2] + a0[3] * wf[3] + a1[0] * wf[4] + a1[1] * wf[5] + a1[2] * wf[6] +
                          a1[3] * wf[7];
            }
        }
    }
#pragma unroll
    for (int b = 0; b < LIN_MAXB; ++b) {
        if (b < B) {
            float v = acc[b];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
            if (lane == 0) {
                float* dst = y + (size_t)b * ldy + n;
                v += bias ? bias[n] : 0.f;
                *dst = accumulate ? *dst + v : v;
            }
        }
    }
}

// dw[n][k] += sum_b dy[b][n] x[b][k] (fp32 master gradient, row-major [N][K]); dbias[n] += sum_b dy[b][n]
__global__ __launch_bounds__(256) void tr_linear_rows_wgrad_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ x,
                                                                   int ldx, int B, int N, int K, float* __restrict__ dw,
                                                                   float* __restrict__ dbias) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int kq = K >> 2;
    if (i >= (size_t)N * kq) return;
    const int n = (int)(i / kq), k = (int)(i % kq) * 4;
    f32x4 acc = *reinterpret_cast<const f32x4*>(dw + (size_t)n * K + k);
    float sb = 0.f;
    for (int b = 0; b < B; ++b) {
        const float g = dy[(size_t)b * ldy + n];
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)b * ldx + k);
        acc += g * xv;
        sb += g;
    }
    *reinterpret_cast<f32x4*>(dw + (size_t)n * K + k) = acc;
    if (k == 0 && dbias) dbias[n] += sb;
}

// The same refresh as a tiled transpose: one workgroup per 64 x 64 (n, c) tile of a layer, all taps.  tr_pack_all_kernel's
// transposed copy reads w[n][c][t] with n fastest -- every lane a different cache line, 3.8 GB of line traffic through L2 for
// 120 MB of weights (337 us).  Here the tile's rows are read as contiguous segments (64 * taps floats per n), parked in LDS
// as bf16 (row pitch = odd number of dwords) and written out along c (forward copy) and along n (transposed copy, taps
// flipped).  descs[i].first = cumulative TILE count.  Pads (Cin -> 16-multiple, N -> 16-multiple) are written as zeros.
// (round 5) whole tiles (N, Cin multiples of 64 -- all but a handful of layers): 16-byte global loads and stores, TAPS a compile-time
// constant (the scalar form below divided by run-time constants per element and stored 2 bytes per lane: 246 us for 120 MB in, 120 MB out)
template <int TAPS>
__device__ __forceinline__ void tr_pack_tile_fast(const float* __restrict__ w, bf16_t* tile, bf16_t* __restrict__ wf, bf16_t* __restrict__ wt,
                                                  int N, int Cin, int n0, int c0) {
    constexpr int ROW = 64 * TAPS, PITCH = ROW + 2, R4 = ROW / 4;
    const int tid = threadIdx.x;
    for (int e = tid; e < 64 * R4; e += 256) {
        const int nl = e / R4, rem = (e - nl * R4) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + ((size_t)(n0 + nl) * Cin + c0) * TAPS + rem);
        uint32_t* dst = reinterpret_cast<uint32_t*>(tile + nl * PITCH + rem);
        dst[0] = rldm::pack_bf16x2(v[0], v[1]);
        dst[1] = rldm::pack_bf16x2(v[2], v[3]);
    }
    __syncthreads();
    typedef unsigned short u16;
    const u16* t16 = reinterpret_cast<const u16*>(tile);
    for (int e = tid; e < 64 * TAPS * 8; e += 256) {                 // (nl, t, 8 channels), channel octet fastest
        const int c8 = e & 7, q = e >> 3, t = q % TAPS, nl = q / TAPS;
        const u16* src = t16 + nl * PITCH + (8 * c8) * TAPS + t;
        uint4 u;
        u.x = src[0] | ((uint32_t)src[TAPS] << 16); u.y = src[2 * TAPS] | ((uint32_t)src[3 * TAPS] << 16);
        u.z = src[4 * TAPS] | ((uint32_t)src[5 * TAPS] << 16); u.w = src[6 * TAPS] | ((uint32_t)src[7 * TAPS] << 16);
        *reinterpret_cast<uint4*>(wf + ((size_t)(n0 + nl) * TAPS + t) * Cin + c0 + 8 * c8) = u;
    }
    if (wt) {
        for (int e = tid; e < 64 * TAPS * 8; e += 256) {             // (cl, t, 8 output channels), octet fastest
            const int n8 = e & 7, q = e >> 3, t = q % TAPS, cl = q / TAPS;
            const u16* src = t16 + (8 * n8) * PITCH + cl * TAPS + t;
            uint4 u;
            u.x = src[0] | ((uint32_t)src[PITCH] << 16); u.y = src[2 * PITCH] | ((uint32_t)src[3 * PITCH] << 16);
            u.z = src[4 * PITCH] | ((uint32_t)src[5 * PITCH] << 16); u.w = src[6 * PITCH] | ((uint32_t)src[7 * PITCH] << 16);
            *reinterpret_cast<uint4*>(wt + ((size_t)(c0 + cl) * TAPS + (TAPS - 1 - t)) * N + n0 + 8 * n8) = u;
        }
    }
}

__global__ __launch_bounds__(256) void tr_pack_tiles_kernel(const float* __restrict__ params, const rldm_pack_desc* __restrict__ d,
                                                            int nlayers) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64 * (64 * 9 + 2)];
    int lo = 0, hi = nlayers - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (d[mid].first <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const rldm_pack_desc L = d[lo];
    const int taps = L.taps, N = L.N, Cin = L.Cin;
    const int Cin_pad = (Cin + 15) / 16 * 16, N_pad = (N + 15) / 16 * 16;
    const int ctiles = (Cin + 63) / 64;
    const int tl = (int)(blockIdx.x - L.first);
    const int n0 = (tl / ctiles) * 64, c0 = (tl % ctiles) * 64;
    const float* w = params + L.param_offset;
    if ((N & 63) == 0 && (Cin & 63) == 0 && (taps == 9 || taps == 1) && (L.param_offset & 3) == 0) {
        bf16_t* wf_ = static_cast<bf16_t*>(L.w_forward);
        bf16_t* wt_ = static_cast<bf16_t*>(L.w_transposed);
        if (taps == 9) tr_pack_tile_fast<9>(w, tile, wf_, wt_, N, Cin, n0, c0);
        else tr_pack_tile_fast<1>(w, tile, wf_, wt_, N, Cin, n0, c0);
        return;
    }
    const int row = 64 * taps, pitch = row + 2;
    const int ncol = min(64, Cin - c0) * taps;                       // valid floats of a row segment
    for (int e = threadIdx.x; e < 64 * row; e += 256) {
        const int nl = e / row, rem = e - nl * row;
        float v = 0.f;
        if (n0 + nl < N && rem < ncol) v = w[((size_t)(n0 + nl) * Cin + c0) * taps + rem];
        tile[nl * pitch + rem] = rldm::f32_to_bf16(v);
    }
    __syncthreads();
    bf16_t* wf = static_cast<bf16_t*>(L.w_forward);
    const int cw = min(64, Cin_pad - c0);                           // columns to write (pads included: zeros from the tile)
    for (int e = threadIdx.x; e < 64 * taps * 64; e += 256) {        // (nl, t, cl), cl fastest
        const int cl = e & 63, q = e >> 6, t = q % taps, nl = q / taps;
        if (n0 + nl < N && cl < cw) wf[((size_t)(n0 + nl) * taps + t) * Cin_pad + c0 + cl] = tile[nl * pitch + cl * taps + t];
    }
    if (L.w_transposed) {
        bf16_t* wt = static_cast<bf16_t*>(L.w_transposed);
        const int nw = min(64, N_pad - n0);
        for (int e = threadIdx.x; e < 64 * taps * 64; e += 256) {    // (cl, t, nl), nl fastest
            const int nl = e & 63, q = e >> 6, t = q % taps, cl = q / taps;
            if (c0 + cl < Cin && nl < nw)
                wt[((size_t)(c0 + cl) * taps + (taps - 1 - t)) * N_pad + n0 + nl] = tile[nl * pitch + cl * taps + t];
        }
    }
}

inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

#define TR_LAUNCH_CHECK() RLDM_HIP_CHECK(hipGetLastError())

extern "C" {

// launch shape of the LDS-staged conv: narrow (64-channel) tile? how many K splits?  (shared by rldm_train_conv and
// rldm_train_conv_splits, which lets the host hand out pre-zeroed outputs to the split launches)
static void conv_lds_plan(int P, int N, int Cin, int taps, bool* narrow_out, int* ksplit_out) {
    const long long gx = (P + 63) / 64;
    long long gy = (N + 127) / 128;
    // 128-channel tiles stage the pixel operand half as often; 64-channel tiles double the workgroups in flight, which
    // is what hides a stage's load latency when the launch has fewer than ~2 workgroups per CU (3x3 levels 1-3: 10-20 %
    // faster; the 512-workgroup level 0 and the two-stage 1x1 convs are faster with the wide tile)
    static const int bn_env = getenv("RLDM_TR_BN") ? atoi(getenv("RLDM_TR_BN")) : 0;
    const bool narrow = bn_env ? bn_env == 64 : (taps == 9 && gx * gy < 512);   // (measured per level)
    if (narrow) gy = (N + 63) / 64;
    // split K when the launch cannot fill the chip (see the kernel): aim at >= 512 workgroups, >= 3 stages each
    static const int ks_env = getenv("RLDM_TR_KSPLIT") ? atoi(getenv("RLDM_TR_KSPLIT")) : -1;
    const int niter = taps * (Cin / (Cin % 64 == 0 ? 64 : 32));
    const long long wgs = gx * gy;
    static const int tgt_env = getenv("RLDM_TR_KSPLIT_TARGET") ? atoi(getenv("RLDM_TR_KSPLIT_TARGET")) : 512;
    static const int minst_env = getenv("RLDM_TR_KSPLIT_MINSTAGES") ? atoi(getenv("RLDM_TR_KSPLIT_MINSTAGES")) : 3;
    int ksplit = wgs > 192 ? 1 : (int)std::min<long long>((tgt_env + wgs - 1) / wgs, niter / minst_env);
    if (ks_env >= 0) ksplit = ks_env;
    *narrow_out = narrow;
    *ksplit_out = std::max(1, std::min(ksplit, niter));
}

int rldm_train_conv_splits(const rldm_train_conv_desc* d, int rowadd_ld) {
    if (!d || d->stride < 1) return 1;
    const int sh = d->mode ? 1 : 0;
    const int P = d->B * ((d->Win << sh) / d->stride) * ((d->Hin << sh) / d->stride);
    if (!(P >= 64 && (rowadd_ld & 3) == 0 && d->Cin % 32 == 0)) return 1;
    bool narrow;
    int ksplit;
    conv_lds_plan(P, d->N, d->Cin, d->taps, &narrow, &ksplit);
    return ksplit;
}

// The scratch buffers of this file (arrival tickets, partial tiles, fp64 accumulators) are process-wide and live on the device that was
// current when the first of them was allocated: ONE training device per process (one process per GPU is how this library scales).
// A call on another device is refused instead of handing it a pointer into the first device's memory.
static int tr_scratch_device_ok() {
    static int dev0 = -1;
    int dev = -1;
    RLDM_HIP_CHECK(hipGetDevice(&dev));
    if (dev0 < 0) dev0 = dev;
    RLDM_REQUIRE(dev == dev0, "the training scratch buffers of this process live on device " + std::to_string(dev0) +
                                  ": one training device per process (current device " + std::to_string(dev) + ")");
    return 0;
}

// arrival counters of the fused split-K launches: zeroed once, every launch leaves them zeroed (one caller thread, stream ordered;
// never reallocated during a stream capture: the first, eager, step of a shape sizes it)
static int fuse_tickets(size_t count, hipStream_t st, unsigned** out) {
    static unsigned* buf = nullptr;
    static size_t cap = 0;
    if (tr_scratch_device_ok()) return 1;
    if (count > cap) {
        RLDM_HIP_CHECK(hipStreamSynchronize(st));
        if (buf) RLDM_HIP_CHECK(hipFree(buf));
        buf = nullptr;
        cap = 0;
        const size_t want = std::max<size_t>(count, 16384);
        RLDM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&buf), want * sizeof(unsigned)));
        RLDM_HIP_CHECK(hipMemset(buf, 0, want * sizeof(unsigned)));
        cap = want;
    }
    *out = buf;
    return 0;
}

// The reduction of the last all-taps weight-gradient launch's partial tiles, waiting for a conv launch to ride on
// (rldm_train_defer_reduce; one caller thread, stream ordered).
struct PendingReduce { const float* part = nullptr; float* dw = nullptr; int slices = 0, N = 0, Cin = 0, taps = 0; hipStream_t st = nullptr; };
static PendingReduce g_red;
static int g_defer_reduce = 0;

static int flush_reduce() {
    if (!g_red.part) return 0;
    tr_wgrad_reduce_vec_kernel<<<nblk((size_t)g_red.N * g_red.Cin * g_red.taps / 4), 256, 0, g_red.st>>>(g_red.part, g_red.slices, g_red.N,
                                                                                                      g_red.Cin, g_red.taps, g_red.dw);
    g_red.part = nullptr;
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

// attach the pending reduction to a conv launch of gx x gy workgroups per z-plane: -> extra z-planes
static int attach_reduce(TrFuse& f, hipStream_t st, unsigned gx, unsigned gy, unsigned gz) {
    if (!g_red.part) return 0;
    if (g_red.st != st) { if (flush_reduce()) return -1; return 0; }
    const size_t items = (size_t)g_red.N * g_red.Cin * g_red.taps / 4;
    const int want = (int)std::min<size_t>((items + 255) / 256, 512);
    const int planes = (int)((want + (size_t)gx * gy - 1) / ((size_t)gx * gy));
    if (gz + planes > 65535) { if (flush_reduce()) return -1; return 0; }
    f.rpart = g_red.part; f.rdw = g_red.dw; f.rslices = g_red.slices; f.rN = g_red.N; f.rCin = g_red.Cin; f.rtaps = g_red.taps;
    f.rz0 = (int)gz; f.rblocks = want;
    g_red.part = nullptr;
    return planes;
}

int rldm_train_defer_reduce(int on) {
    g_defer_reduce = on;
    return on ? 0 : flush_reduce();
}

int rldm_train_flush_reduce(void) { return flush_reduce(); }

static TrFuse to_device_fuse(const rldm_train_fuse* fu, int Cin, int N) {
    TrFuse f{};
    if (!fu) { f.C0 = Cin; return f; }
    f.x1 = fu->x1; f.C0 = fu->x1 ? fu->C0 : Cin;
    f.cs0 = fu->cs0; f.cs1 = fu->cs1; f.gamma = fu->gamma; f.beta = fu->beta; f.silu = fu->silu; f.groups = fu->groups; f.eps = fu->eps;
    f.cs_out = fu->cs_out;
    f.g0 = fu->g0; f.g1 = fu->g1; f.G0 = fu->g1 ? fu->G0 : N; f.gcs0 = fu->gcs0; f.gcs1 = fu->gcs1; f.ggamma = fu->ggamma; f.gbeta = fu->gbeta;
    f.gsilu = fu->gsilu; f.ggroups = fu->ggroups; f.geps = fu->geps; f.gs_out = fu->gs_out;
    return f;
}

// can the fused conv kernels run this launch?  (LDS-staged kernels; tiles inside one image; chunks inside one source)
static bool conv_fuse_ok(const rldm_train_conv_desc* d, const rldm_train_fuse* fu, int rowadd_ld) {
    const int sh = d->mode ? 1 : 0;
    const int Wout = (d->Win << sh) / d->stride, Hout = (d->Hin << sh) / d->stride;
    const int P = d->B * Wout * Hout, CK = d->Cin % 64 == 0 ? 64 : 32;
    if (!(P >= 64 && (rowadd_ld & 3) == 0 && d->Cin % 32 == 0 && (Wout * Hout) % 64 == 0)) return false;
    if (fu->x1 && (fu->C0 <= 0 || fu->C0 >= d->Cin || fu->C0 % CK != 0 || (d->Cin - fu->C0) % 4 != 0 || fu->C0 % 4 != 0)) return false;
    if (fu->cs0 && (d->Cin > 768 || fu->groups < 1 || fu->groups > 64 || d->Cin % fu->groups != 0 || !fu->gamma || !fu->beta || (fu->x1 && !fu->cs1))) return false;
    if (fu->cs_out && fu->gs_out) return false;
    if ((fu->cs_out || fu->gs_out) && d->N % 4 != 0) return false;          // (the tile epilogue works on channel quads)
    if (fu->gs_out) {
        if (!fu->g0 || !fu->gcs0 || !fu->ggamma || !fu->gbeta || fu->ggroups < 1 || d->N % fu->ggroups != 0) return false;
        if (fu->g1 && (!fu->gcs1 || fu->G0 <= 0 || fu->G0 >= d->N || fu->G0 % 4 != 0)) return false;
    }
    return true;
}

int rldm_train_conv_fused_ok(const rldm_train_conv_desc* d, const rldm_train_fuse* fu, int rowadd_ld) {
    return d && fu && d->stride >= 1 && conv_fuse_ok(d, fu, rowadd_ld) ? 1 : 0;
}

static int train_conv_impl(const rldm_train_conv_desc* d, const rldm_train_fuse* fu, const float* x, const void* w_packed, const float* bias,
                           const float* rowadd, int rowadd_ld, const float* res, float* y, int accumulate, void* stream);

int rldm_train_conv(const rldm_train_conv_desc* d, const float* x, const void* w_packed, const float* bias, const float* rowadd,
                    int rowadd_ld, const float* res, float* y, int accumulate, void* stream) {
    return train_conv_impl(d, nullptr, x, w_packed, bias, rowadd, rowadd_ld, res, y, accumulate, stream);
}

int rldm_train_conv_fused(const rldm_train_conv_desc* d, const rldm_train_fuse* fu, const float* x, const void* w_packed, const float* bias,
                          const float* rowadd, int rowadd_ld, const float* res, float* y, int prezeroed, void* stream) {
    RLDM_REQUIRE(d && fu, "null argument");
    RLDM_REQUIRE(conv_fuse_ok(d, fu, rowadd_ld), "rldm_train_conv_fused: shape not supported (ask rldm_train_conv_fused_ok)");
    return train_conv_impl(d, fu, x, w_packed, bias, rowadd, rowadd_ld, res, y, prezeroed, stream);
}

static int train_conv_impl(const rldm_train_conv_desc* d, const rldm_train_fuse* fu, const float* x, const void* w_packed, const float* bias,
                           const float* rowadd, int rowadd_ld, const float* res, float* y, int accumulate, void* stream) {
    RLDM_REQUIRE(d && x && w_packed && y, "null argument");
    RLDM_REQUIRE((d->taps == 1 || d->taps == 9) && (d->stride == 1 || d->stride == 2) && d->mode >= 0 && d->mode <= 2, "bad conv desc");
    RLDM_REQUIRE(d->B > 0 && d->Win > 0 && d->Hin > 0 && d->Cin > 0 && d->N > 0, "bad shape");
    TrConv p;
    p.x = x; p.w = static_cast<const bf16_t*>(w_packed); p.bias = bias; p.rowadd = rowadd; p.res = res; p.y = y;
    p.B = d->B; p.Win = d->Win; p.Hin = d->Hin; p.Cin = d->Cin; p.Cin_pad = (d->Cin + 15) / 16 * 16;
    const int sh = d->mode ? 1 : 0;
    p.Wout = (d->Win << sh) / d->stride; p.Hout = (d->Hin << sh) / d->stride;
    RLDM_REQUIRE(p.Wout > 0 && p.Hout > 0, "empty output");
    p.N = d->N; p.taps = d->taps; p.stride = d->stride; p.mode = d->mode; p.rowadd_ld = rowadd_ld; p.accumulate = accumulate;
    const int P = p.B * p.Wout * p.Hout;
    dim3 grid((P + 63) / 64, (p.N + 127) / 128);
    const bool aligned = (rowadd_ld & 3) == 0;         // (vector epilogue reads the per-sample row 16 bytes at a time)
    const bool lds = P >= 64 && aligned;               // Linear layers on a handful of rows: the direct kernel
    hipStream_t st = (hipStream_t)stream;
    TrFuse f = to_device_fuse(fu, p.Cin, p.N);
    if (lds && p.Cin % 32 == 0) {
        bool narrow;
        int ksplit;
        conv_lds_plan(P, p.N, p.Cin, p.taps, &narrow, &ksplit);
        if (narrow) grid.y = (p.N + 63) / 64;
        if (ksplit > 1) {
            if (!accumulate) tr_zero_kernel<<<nblk((size_t)P * p.N / 4 + 1), 256, 0, st>>>(y, (size_t)P * p.N);   // (a kernel, not a memset node: see tr_zero_kernel)
            grid.z = ksplit;
            if (fu && (f.cs_out || f.gs_out) && fuse_tickets((size_t)grid.x * grid.y, st, &f.tickets)) return 1;
        } else if (fu && (f.cs_out || f.gs_out)) {
            RLDM_REQUIRE(!accumulate, "rldm_train_conv_fused: the fused epilogue writes y (no accumulation)");
            p.accumulate = 0;
        }
        {                                           // the previous weight gradient's reduction rides on this launch
            const int planes = attach_reduce(f, st, grid.x, grid.y, grid.z);
            if (planes < 0) return 1;
            grid.z += planes;
        }
        static const bool nohalo_env = getenv("RLDM_TR_NO_HALO") != nullptr;           // A/B: the per-tap staging kernel
        const int H = p.Hout;
        const bool halo = !nohalo_env && ksplit == 1 && p.taps == 9 && p.stride == 1 && p.mode == 0 && p.Cin % 64 == 0 && p.N % 4 == 0 &&
                          H >= 2 && H <= 32 && (H & (H - 1)) == 0 && (p.Wout * H) % 64 == 0 && p.Wout % (64 / H) == 0 && p.Wout >= 64 / H + 2;
        static const int pt_env = getenv("RLDM_TR_PT") ? atoi(getenv("RLDM_TR_PT")) : 0;
        const bool wide_px = pt_env == 128 && !f.rblocks && halo && !narrow && (p.Wout * H) % 128 == 0 && p.Wout % (128 / H) == 0 && p.Wout >= 128 / H + 2;
        if (fu) {                                   // the fused instances (conv_fuse_ok held)
            if (halo) {
                if (narrow) tr_conv_halo_kernel<64, 64, true><<<grid, 256, 0, st>>>(p, f);
                else tr_conv_halo_kernel<128, 64, true><<<grid, 256, 0, st>>>(p, f);
            } else if (p.Cin % 64 == 0) {
                if (narrow) tr_conv_lds_kernel<64, 64, true><<<grid, 256, 0, st>>>(p, f);
                else tr_conv_lds_kernel<64, 128, true><<<grid, 256, 0, st>>>(p, f);
            } else {
                if (narrow) tr_conv_lds_kernel<32, 64, true><<<grid, 256, 0, st>>>(p, f);
                else tr_conv_lds_kernel<32, 128, true><<<grid, 256, 0, st>>>(p, f);
            }
        } else if (halo && wide_px) {
            grid.x = P / 128;
            tr_conv_halo_kernel<128, 128><<<grid, 512, 0, st>>>(p, f);
        } else if (halo) {
            if (narrow) tr_conv_halo_kernel<64, 64><<<grid, 256, 0, st>>>(p, f);
            else tr_conv_halo_kernel<128, 64><<<grid, 256, 0, st>>>(p, f);
        } else if (p.Cin % 64 == 0) {
            if (narrow) tr_conv_lds_kernel<64, 64><<<grid, 256, 0, st>>>(p, f);
            else tr_conv_lds_kernel<64, 128><<<grid, 256, 0, st>>>(p, f);
        } else {
            if (narrow) tr_conv_lds_kernel<32, 64><<<grid, 256, 0, st>>>(p, f);
            else tr_conv_lds_kernel<32, 128><<<grid, 256, 0, st>>>(p, f);
        }
    } else {
        RLDM_REQUIRE(!fu, "rldm_train_conv_fused: not an LDS-staged shape");
        if (flush_reduce()) return 1;
        if (p.Cin % 16 == 0) tr_conv_kernel<true><<<grid, 256, 0, st>>>(p);
        else tr_conv_kernel<false><<<grid, 256, 0, st>>>(p);
    }
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_wgrad(const rldm_train_conv_desc* d, const float* dy, const float* x, float* dw, void* stream) {
    return rldm_train_wgrad_bias(d, dy, x, dw, nullptr, 0, 0, nullptr, stream);
}

static bool wgrad_v2_shape(const rldm_train_conv_desc* d) {
    const int sh = d->mode ? 1 : 0;
    const int Wout = (d->Win << sh) / d->stride, Hout = (d->Hin << sh) / d->stride;
    if (!(d->stride == 1 && d->mode <= 1 && d->N % 64 == 0 && d->Cin % 64 == 0 && Hout >= 2 && Hout <= 16 && (Hout & (Hout - 1)) == 0)) return false;
    for (int l = 6; l >= 3; --l)
        if ((Hout << l) <= 128 && ((Hout + 2) << l) <= 160 && Wout % (1 << l) == 0) return true;
    return false;
}

int rldm_train_wgrad_fused_ok(const rldm_train_conv_desc* d, const rldm_train_fuse* fu) {
    if (!d || !fu || getenv("RLDM_TR_WG_V1")) return 0;
    if (!wgrad_v2_shape(d) || d->mode != 0 || d->B > 16) return 0;
    if (fu->x1 && (fu->C0 <= 0 || fu->C0 >= d->Cin || fu->C0 % 64 != 0)) return 0;
    if (fu->cs0 && (fu->groups < 1 || d->Cin % fu->groups != 0 || !fu->gamma || !fu->beta || (fu->x1 && !fu->cs1))) return 0;
    return 1;
}

// ---- (round 6) grouped weight gradients: the queue ------------------------------------------------------------------------------------
// rldm_train_wgrad_group(1): all-taps weight gradients are queued instead of launched; rldm_train_wgrad_group_flush (or group(0)) runs
// the queue as a few launches of tr_wgrad2_group_kernel.  One caller thread, one stream per queue (a call on another stream flushes
// first); the caller keeps every queued operand alive and unmodified until the flush (rangeldm_amd/training.py does).
struct WgQueued { WgItem it; int taps; bool fu, v3; int tiles; size_t part_floats; size_t smem; double unit; };
static std::vector<WgQueued> g_wgq;
static hipStream_t g_wgq_stream = nullptr;
static int g_wg_group = 0;

static int wg_group_flush() {
    if (g_wgq.empty()) return 0;
    if (tr_scratch_device_ok()) return 1;
    hipStream_t st = g_wgq_stream;
    // K slices.  With every layer of a class in one launch the chip is full whatever a single layer brings, so the slices per tile --
    // partial tiles to write and re-read, a last arriver to sum them -- shrink from the launch-per-layer form's 64 to <= 16: a workgroup
    // takes `budget` units of work (a unit = a chunk of 128 pixels x 9 taps; a 1x1 tap set costs about a third).  Measured per class at
    // the RangeLDM size, batch 8 (profiles/round6_wgrad_group_budget.txt): 3x3 launches are fastest at 32 units (712 / 276 us; 16: 769 /
    // 307, 8: 996 / 434, 4: 1543 / 514 -- the last arriver's serial sum over the slices is the tail), the 1x1 launches at 8 - 16.
    static const int budget_env = getenv("RLDM_TR_WG_GROUP_CPW") ? atoi(getenv("RLDM_TR_WG_GROUP_CPW")) : 0;
    for (int cls = 0; cls < 8; ++cls) {
        const int taps = (cls & 1) ? 1 : 9;
        const bool fu = (cls & 2) != 0, v3 = (cls & 4) != 0;
        const double budget = budget_env ? (double)budget_env : (taps == 9 ? 32.0 : 12.0);
        for (auto& q : g_wgq) {
            if (q.taps != taps || q.fu != fu || q.v3 != v3) continue;
            WgItem& it = q.it;
            int cpw = std::max(1, (int)(budget / q.unit + 0.5));
            cpw = std::min(cpw, it.nchunks);
            int Z = (it.nchunks + cpw - 1) / cpw;
            cpw = (it.nchunks + Z - 1) / Z;
            Z = (it.nchunks + cpw - 1) / cpw;
            it.cpw = cpw; it.Z = Z;
            q.part_floats = Z > 1 ? (size_t)q.taps * Z * it.N * it.Cin : 0;
        }
    }
    // scratch of the partial tiles (layers with more than one K slice) and the arrival tickets: grown outside captures only (the first,
    // eager, step of a shape sizes them -- as every scratch buffer of this file)
    size_t need = 0, ntick = 0;
    for (auto& q : g_wgq) { need += q.part_floats; ntick += (size_t)q.tiles; }
    static float* scratch = nullptr;
    static size_t scratch_cap = 0;
    static unsigned* tickets = nullptr;
    static size_t tick_cap = 0;
    if (need > scratch_cap) {
        RLDM_HIP_CHECK(hipStreamSynchronize(st));
        if (scratch) RLDM_HIP_CHECK(hipFree(scratch));
        scratch = nullptr; scratch_cap = 0;
        RLDM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&scratch), need * sizeof(float)));
        scratch_cap = need;
    }
    if (ntick > tick_cap) {
        RLDM_HIP_CHECK(hipStreamSynchronize(st));
        if (tickets) RLDM_HIP_CHECK(hipFree(tickets));
        tickets = nullptr; tick_cap = 0;
        const size_t want = std::max<size_t>(ntick, 4096);
        RLDM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&tickets), want * sizeof(unsigned)));
        RLDM_HIP_CHECK(hipMemset(tickets, 0, want * sizeof(unsigned)));        // zeroed once; every launch leaves them zeroed
        tick_cap = want;
    }
    typedef void (*GroupKernel)(const WgGroup);
    static const GroupKernel kernels[8] = {tr_wgrad2_group_kernel<9, false, false>, tr_wgrad2_group_kernel<1, false, false>,
                                           tr_wgrad2_group_kernel<9, true, false>,  tr_wgrad2_group_kernel<1, true, false>,
                                           tr_wgrad2_group_kernel<9, false, true>,  tr_wgrad2_group_kernel<1, false, true>,
                                           tr_wgrad2_group_kernel<9, true, true>,   tr_wgrad2_group_kernel<1, true, true>};
    static bool attr = false;                       // (128 KiB: the kernel also has a static LDS word, the tiles need <= 88 KiB)
    if (!attr) {
        for (int k = 0; k < 8; ++k)
            RLDM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernels[k]), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr = true;
    }
    size_t poff = 0, toff = 0;
    for (auto& q : g_wgq) {
        q.it.part = scratch + poff; poff += q.part_floats;
        q.it.ticket_off = (unsigned)toff; toff += (size_t)q.tiles;
    }
    // one launch per (instantiation, <= kWgGroupMax layers), long workgroups first (the layers with the most pixels per slice)
    for (int cls = 0; cls < 8; ++cls) {
        const int taps = (cls & 1) ? 1 : 9;
        const bool fu = (cls & 2) != 0, v3 = (cls & 4) != 0;
        std::vector<const WgQueued*> sel;
        for (auto& q : g_wgq) if (q.taps == taps && q.fu == fu && q.v3 == v3) sel.push_back(&q);
        std::stable_sort(sel.begin(), sel.end(), [](const WgQueued* a, const WgQueued* b) {
            return (long long)a->it.cpw * (a->it.H << a->it.lwc) > (long long)b->it.cpw * (b->it.H << b->it.lwc); });
        for (size_t i0 = 0; i0 < sel.size(); i0 += kWgGroupMax) {
            WgGroup g;
            memset(static_cast<void*>(&g), 0, sizeof(g));
            g.n = (int)std::min<size_t>(kWgGroupMax, sel.size() - i0);
            g.tickets = tickets;
            size_t smem = 0;
            int blocks = 0;
            for (int i = 0; i < g.n; ++i) {
                const WgQueued* q = sel[i0 + i];
                g.item[i] = q->it;
                g.first[i] = blocks;
                blocks += q->tiles * q->it.Z;
                smem = std::max(smem, q->smem);
            }
            g.first[g.n] = blocks;
            kernels[cls]<<<blocks, 256, smem, st>>>(g);
            TR_LAUNCH_CHECK();
        }
    }
    g_wgq.clear();
    return 0;
}

int rldm_train_wgrad_group(int on) {
    g_wg_group = on;
    return on ? 0 : wg_group_flush();
}

int rldm_train_wgrad_group_flush(void) { return wg_group_flush(); }

int rldm_train_wgrad_group_pending(void) { return (int)g_wgq.size(); }

static int train_wgrad_impl(const rldm_train_conv_desc* d, const rldm_train_fuse* fu, const float* dy, const float* x, float* dw, float* rows,
                            int rows_ld, int rows_accumulate, float* total, void* stream);

int rldm_train_wgrad_bias(const rldm_train_conv_desc* d, const float* dy, const float* x, float* dw, float* rows, int rows_ld,
                          int rows_accumulate, float* total, void* stream) {
    return train_wgrad_impl(d, nullptr, dy, x, dw, rows, rows_ld, rows_accumulate, total, stream);
}

int rldm_train_wgrad_fused(const rldm_train_conv_desc* d, const rldm_train_fuse* fu, const float* dy, const float* x, float* dw, float* rows,
                           int rows_ld, int rows_accumulate, float* total, void* stream) {
    RLDM_REQUIRE(d && fu && rldm_train_wgrad_fused_ok(d, fu), "rldm_train_wgrad_fused: shape not supported (ask rldm_train_wgrad_fused_ok)");
    return train_wgrad_impl(d, fu, dy, x, dw, rows, rows_ld, rows_accumulate, total, stream);
}

static int train_wgrad_impl(const rldm_train_conv_desc* d, const rldm_train_fuse* fu, const float* dy, const float* x, float* dw, float* rows,
                            int rows_ld, int rows_accumulate, float* total, void* stream) {
    RLDM_REQUIRE(d && dy && x && dw, "null argument");
    if (flush_reduce()) return 1;                     // (the partial-tile scratch is about to be rewritten)
    RLDM_REQUIRE((d->taps == 1 || d->taps == 9) && (d->stride == 1 || d->stride == 2) && d->mode >= 0 && d->mode <= 2, "bad conv desc");
    TrWgrad p;
    p.dy = dy; p.x = x; p.dw = dw;
    p.B = d->B; p.Win = d->Win; p.Hin = d->Hin; p.Cin = d->Cin; p.N = d->N; p.taps = d->taps; p.stride = d->stride; p.mode = d->mode;
    const int sh = d->mode ? 1 : 0;
    p.Wout = (d->Win << sh) / d->stride; p.Hout = (d->Hin << sh) / d->stride;
    const int P = p.B * p.Wout * p.Hout;
    const int tiles = ((p.N + 63) / 64) * ((p.Cin + 63) / 64) * p.taps;
    // a wave contracts `chunk` pixels into its own partial tile; ~1024 pixels per wave, fewer only to fill the chip
    int chunk = getenv("RLDM_TR_WG_CHUNK") ? atoi(getenv("RLDM_TR_WG_CHUNK")) : 1024;
    while (chunk > 128 && (long long)tiles * ((P + 4 * chunk - 1) / (4 * chunk)) < 256) chunk >>= 1;
    int splits = (P + 4 * chunk - 1) / (4 * chunk);
    p.chunk = chunk;
    size_t need = (size_t)p.taps * splits * 4 * p.N * p.Cin * sizeof(float);
    // all-taps kernel (see tr_wgrad2_kernel): stride 1, 64-multiples of channels, H a power of two <= 16, chunk of WC columns
    static const bool v1_env = getenv("RLDM_TR_WG_V1") != nullptr;                // A/B: the one-tap-per-workgroup kernel
    TrWgrad2 w2{};
    bool v2 = !v1_env && d->stride == 1 && d->mode <= 1 && p.N % 64 == 0 && p.Cin % 64 == 0 && p.Hout >= 2 && p.Hout <= 16 &&
              (p.Hout & (p.Hout - 1)) == 0;
    if (v2) {
        const int H = p.Hout, W = p.Wout;
        int lwc = -1;
        for (int l = 6; l >= 3; --l)
            if ((H << l) <= 128 && ((H + 2) << l) <= 160 && W % (1 << l) == 0) { lwc = l; break; }
        if (lwc < 0) v2 = false;
        else {
            w2.B = p.B; w2.W = W; w2.H = H; w2.Win = p.Win; w2.Hin = p.Hin; w2.Cin = p.Cin; w2.N = p.N; w2.mode = d->mode; w2.lwc = lwc;
            const int WC = 1 << lwc;
            w2.nchunks = p.B * (W / WC);
            const int tiles2 = (p.N / 64) * (p.Cin / 64);
            static const int wg_env = getenv("RLDM_TR_WG_BLOCKS") ? atoi(getenv("RLDM_TR_WG_BLOCKS")) : 256;
            // (few chunks -- the 32 x 2 level -- : two per workgroup halve the partial tiles for the same kernel time)
            static const int cpw_env = getenv("RLDM_TR_WG_CPW") ? atoi(getenv("RLDM_TR_WG_CPW")) : 0;
            const int cpw_min = cpw_env ? cpw_env : (w2.nchunks <= 16 ? 2 : 1);
            int Z = std::max(1, std::min(std::max(1, w2.nchunks / cpw_min), (wg_env + tiles2 - 1) / tiles2));
            w2.cpw = (w2.nchunks + Z - 1) / Z;
            Z = (w2.nchunks + w2.cpw - 1) / w2.cpw;
            w2.pitchA = WC * H + 8;
            w2.pitchB = (p.taps == 9 ? (H + 2) : H) * WC + 8;
            splits = Z;
            need = (size_t)p.taps * Z * p.N * p.Cin * sizeof(float);
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (v2 && g_wg_group) {
        // queued for a grouped launch (tr_wgrad2_group_kernel; K slices: wg_group_flush)
        if (!g_wgq.empty() && g_wgq_stream != st && wg_group_flush()) return 1;
        g_wgq_stream = st;
        const int KP = w2.H << w2.lwc;
        WgQueued q{};
        q.unit = (KP / 128.0) * (p.taps == 9 ? 1.0 : 0.35);
        WgItem& it = q.it;
        it.dy = dy; it.x = x; it.dw = dw; it.rows = rows; it.total = total; it.rows_ld = rows_ld;
        it.B = w2.B; it.W = w2.W; it.H = w2.H; it.Win = w2.Win; it.Hin = w2.Hin; it.Cin = w2.Cin; it.N = w2.N; it.mode = w2.mode;
        it.lwc = w2.lwc; it.cpw = 1; it.nchunks = w2.nchunks; it.Z = 1;         // (cpw / Z: decided per class at the flush)
        const TrFuse f = to_device_fuse(fu, p.Cin, p.N);
        it.x1 = f.x1; it.C0 = f.C0; it.cs0 = f.cs0; it.cs1 = f.cs1; it.gamma = f.gamma; it.beta = f.beta; it.silu = f.silu;
        it.groups = f.groups; it.eps = f.eps;
        q.taps = p.taps; q.fu = fu != nullptr;
        static const bool v3_off = getenv("RLDM_TR_WG_V3") && atoi(getenv("RLDM_TR_WG_V3")) == 0;       // (A/B: the round-5 staging)
        q.v3 = !v3_off && w2.H >= 8 && w2.mode == 0;
        q.tiles = (p.N / 64) * (p.Cin / 64);
        q.part_floats = 0;
        const int pitchB = q.v3 ? ((1 << w2.lwc) + (p.taps == 9 ? 2 : 0)) * w2.H + 8 : w2.pitchB;
        q.smem = (size_t)64 * w2.pitchA * 2 + (size_t)(p.taps == 9 ? 3 : 1) * 64 * pitchB * 2;
        if (fu) q.smem += (size_t)p.B * 64 * sizeof(float2);
        q.smem = std::max(q.smem, (size_t)32 * (64 * p.taps + 1) * sizeof(float));      // (Z == 1: the tile goes through LDS into dw)
        if (rows && !rows_accumulate) tr_zero2d_kernel<<<nblk((size_t)p.B * p.N), 256, 0, st>>>(rows, rows_ld, p.N, p.B);
        TR_LAUNCH_CHECK();
        g_wgq.push_back(q);
        return 0;
    }
    static float* scratch = nullptr;                   // (one caller thread; launches are stream ordered)
    static size_t scratch_cap = 0;
    if (tr_scratch_device_ok()) return 1;
    if (need > scratch_cap) {
        RLDM_HIP_CHECK(hipStreamSynchronize(st));
        if (scratch) RLDM_HIP_CHECK(hipFree(scratch));
        scratch = nullptr;
        scratch_cap = 0;
        RLDM_HIP_CHECK(hipMalloc(&scratch, need));
        scratch_cap = need;
    }
    p.dw = scratch;
    if (v2) {
        static const bool part_env = getenv("RLDM_TR_WG_PARTIALS") != nullptr;      // A/B: partial tiles + reduction launch
        w2.dy = dy; w2.x = x; w2.part = scratch;
        w2.dw = (part_env || p.taps == 9) ? nullptr : dw;      // (measured: 3x3 tiles are 9x larger, their 38 MB of atomics lose to the
                                                                //  partial tiles + reduction launch by 10-30 %; 1x1 wins 25-35 %)
        w2.rows = rows; w2.rows_ld = rows_ld; w2.total = total;
        if (rows && !rows_accumulate) tr_zero2d_kernel<<<nblk((size_t)p.B * p.N), 256, 0, st>>>(rows, rows_ld, p.N, p.B);
        size_t smem = (size_t)64 * w2.pitchA * 2 + (size_t)(p.taps == 9 ? 3 : 1) * 64 * w2.pitchB * 2;
        if (fu) smem += (size_t)p.B * 64 * sizeof(float2);          // GroupNorm coefficients of the tile's channels per image
        if (w2.dw) smem = std::max(smem, (size_t)32 * (64 * p.taps + 1) * sizeof(float));
        static bool attr = false;
        if (!attr) {
            RLDM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tr_wgrad2_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            RLDM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tr_wgrad2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            RLDM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tr_wgrad2_kernel<9, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            RLDM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tr_wgrad2_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr = true;
        }
        const dim3 grid((p.N / 64) * (p.Cin / 64), splits);
        const TrFuse f = to_device_fuse(fu, p.Cin, p.N);
        if (fu) {
            if (p.taps == 9) tr_wgrad2_kernel<9, true><<<grid, 256, smem, st>>>(w2, f);
            else tr_wgrad2_kernel<1, true><<<grid, 256, smem, st>>>(w2, f);
        } else if (p.taps == 9) tr_wgrad2_kernel<9><<<grid, 256, smem, st>>>(w2, f);
        else tr_wgrad2_kernel<1><<<grid, 256, smem, st>>>(w2, f);
        TR_LAUNCH_CHECK();
        if (!w2.dw) {
            if (g_defer_reduce) {                     // rides on the next conv launch (or is flushed by whatever comes first)
                g_red.part = scratch; g_red.dw = dw; g_red.slices = splits; g_red.N = p.N; g_red.Cin = p.Cin; g_red.taps = p.taps; g_red.st = st;
            } else
                tr_wgrad_reduce_vec_kernel<<<nblk((size_t)p.N * p.Cin * p.taps / 4), 256, 0, st>>>(scratch, splits, p.N, p.Cin, p.taps, dw);
        }
        TR_LAUNCH_CHECK();
        return 0;
    }
    RLDM_REQUIRE(!fu, "rldm_train_wgrad_fused: not an all-taps shape");
    if (rows || total) {                              // the one-tap kernel does not carry the column sums
        const int rc = rldm_train_colsum(dy, p.B, p.Wout * p.Hout, p.N, rows, rows_ld, rows_accumulate, total, stream);
        if (rc) return rc;
    }
    tr_wgrad_kernel<<<dim3(((p.N + 63) / 64) * ((p.Cin + 63) / 64), p.taps, splits), 256, 0, st>>>(p);
    TR_LAUNCH_CHECK();
    tr_wgrad_reduce_kernel<<<nblk((size_t)p.N * p.Cin * p.taps), 256, 0, st>>>(scratch, splits * 4, p.N, p.Cin, p.taps, dw);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_linear_rows(const float* x, int ldx, const void* w_packed, int K, const float* bias, float* y, int ldy, int B, int N,
                           int accumulate, void* stream) {
    RLDM_REQUIRE(x && w_packed && y, "null argument");
    RLDM_REQUIRE(B >= 1 && B <= LIN_MAXB && K % 8 == 0 && ldx % 4 == 0, "rows <= 16, K a multiple of 8");
    const int Kp = (K + 15) / 16 * 16;
    tr_linear_rows_kernel<<<(N + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, ldx, static_cast<const bf16_t*>(w_packed), Kp, K, bias, y, ldy, B,
                                                                       N, accumulate);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_linear_rows_wgrad(const float* dy, int ldy, const float* x, int ldx, int B, int N, int K, float* dw, float* dbias,
                                 void* stream) {
    RLDM_REQUIRE(dy && x && dw, "null argument");
    RLDM_REQUIRE(B >= 1 && K % 4 == 0 && ldx % 4 == 0, "K a multiple of 4");
    tr_linear_rows_wgrad_kernel<<<nblk((size_t)N * (K / 4)), 256, 0, (hipStream_t)stream>>>(dy, ldy, x, ldx, B, N, K, dw, dbias);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_colsum(const float* dy, int B, int npix, int N, float* rows, int rows_ld, int rows_accumulate, float* total,
                      void* stream) {
    RLDM_REQUIRE(dy && (rows || total), "null argument");
    hipStream_t st = (hipStream_t)stream;
    if (rows && !rows_accumulate) tr_zero2d_kernel<<<nblk((size_t)B * N), 256, 0, st>>>(rows, rows_ld, N, B);
    tr_colsum_kernel<<<dim3((N + 63) / 64, B, (npix + 255) / 256), 256, 0, st>>>(dy, npix, N, rows, rows_ld, total);
    TR_LAUNCH_CHECK();
    return 0;
}

// fp64 accumulators of the slab GroupNorm kernels: one growing device buffer (one caller thread; launches are stream
// ordered, so consecutive launches may share it).  Never reallocated while a stream capture is running: the first
// (eager) step of a shape sizes it.
static int gn_accumulators(size_t count, hipStream_t st, double** out) {
    static double* buf = nullptr;
    static size_t cap = 0;
    if (tr_scratch_device_ok()) return 1;
    if (count > cap) {
        RLDM_HIP_CHECK(hipStreamSynchronize(st));
        if (buf) RLDM_HIP_CHECK(hipFree(buf));
        buf = nullptr;
        cap = 0;
        const size_t want = std::max<size_t>(count, 4096);
        RLDM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&buf), want * sizeof(double)));
        RLDM_HIP_CHECK(hipMemset(buf, 0, want * sizeof(double)));        // zero once; every user leaves it zeroed (finish kernels)
        cap = want;
    }
    *out = buf;
    return 0;
}

int rldm_train_gn_forward(const float* x, int B, int npix, int C, int groups, float eps, const float* gamma, const float* beta,
                          int silu, float* stats, float* y, void* stream) {
    RLDM_REQUIRE(x && gamma && beta && stats && y, "null argument");
    RLDM_REQUIRE(C % groups == 0, "channels must be a multiple of the group count");
    hipStream_t st = (hipStream_t)stream;
    const bool slab = 1024 % C == 0 && C >= 4 && groups <= 64 && (size_t)npix * C >= (1u << 18);     // large tensors, see the kernel
    if (slab) {
        double* acc = nullptr;
        if (gn_accumulators((size_t)B * groups * 2, st, &acc)) return 1;
        tr_gn_stats_slab_kernel<<<dim3((npix + 63) / 64, B), 256, 0, st>>>(x, npix, C, groups, acc);
        tr_gn_stats_finish_kernel<<<nblk((size_t)B * groups), 256, 0, st>>>(acc, B * groups, (double)npix * (C / groups), eps,
                                                                           reinterpret_cast<float2*>(stats));
    } else if ((C / groups) % 4 == 0 && C / groups <= 16 && (C / groups & (C / groups - 1)) == 0) {
        static const bool nofuse = getenv("RLDM_TR_GN_NOFUSE") != nullptr;
        if (!nofuse) {
            tr_gn_fwd_fused_vec_kernel<<<dim3(groups, B), 256, 0, st>>>(x, npix, C, groups, eps, gamma, beta, silu,
                                                                        reinterpret_cast<float2*>(stats), y);
            TR_LAUNCH_CHECK();
            return 0;
        }
        tr_gn_stats_vec_kernel<<<dim3(groups, B), 256, 0, st>>>(x, npix, C, groups, eps, reinterpret_cast<float2*>(stats));
    } else
        tr_gn_stats_kernel<<<dim3(groups, B), 256, 0, st>>>(x, npix, C, groups, eps, reinterpret_cast<float2*>(stats));
    const size_t total = (size_t)B * npix * C;
    const int cpg = C / groups;
    if (cpg % 4 == 0 && total < (1ull << 31))
        tr_gn_fwd_vec_kernel<<<nblk(total / 4), 256, 0, st>>>(x, reinterpret_cast<const float2*>(stats), gamma, beta, (unsigned)npix * C, C,
                                                            cpg, groups, silu, (unsigned)(total / 4), y);
    else
        tr_gn_fwd_kernel<<<nblk(total), 256, 0, st>>>(x, reinterpret_cast<const float2*>(stats), gamma, beta, npix, C, groups, silu, total, y);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_gn_backward(const float* x, const float* dy, const float* stats, int B, int npix, int C, int groups,
                           const float* gamma, const float* beta, int silu, float* scratch, float* dx, int accumulate,
                           float* dgamma, float* dbeta, void* stream) {
    RLDM_REQUIRE(x && dy && stats && gamma && beta && scratch && dx && dgamma && dbeta, "null argument");
    RLDM_REQUIRE(C % groups == 0 && C / groups <= 256, "channels per group must be <= 256");
    hipStream_t st = (hipStream_t)stream;
    const bool slab = 1024 % C == 0 && C >= 4 && C <= 1024 && groups <= 64 && (size_t)npix * C >= (1u << 18);
    if (slab) {
        double* acc = nullptr;
        if (gn_accumulators((size_t)B * groups * 2, st, &acc)) return 1;
        static const int slab_env = getenv("RLDM_TR_GN_SLAB") ? atoi(getenv("RLDM_TR_GN_SLAB")) : 64;
        tr_gn_bwd_reduce_slab_kernel<<<dim3((npix + slab_env - 1) / slab_env, B), 256, 0, st>>>(x, dy, reinterpret_cast<const float2*>(stats), gamma,
                                                                               beta, npix, C, groups, silu, acc, dgamma, dbeta);
        tr_gn_sums_finish_kernel<<<nblk((size_t)B * groups), 256, 0, st>>>(acc, B * groups, reinterpret_cast<float2*>(scratch));
    } else if (gn_vec_reduce_env() && (C / groups) % 4 == 0 && C / groups <= 16 && (C / groups & (C / groups - 1)) == 0)
        // (A/B only: measured 14.0 us against 10.1 us for the channel-pinned kernel below -- four sigmoids per iteration and the
        //  wider LDS epilogue cost more than the 16-byte loads save; fusing the apply pass into it was no faster either)
        tr_gn_bwd_reduce_vec_kernel<<<dim3(groups, B), 256, 0, st>>>(x, dy, reinterpret_cast<const float2*>(stats), gamma, beta, npix, C,
                                                                 groups, silu, reinterpret_cast<float2*>(scratch), dgamma, dbeta);
    else
        tr_gn_bwd_reduce_kernel<<<dim3(groups, B), 256, 0, st>>>(x, dy, reinterpret_cast<const float2*>(stats), gamma, beta, npix, C,
                                                                 groups, silu, reinterpret_cast<float2*>(scratch), dgamma, dbeta);
    const size_t total = (size_t)B * npix * C;
    if ((C / groups) % 4 == 0 && total < (1ull << 31))
        tr_gn_bwd_apply_vec_kernel<<<nblk(total / 4), 256, 0, st>>>(x, dy, reinterpret_cast<const float2*>(stats),
                                                                  reinterpret_cast<const float2*>(scratch), gamma, beta, (unsigned)npix * C,
                                                                  C, C / groups, groups, silu, accumulate,
                                                                  1.f / ((float)npix * (C / groups)), (unsigned)(total / 4), dx);
    else
        tr_gn_bwd_apply_kernel<<<nblk(total), 256, 0, st>>>(x, dy, reinterpret_cast<const float2*>(stats),
                                                        reinterpret_cast<const float2*>(scratch), gamma, beta, npix, C, groups, silu,
                                                        accumulate, total, dx);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_chan_stats(const float* x, int B, int npix, int C, float* cs, void* stream) {
    RLDM_REQUIRE(x && cs && B > 0 && npix > 0, "null argument");
    RLDM_REQUIRE(C % 4 == 0 && C <= 1024, "channels: a multiple of 4, <= 1024");
    const int slab = npix >= 4096 ? 256 : 64;
    tr_chan_stats_kernel<<<dim3((npix + slab - 1) / slab, B), 256, 0, (hipStream_t)stream>>>(x, npix, C, slab, cs);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_gn_backward_apply(const float* dz, const float* x0, const float* x1, int C0, const float* cs0, const float* cs1,
                                 const float* gs, int B, int npix, int C, int groups, float eps, const float* gamma, const float* res,
                                 float* dx0, int accumulate0, float* dx1, int accumulate1, float* dgamma, float* dbeta, void* stream) {
    RLDM_REQUIRE(dz && x0 && cs0 && gs && gamma && dx0, "null argument");
    RLDM_REQUIRE(C % 4 == 0 && C <= 768 && groups >= 1 && groups <= 64 && C % groups == 0, "channels: a multiple of 4 and of the groups, <= 768");
    RLDM_REQUIRE(!x1 || (cs1 && dx1 && C0 > 0 && C0 < C && C0 % 4 == 0), "two sources: C0 a multiple of 4 inside (0, C)");
    RLDM_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "d gamma and d beta come together");
    TrGnApply a;
    a.dz = dz; a.x0 = x0; a.x1 = x1; a.cs0 = cs0; a.cs1 = cs1; a.gs = gs; a.gamma = gamma; a.res = res; a.dx0 = dx0; a.dx1 = dx1;
    a.dgamma = dgamma; a.dbeta = dbeta; a.B = B; a.npix = npix; a.C = C; a.C0 = x1 ? C0 : C; a.groups = groups;
    a.acc0 = accumulate0; a.acc1 = accumulate1; a.eps = eps;
    // slabs of >= 16 pixels, ~1024 blocks at most
    int slab = 16;
    while ((long long)B * ((npix + slab - 1) / slab) > 512) slab *= 2;
    a.slab = slab;
    tr_gn_bwd_apply2_kernel<<<dim3((npix + slab - 1) / slab, B), 256, 0, (hipStream_t)stream>>>(a);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_attention_forward(const float* q, const float* k, const float* v, int B, int L, int C, float* o, float* lse,
                                 void* stream) {
    RLDM_REQUIRE(q && k && v && o && lse, "null argument");
    RLDM_REQUIRE(C % 8 == 0, "head_dim is 8");
    if (!attention_scalar()) return rldm::tr_attention_forward_mfma(q, k, v, C, B, L, C, o, lse, (hipStream_t)stream);
    tr_attn_fwd_kernel<<<dim3((L + 127) / 128, C / 8, B), 128, 0, (hipStream_t)stream>>>(q, k, v, L, C, 0.35355339059327373f, o, lse);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_attention_qkv_forward(const float* qkv, int B, int L, int C, float* o, float* lse, void* stream) {
    RLDM_REQUIRE(qkv && o && lse, "null argument");
    RLDM_REQUIRE(C % 8 == 0, "head_dim is 8");
    return rldm::tr_attention_forward_mfma(qkv, qkv + C, qkv + 2 * C, 3 * C, B, L, C, o, lse, (hipStream_t)stream);
}

int rldm_train_attention_qkv_backward(const float* qkv, const float* o, const float* dO, const float* lse, int B, int L, int C,
                                      float* delta, float* dqkv, void* stream) {
    RLDM_REQUIRE(qkv && o && dO && lse && delta && dqkv, "null argument");
    RLDM_REQUIRE(C % 8 == 0, "head_dim is 8");
    return rldm::tr_attention_backward_mfma(qkv, qkv + C, qkv + 2 * C, 3 * C, o, dO, lse, B, L, C, delta, dqkv, dqkv + C, dqkv + 2 * C,
                                            (hipStream_t)stream);
}

int rldm_train_attention_backward(const float* q, const float* k, const float* v, const float* o, const float* dO, const float* lse,
                                  int B, int L, int C, float* delta, float* dq, float* dk, float* dv, void* stream) {
    RLDM_REQUIRE(q && k && v && o && dO && lse && delta && dq && dk && dv, "null argument");
    RLDM_REQUIRE(C % 8 == 0, "head_dim is 8");
    hipStream_t st = (hipStream_t)stream;
    if (!attention_scalar()) return rldm::tr_attention_backward_mfma(q, k, v, C, o, dO, lse, B, L, C, delta, dq, dk, dv, st);
    const float scale = 0.35355339059327373f;
    const dim3 grid((L + 127) / 128, C / 8, B);
    tr_attn_bwd_dq_kernel<<<grid, 128, 0, st>>>(q, k, v, o, dO, lse, L, C, scale, dq, delta);
    tr_attn_bwd_dkv_kernel<<<grid, 128, 0, st>>>(q, k, v, dO, lse, delta, L, C, scale, dk, dv);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_add(const float* a, const float* b, float* y, int64_t n, void* stream) {
    RLDM_REQUIRE(a && b && y && n >= 0, "null argument");
    if (n) tr_add_kernel<<<nblk((size_t)n), 256, 0, (hipStream_t)stream>>>(a, b, y, (size_t)n);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_copy_channels(const float* src, int src_ld, int src_off, float* dst, int dst_ld, int dst_off, int ncopy,
                             int64_t npix, int accumulate, void* stream) {
    RLDM_REQUIRE(src && dst && ncopy > 0 && npix >= 0, "null argument");
    RLDM_REQUIRE(src_off + ncopy <= src_ld && dst_off + ncopy <= dst_ld, "channel slice out of range");
    if (npix && ((src_ld | src_off | dst_ld | dst_off | ncopy) & 3) == 0 && (size_t)npix * ncopy < (1ull << 32)) {
        const unsigned total4 = (unsigned)((size_t)npix * ncopy / 4);
        tr_copy_channels_vec_kernel<<<nblk(total4), 256, 0, (hipStream_t)stream>>>(src, src_ld, src_off, dst, dst_ld, dst_off, ncopy / 4,
                                                                                 total4, accumulate);
    } else if (npix) tr_copy_channels_kernel<<<nblk((size_t)npix * ncopy), 256, 0, (hipStream_t)stream>>>(src, src_ld, src_off, dst, dst_ld,
                                                                                                   dst_off, ncopy, (size_t)npix, accumulate);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_sum2x2(const float* du, int B, int W, int H, int C, float* dx, void* stream) {
    RLDM_REQUIRE(du && dx, "null argument");
    tr_sum2x2_kernel<<<nblk((size_t)B * W * H * C), 256, 0, (hipStream_t)stream>>>(du, B, W, H, C, dx);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_silu(const float* x, const float* dy, float* y, int64_t n, int backward, int accumulate, void* stream) {
    RLDM_REQUIRE(x && y && (!backward || dy), "null argument");
    if (n) tr_silu_kernel<<<nblk((size_t)n), 256, 0, (hipStream_t)stream>>>(x, dy, y, (size_t)n, backward, accumulate);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_timestep_embedding(const int64_t* timesteps, int B, int dim, float* out, void* stream) {
    RLDM_REQUIRE(timesteps && out && dim % 2 == 0, "null argument");
    tr_timestep_embed_kernel<<<nblk((size_t)B * dim), 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const long long*>(timesteps), B,
                                                                                   dim, out);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_pack_input(const float* x, int B, int C, int W, int H, int pos_encoding, float* y, void* stream) {
    RLDM_REQUIRE(x && y, "null argument");
    tr_pack_input_kernel<<<nblk((size_t)B * W * H * (C + (pos_encoding ? 1 : 0))), 256, 0, (hipStream_t)stream>>>(
        x, B, C, W, H, pos_encoding ? 1 : 0, y);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_unpack_output(const float* x, int B, int C, int W, int H, float* y, void* stream) {
    RLDM_REQUIRE(x && y, "null argument");
    tr_unpack_kernel<<<nblk((size_t)B * W * H * C), 256, 0, (hipStream_t)stream>>>(x, B, C, W, H, y);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_mse(const float* pred, const float* target, const float* weight, int B, int C, int W, int H, float* dpred,
                   double* loss, void* stream) {
    RLDM_REQUIRE(pred && target && dpred && loss, "null argument");
    hipStream_t st = (hipStream_t)stream;
    tr_zero_kernel<<<1, 256, 0, st>>>(reinterpret_cast<float*>(loss), 2);
    tr_mse_kernel<<<nblk((size_t)B * W * H * C), 256, 0, st>>>(pred, target, weight, B, C, W, H, dpred, loss);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_sqnorm(const float* g, int64_t n, double* out, void* stream) {
    RLDM_REQUIRE(g && out && n >= 0, "null argument");
    hipStream_t st = (hipStream_t)stream;
    tr_zero_kernel<<<1, 256, 0, st>>>(reinterpret_cast<float*>(out), 2);
    if (n) tr_sqnorm_kernel<<<std::min<unsigned>(nblk((size_t)n), 2048u), 256, 0, st>>>(g, (size_t)n, out);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_hyper_step(int64_t* step_counter, const rldm_hyper_config* c, float* dyn, void* stream) {
    RLDM_REQUIRE(step_counter && c && dyn, "null argument");
    tr_hyper_kernel<<<1, 64, 0, (hipStream_t)stream>>>(reinterpret_cast<long long*>(step_counter), *c, dyn);
    TR_LAUNCH_CHECK();
    return 0;
}

static int adamw_launch(float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* ema, const double* sqnorm,
                        int64_t n, const rldm_adamw_config* c, const float* dyn, int zero_grads, void* stream);

int rldm_train_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema, const double* sqnorm,
                     int64_t n, const rldm_adamw_config* c, void* stream) {
    RLDM_REQUIRE(c && c->step >= 1, "step counts from 1 (torch.optim.AdamW)");
    return adamw_launch(params, const_cast<float*>(grads), exp_avg, exp_avg_sq, ema, sqnorm, n, c, nullptr, 0, stream);
}

int rldm_train_adamw_dyn(float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* ema, const double* sqnorm,
                         int64_t n, const rldm_adamw_config* c, const float* dyn, int zero_grads, void* stream) {
    RLDM_REQUIRE(dyn, "null argument");
    return adamw_launch(params, grads, exp_avg, exp_avg_sq, ema, sqnorm, n, c, dyn, zero_grads, stream);
}

static int adamw_launch(float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* ema, const double* sqnorm,
                        int64_t n, const rldm_adamw_config* c, const float* dyn, int zero_grads, void* stream) {
    RLDM_REQUIRE(params && grads && exp_avg && exp_avg_sq && c && n >= 0, "null argument");
    TrAdam a;
    a.dyn = dyn; a.zero_grads = zero_grads;
    a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.ema = ema; a.sqnorm = sqnorm; a.n = (size_t)n;
    a.lr = c->lr; a.b1 = c->beta1; a.b2 = c->beta2; a.eps = c->eps; a.wd = c->weight_decay;
    const int step = c->step >= 1 ? c->step : 1;
    a.bc1 = (float)(1.0 - pow((double)c->beta1, (double)step));
    a.bc2 = (float)(1.0 - pow((double)c->beta2, (double)step));
    a.max_norm = c->max_grad_norm; a.ema_decay = c->ema_decay;
    if (n) tr_adamw_kernel<<<nblk((size_t)n), 256, 0, (hipStream_t)stream>>>(a);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_pack_weights(const float* w, int N, int Cin, int taps, void* w_forward, void* w_transposed, void* stream) {
    RLDM_REQUIRE(w && w_forward && (taps == 1 || taps == 9), "null argument");
    const int Cin_pad = (Cin + 15) / 16 * 16, N_pad = (N + 15) / 16 * 16;
    const size_t n = std::max((size_t)N * taps * Cin_pad, w_transposed ? (size_t)Cin * taps * N_pad : (size_t)0);
    tr_pack_weights_kernel<<<nblk(n), 256, 0, (hipStream_t)stream>>>(w, N, Cin, taps, Cin_pad, N_pad, static_cast<bf16_t*>(w_forward),
                                                                    static_cast<bf16_t*>(w_transposed));
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_pack_weights_all(const float* params, const rldm_pack_desc* descs, int num_layers, int64_t total, void* stream) {
    RLDM_REQUIRE(params && descs && num_layers > 0 && total > 0, "null argument");
    tr_pack_all_kernel<<<nblk((size_t)total), 256, 0, (hipStream_t)stream>>>(params, descs, num_layers, (long long)total);
    TR_LAUNCH_CHECK();
    return 0;
}

int rldm_train_pack_weights_tiled(const float* params, const rldm_pack_desc* descs, int num_layers, int64_t total_tiles, void* stream) {
    RLDM_REQUIRE(params && descs && num_layers > 0 && total_tiles > 0, "null argument");
    tr_pack_tiles_kernel<<<(unsigned)total_tiles, 256, 0, (hipStream_t)stream>>>(params, descs, num_layers);
    TR_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
