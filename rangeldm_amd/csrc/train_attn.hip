// Training attention, head_dim 8, on the matrix cores: forward with log-sum-exp, dq, (dk, dv).
// diffusers Attention / AttnProcessor2_0 under `mixed_precision: bf16` (ldm/train_unconditional.py:512 inside
// accelerator.autocast): q, k, v, dO and the probabilities are rounded to bf16 into the MFMAs, accumulation, softmax
// statistics, delta and all outputs are fp32.  Activations are fp32 [B][L][C], head h = channels 8h .. 8h+7.
//
// All three kernels use the idiom of attention.hip: the 32x32 score tile comes out of one MFMA with every lane owning ONE
// column and 16 rows, the probabilities go straight back into a 32x32x16 MFMA as its B operand, and the other operand of
// that MFMA is a TRANSPOSED image of the head in LDS stored in the order the lanes hold their rows (so its fragment is one
// 16-byte read).  Only 8 of the 32 output rows carry data (head_dim 8): the kernels are exp2 / LDS bound, idle MFMA rows
// are free.
//   forward   lane = query; S^T = K Q^T (32x32x8, the running maximum rides in the C operand), O^T += [V^T; 1] P^T.
//   dq        lane = query; S^T - lse and dP^T - delta come out of two 32x32x8 MFMAs whose C operands are the lane's
//             -lse and -delta; dS = exp2(S - lse) (dP - delta); dQ^T += K^T dS^T.  Also writes delta = dO . O.
//   dk, dv    lane = key; here lse / delta vary along the ROWS of the tile, so they ride inside the contraction instead:
//             rows of [q | lse_hi lse_mid lse_lo] against the key's [k | -1 -1 -1] on a 32x32x16 MFMA (three bf16 pieces
//             carry 24 bits of lse), the same for [dO | delta pieces] x [v | -1 -1 -1]; dV^T += dO^T P, dK^T += Q^T dS.
#include "common.h"

#include <algorithm>

namespace rldm {

namespace {

constexpr float kScale = 0.35355339059327373f;          // 1 / sqrt(8)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// position of key / query `i` inside the transposed images: within every 16, the order the 32x32x16 MFMA's B operand
// holds them after a 32x32 score tile (element 4*hh + (e & 3) + 8*(e >> 2) at slot 8*hh + e)
__device__ __forceinline__ int perm_pos(int i) {
    const int j = i & 15;
    return (i & ~15) + 8 * ((j >> 2) & 1) + (j & 3) + 4 * (j >> 3);
}

__device__ __forceinline__ uint4 pack8(const float4 a, const float4 b, float s) {
    uint4 u;
    u.x = pack_bf16x2(a.x * s, a.y * s); u.y = pack_bf16x2(a.z * s, a.w * s);
    u.z = pack_bf16x2(b.x * s, b.y * s); u.w = pack_bf16x2(b.z * s, b.w * s);
    return u;
}

__device__ __forceinline__ void scatter8(bf16_t* t, int stride, int pos, const uint4 u) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int d = 0; d < 8; ++d) t[d * stride + pos] = (bf16_t)((d & 1) ? (w[d >> 1] >> 16) : (w[d >> 1] & 0xffffu));
}

// x = hi + mid + lo with three bf16 pieces (24 significant bits)
__device__ __forceinline__ void split3(float x, bf16_t* out) {
    const bf16_t h = f32_to_bf16(x);
    const float r1 = x - bf16_to_f32(h);
    const bf16_t m = f32_to_bf16(r1);
    const float r2 = r1 - bf16_to_f32(m);
    out[0] = h; out[1] = m; out[2] = f32_to_bf16(r2);
}

struct AttnArgs {
    const float* q; const float* k; const float* v; const float* o; const float* dO; const float* lse;
    float* out; float* lse_out; float* delta; float* dq; float* dk; float* dv;
    int L, Lp, C, waves;
    int ld;             // row stride (floats) of q / k / v and dq / dk / dv: C, or 3C when they are thirds of one [B][L][3C] tensor
};

// ---- forward -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void tr_attn_fwd_mfma_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = p.waves * 64, L = p.L, Lp = p.Lp, C = p.C;
    const int heads = C >> 3, qblocks = ((Lp >> 5) + p.waves - 1) / p.waves;
    int bid = blockIdx.x;
    const int qb = bid % qblocks; bid /= qblocks;
    const int h = bid % heads, b = bid / heads;
    const int l31 = lane & 31, hh = lane >> 5;
    const int vst = Lp + 8;
    bf16_t* sK = reinterpret_cast<bf16_t*>(smem);         // [Lp][8]
    bf16_t* sVt = sK + (size_t)Lp * 8;                    // [10][vst]: V^T, ones, zeros
    const int ld = p.ld;
    const size_t base = (size_t)b * L * C + h * 8;        // o
    const size_t qbase = (size_t)b * L * ld + h * 8;      // q, k, v

    for (int key = tid; key < Lp; key += NT) {
        uint4 kv = make_uint4(0u, 0u, 0u, 0u), vv = kv;
        if (key < L) {
            const float4* kp = reinterpret_cast<const float4*>(p.k + qbase + (size_t)key * ld);
            const float4* vp = reinterpret_cast<const float4*>(p.v + qbase + (size_t)key * ld);
            kv = pack8(kp[0], kp[1], 1.f);
            vv = pack8(vp[0], vp[1], 1.f);
        }
        *reinterpret_cast<uint4*>(sK + (size_t)key * 8) = kv;
        const int pos = perm_pos(key);
        scatter8(sVt, vst, pos, vv);
        sVt[8 * vst + pos] = (bf16_t)0x3f80;
        sVt[9 * vst + pos] = (bf16_t)0;
    }
    __syncthreads();
    const int q0 = (qb * p.waves + wave) * 32;
    if (q0 >= L) return;
    const int qrow = min(q0 + l31, L - 1);
    s16x4 qf;
    {
        const float4 qv = *reinterpret_cast<const float4*>(p.q + qbase + (size_t)qrow * ld + 4 * hh);
        const uint2 u = make_uint2(pack_bf16x2(qv.x * (kScale * kLog2e), qv.y * (kScale * kLog2e)),
                                   pack_bf16x2(qv.z * (kScale * kLog2e), qv.w * (kScale * kLog2e)));
        qf = __builtin_bit_cast(s16x4, u);
    }
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    const bf16_t* vrow_ptr = sVt + min(l31, 9) * vst + 8 * hh;
    const bf16_t* krow_ptr = sK + l31 * 8 + 4 * hh;
    const bool ragged = (L & 31) != 0;
    f32x16 cn;                                             // -m (log2 units), see attention.hip
    {
        const s16x4 kf = *reinterpret_cast<const s16x4*>(krow_ptr);
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        s = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(kf, qf, s, 0, 0, 0);
        if (ragged && 32 > L) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r & 3) + 8 * (r >> 2) + 4 * hh >= L) s[r] = -1e30f;
        }
        float tmax = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
#pragma unroll
        for (int r = 0; r < 16; ++r) cn[r] = -tmax;
    }
    for (int k0 = 0; k0 < Lp; k0 += 32) {
        const s16x4 kf = *reinterpret_cast<const s16x4*>(krow_ptr + k0 * 8);
        const uint4 v0 = *reinterpret_cast<const uint4*>(vrow_ptr + k0);
        const uint4 v1 = *reinterpret_cast<const uint4*>(vrow_ptr + k0 + 16);
        f32x16 s = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(kf, qf, cn, 0, 0, 0);
        if (ragged && k0 + 32 > L) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) s[r] = -1e30f;
        }
        float tmax = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
        if (__builtin_amdgcn_ballot_w64(tmax > 8.0f) != 0ull) {
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float d = fmaxf(tmax, 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] -= d; cn[r] -= d; }
#pragma unroll
            for (int r = 0; r < 5; ++r) o[r] *= alpha;
        }
        uint32_t pk[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2) pk[r >> 1] = pack_bf16x2(__builtin_amdgcn_exp2f(s[r]), __builtin_amdgcn_exp2f(s[r + 1]));
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0),
                                                    __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3])), o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1),
                                                    __builtin_bit_cast(bf16x8, make_uint4(pk[4], pk[5], pk[6], pk[7])), o, 0, 0, 0);
    }
    const float denom = __shfl(o[4], l31);                 // row 8 (the ones row) = register 4 of half 0
    const float inv = 1.0f / denom;
    if (q0 + l31 < L) {
        *reinterpret_cast<float4*>(p.out + base + (size_t)(q0 + l31) * C + 4 * hh) =
            make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
        if (hh == 0) p.lse_out[((size_t)b * heads + h) * L + q0 + l31] = (-cn[0] + __log2f(denom)) * kLn2;
    }
}

// ---- dq (+ delta) --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void tr_attn_dq_mfma_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = p.waves * 64, L = p.L, Lp = p.Lp, C = p.C;
    const int heads = C >> 3, qblocks = ((Lp >> 5) + p.waves - 1) / p.waves;
    int bid = blockIdx.x;
    const int qb = bid % qblocks; bid /= qblocks;
    const int h = bid % heads, b = bid / heads;
    const int l31 = lane & 31, hh = lane >> 5;
    const int vst = Lp + 8;
    bf16_t* sK = reinterpret_cast<bf16_t*>(smem);         // [Lp][8]
    bf16_t* sV = sK + (size_t)Lp * 8;                     // [Lp][8]
    bf16_t* sKt = sV + (size_t)Lp * 8;                    // [9][vst]: K^T, zeros
    const int ld = p.ld;
    const size_t base = (size_t)b * L * C + h * 8;        // o, dO
    const size_t qbase = (size_t)b * L * ld + h * 8;      // q, k, v, dq

    for (int key = tid; key < Lp; key += NT) {
        uint4 kv = make_uint4(0u, 0u, 0u, 0u), vv = kv;
        if (key < L) {
            const float4* kp = reinterpret_cast<const float4*>(p.k + qbase + (size_t)key * ld);
            const float4* vp = reinterpret_cast<const float4*>(p.v + qbase + (size_t)key * ld);
            kv = pack8(kp[0], kp[1], 1.f);
            vv = pack8(vp[0], vp[1], 1.f);
        }
        *reinterpret_cast<uint4*>(sK + (size_t)key * 8) = kv;
        *reinterpret_cast<uint4*>(sV + (size_t)key * 8) = vv;
        const int pos = perm_pos(key);
        scatter8(sKt, vst, pos, kv);
        sKt[8 * vst + pos] = (bf16_t)0;
    }
    __syncthreads();
    const int q0 = (qb * p.waves + wave) * 32;
    if (q0 >= L) return;
    const int qrow = min(q0 + l31, L - 1);
    const size_t ooff = base + (size_t)qrow * C + 4 * hh;
    const size_t qoff = qbase + (size_t)qrow * ld + 4 * hh;
    s16x4 qf, dof;
    float delta;
    {
        const float4 qv = *reinterpret_cast<const float4*>(p.q + qoff);
        const float4 dv = *reinterpret_cast<const float4*>(p.dO + ooff);
        const float4 ov = *reinterpret_cast<const float4*>(p.o + ooff);
        const float c = kScale * kLog2e;
        qf = __builtin_bit_cast(s16x4, make_uint2(pack_bf16x2(qv.x * c, qv.y * c), pack_bf16x2(qv.z * c, qv.w * c)));
        dof = __builtin_bit_cast(s16x4, make_uint2(pack_bf16x2(dv.x, dv.y), pack_bf16x2(dv.z, dv.w)));
        delta = dv.x * ov.x + dv.y * ov.y + dv.z * ov.z + dv.w * ov.w;
        delta += __shfl_xor(delta, 32);
    }
    const size_t srow = ((size_t)b * heads + h) * L + qrow;
    const float lse2 = p.lse[srow] * kLog2e;
    f32x16 cn, cd, acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { cn[r] = -lse2; cd[r] = -delta; acc[r] = 0.f; }
    const bf16_t* krow_ptr = sK + l31 * 8 + 4 * hh;
    const bf16_t* vrow_ptr = sV + l31 * 8 + 4 * hh;
    const bf16_t* kt_ptr = sKt + min(l31, 8) * vst + 8 * hh;
    const bool ragged = (L & 31) != 0;
    for (int k0 = 0; k0 < Lp; k0 += 32) {
        const s16x4 kf = *reinterpret_cast<const s16x4*>(krow_ptr + k0 * 8);
        const s16x4 vf = *reinterpret_cast<const s16x4*>(vrow_ptr + k0 * 8);
        const uint4 t0 = *reinterpret_cast<const uint4*>(kt_ptr + k0);
        const uint4 t1 = *reinterpret_cast<const uint4*>(kt_ptr + k0 + 16);
        f32x16 s = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(kf, qf, cn, 0, 0, 0);          // s - lse   (log2 units)
        const f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(vf, dof, cd, 0, 0, 0);  // dP - delta
        if (ragged && k0 + 32 > L) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) s[r] = -1e30f;
        }
        uint32_t pk[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2)
            pk[r >> 1] = pack_bf16x2(__builtin_amdgcn_exp2f(s[r]) * dp[r], __builtin_amdgcn_exp2f(s[r + 1]) * dp[r + 1]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, t0),
                                                      __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3])), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, t1),
                                                      __builtin_bit_cast(bf16x8, make_uint4(pk[4], pk[5], pk[6], pk[7])), acc, 0, 0, 0);
    }
    if (q0 + l31 < L) {
        *reinterpret_cast<float4*>(p.dq + qoff) = make_float4(acc[0] * kScale, acc[1] * kScale, acc[2] * kScale, acc[3] * kScale);
        if (hh == 0) p.delta[srow] = delta;
    }
}

// ---- dk, dv --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void tr_attn_dkv_mfma_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = p.waves * 64, L = p.L, Lp = p.Lp, C = p.C;
    const int heads = C >> 3, kblocks = ((Lp >> 5) + p.waves - 1) / p.waves;
    int bid = blockIdx.x;
    const int kb = bid % kblocks; bid /= kblocks;
    const int h = bid % heads, b = bid / heads;
    const int l31 = lane & 31, hh = lane >> 5;
    const int vst = Lp + 8;
    bf16_t* sQ = reinterpret_cast<bf16_t*>(smem);         // [Lp][16]: q (scaled) | lse pieces | 0
    bf16_t* sD = sQ + (size_t)Lp * 16;                    // [Lp][16]: dO | delta pieces | 0
    bf16_t* sQt = sD + (size_t)Lp * 16;                   // [9][vst]: Q^T, zeros
    bf16_t* sDt = sQt + 9 * vst;                          // [9][vst]: dO^T, zeros
    const int ld = p.ld;
    const size_t base = (size_t)b * L * C + h * 8;        // dO
    const size_t qbase = (size_t)b * L * ld + h * 8;      // q, k, v, dk, dv
    const size_t sbase = ((size_t)b * heads + h) * L;

    for (int i = tid; i < Lp; i += NT) {
        uint4 qv = make_uint4(0u, 0u, 0u, 0u), dv = qv;
        bf16_t e0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, e1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (i < L) {
            const float4* qp = reinterpret_cast<const float4*>(p.q + qbase + (size_t)i * ld);
            const float4* dp = reinterpret_cast<const float4*>(p.dO + base + (size_t)i * C);
            qv = pack8(qp[0], qp[1], kScale * kLog2e);
            dv = pack8(dp[0], dp[1], 1.f);
            split3(p.lse[sbase + i] * kLog2e, e0);
            split3(p.delta[sbase + i], e1);
        } else {
            e0[0] = f32_to_bf16(1e30f);                   // padded query: s = -1e30, p = 0
        }
        *reinterpret_cast<uint4*>(sQ + (size_t)i * 16) = qv;
        *reinterpret_cast<uint4*>(sQ + (size_t)i * 16 + 8) =
            make_uint4((uint32_t)e0[0] | ((uint32_t)e0[1] << 16), (uint32_t)e0[2], 0u, 0u);
        *reinterpret_cast<uint4*>(sD + (size_t)i * 16) = dv;
        *reinterpret_cast<uint4*>(sD + (size_t)i * 16 + 8) =
            make_uint4((uint32_t)e1[0] | ((uint32_t)e1[1] << 16), (uint32_t)e1[2], 0u, 0u);
        const int pos = perm_pos(i);
        scatter8(sQt, vst, pos, qv);
        scatter8(sDt, vst, pos, dv);
        sQt[8 * vst + pos] = (bf16_t)0;
        sDt[8 * vst + pos] = (bf16_t)0;
    }
    __syncthreads();
    const int key0 = (kb * p.waves + wave) * 32;
    if (key0 >= L) return;
    const int krow = min(key0 + l31, L - 1);
    // B operands: half 0 = the key's k / v, half 1 = (-1, -1, -1, 0 ...) against the lse / delta pieces
    uint4 kB = make_uint4(0xbf80bf80u, 0x0000bf80u, 0u, 0u), vB = kB;
    if (hh == 0) {
        const float4* kp = reinterpret_cast<const float4*>(p.k + qbase + (size_t)krow * ld);
        const float4* vp = reinterpret_cast<const float4*>(p.v + qbase + (size_t)krow * ld);
        kB = pack8(kp[0], kp[1], 1.f);
        vB = pack8(vp[0], vp[1], 1.f);
    }
    f32x16 accK, accV, zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accK[r] = 0.f; accV[r] = 0.f; zero[r] = 0.f; }
    const bf16_t* qrow_ptr = sQ + l31 * 16 + 8 * hh;
    const bf16_t* drow_ptr = sD + l31 * 16 + 8 * hh;
    const bf16_t* qt_ptr = sQt + min(l31, 8) * vst + 8 * hh;
    const bf16_t* dt_ptr = sDt + min(l31, 8) * vst + 8 * hh;
    for (int q0 = 0; q0 < Lp; q0 += 32) {
        const uint4 qa = *reinterpret_cast<const uint4*>(qrow_ptr + q0 * 16);
        const uint4 da = *reinterpret_cast<const uint4*>(drow_ptr + q0 * 16);
        const uint4 qt0 = *reinterpret_cast<const uint4*>(qt_ptr + q0), qt1 = *reinterpret_cast<const uint4*>(qt_ptr + q0 + 16);
        const uint4 dt0 = *reinterpret_cast<const uint4*>(dt_ptr + q0), dt1 = *reinterpret_cast<const uint4*>(dt_ptr + q0 + 16);
        // rows = queries q0 + (r & 3) + 8 (r >> 2) + 4 hh, column = this lane's key
        const f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa), __builtin_bit_cast(bf16x8, kB), zero, 0, 0, 0);
        const f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, da), __builtin_bit_cast(bf16x8, vB), zero, 0, 0, 0);
        uint32_t pp[8], ps[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float p0 = __builtin_amdgcn_exp2f(s[r]), p1 = __builtin_amdgcn_exp2f(s[r + 1]);
            pp[r >> 1] = pack_bf16x2(p0, p1);
            ps[r >> 1] = pack_bf16x2(p0 * dp[r], p1 * dp[r + 1]);
        }
        accV = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, dt0),
                                                       __builtin_bit_cast(bf16x8, make_uint4(pp[0], pp[1], pp[2], pp[3])), accV, 0, 0, 0);
        accV = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, dt1),
                                                       __builtin_bit_cast(bf16x8, make_uint4(pp[4], pp[5], pp[6], pp[7])), accV, 0, 0, 0);
        accK = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qt0),
                                                       __builtin_bit_cast(bf16x8, make_uint4(ps[0], ps[1], ps[2], ps[3])), accK, 0, 0, 0);
        accK = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qt1),
                                                       __builtin_bit_cast(bf16x8, make_uint4(ps[4], ps[5], ps[6], ps[7])), accK, 0, 0, 0);
    }
    if (key0 + l31 < L) {
        const size_t off = qbase + (size_t)(key0 + l31) * ld + 4 * hh;
        *reinterpret_cast<float4*>(p.dk + off) = make_float4(accK[0] * kLn2, accK[1] * kLn2, accK[2] * kLn2, accK[3] * kLn2);
        *reinterpret_cast<float4*>(p.dv + off) = make_float4(accV[0], accV[1], accV[2], accV[3]);
    }
}

int pick_waves(int Lp) { return std::min(8, Lp >> 5); }

}  // namespace

int tr_attention_forward_mfma(const float* q, const float* k, const float* v, int ld, int B, int L, int C, float* o, float* lse,
                              hipStream_t st) {
    AttnArgs a{};
    a.q = q; a.k = k; a.v = v; a.out = o; a.lse_out = lse; a.L = L; a.C = C; a.ld = ld;
    a.Lp = (L + 31) / 32 * 32;
    a.waves = pick_waves(a.Lp);
    const int qblocks = ((a.Lp >> 5) + a.waves - 1) / a.waves;
    const size_t smem = (size_t)a.Lp * 16 + (size_t)10 * (a.Lp + 8) * 2;
    RLDM_REQUIRE(smem <= 160 * 1024, "sequence too long for the LDS-resident head");
    static bool attr = false;
    if (!attr) {
        RLDM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tr_attn_fwd_mfma_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    tr_attn_fwd_mfma_kernel<<<dim3((unsigned)(B * (C / 8) * qblocks)), a.waves * 64, smem, st>>>(a);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int tr_attention_backward_mfma(const float* q, const float* k, const float* v, int ld, const float* o, const float* dO,
                               const float* lse, int B, int L, int C, float* delta, float* dq, float* dk, float* dv, hipStream_t st) {
    AttnArgs a{};
    a.ld = ld;
    a.q = q; a.k = k; a.v = v; a.o = o; a.dO = dO; a.lse = lse; a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv; a.L = L; a.C = C;
    a.Lp = (L + 31) / 32 * 32;
    a.waves = pick_waves(a.Lp);
    const int blocks = ((a.Lp >> 5) + a.waves - 1) / a.waves;
    const size_t smem_dq = (size_t)a.Lp * 32 + (size_t)9 * (a.Lp + 8) * 2;
    const size_t smem_dkv = (size_t)a.Lp * 64 + (size_t)18 * (a.Lp + 8) * 2;
    RLDM_REQUIRE(smem_dkv <= 160 * 1024, "sequence too long for the LDS-resident head");
    static bool attr = false;
    if (!attr) {
        RLDM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tr_attn_dq_mfma_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        RLDM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tr_attn_dkv_mfma_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const dim3 grid((unsigned)(B * (C / 8) * blocks));
    tr_attn_dq_mfma_kernel<<<grid, a.waves * 64, smem_dq, st>>>(a);
    tr_attn_dkv_mfma_kernel<<<grid, a.waves * 64, smem_dkv, st>>>(a);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace rldm
