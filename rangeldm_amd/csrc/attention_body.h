// Device bodies of the fused attention launch (attention.hip), shared with the persistent trunk kernel (trunk.hip): the key loop on
// a given q / k / v (attention_tile) and the second-generation GroupNorm + q/k/v projection + softmax . V kernel as a function.
// TRUNK = true: a phase of the persistent launch -- x was published by the previous phase (read past the L1, behind the cluster
// wait), the phase ends by arriving on the cluster's counter.
#pragma once
#include "kernels.h"
#include "trunk_seam.h"

#include <type_traits>

#ifndef RLDM_ATTN_WEAVE
#define RLDM_ATTN_WEAVE 1           /* 0: the round-2 block form of the two-chain key loop (A/B builds) */
#endif

namespace rldm {

// One 32-query tile against all Lp keys staged in LDS (sK rows, sVt = V^T in consumption order + ones + zero rows).
// qf: B operand of S^T for lane (query l31, half hh) = q[query][4*hh .. 4*hh+3] (pre-scaled by log2(e)/sqrt(8)).
__device__ __forceinline__ void attention_tile(const bf16_t* sK, const bf16_t* sVt, int vst, int L, int Lp, int C, int q0,
                                               s16x4 qf, bf16_t* out_bh, int l31, int hh) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;

    // A operand rows of the PV MFMA: row l31 < 8 -> V^T[d = l31], row 8 -> ones, rows 9..31 -> the zero row (one address)
    const bf16_t* vrow_ptr = sVt + min(l31, 9) * vst + 8 * hh;
    const bf16_t* krow_ptr = sK + l31 * 8 + 4 * hh;
    const bool ragged = (L & 31) != 0;

    // Running maximum m of the query (log2 units), kept as the MFMA's C operand: s = k.q - m comes out of the matrix core.
    // m is exact after the first tile and afterwards only raised when some score of the wave exceeds it by more than 8
    // (p <= 2^8 then: harmless in fp32 / bf16), so the rescale of o, the subtraction and the refresh of C are rare.
    f32x16 cn;
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
        float tmax = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) tmax = fmaxf(fmaxf(tmax, s[r]), s[r + 1]);
        tmax = fmaxf(tmax, s[15]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
#pragma unroll
        for (int r = 0; r < 16; ++r) cn[r] = -tmax;
    }

    for (int k0 = 0; k0 < Lp; k0 += 32) {
        const s16x4 kf = *reinterpret_cast<const s16x4*>(krow_ptr + k0 * 8);
        const uint4 v0 = *reinterpret_cast<const uint4*>(vrow_ptr + k0);
        const uint4 v1 = *reinterpret_cast<const uint4*>(vrow_ptr + k0 + 16);
        f32x16 s = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(kf, qf, cn, 0, 0, 0);
        // lane (query l31, half hh), register r <-> key k0 + (r&3) + 8*(r>>2) + 4*hh ; scores are in log2 units, relative to m
        if (ragged && k0 + 32 > L) {                        // last tile: keys >= L get -inf scores
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) s[r] = -1e30f;
        }
        float tmax = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) tmax = fmaxf(fmaxf(tmax, s[r]), s[r + 1]);
        tmax = fmaxf(tmax, s[15]);
        if (__builtin_amdgcn_ballot_w64(tmax > 8.0f) != 0ull) {
            // raise m (both halves of a query agree on it), rescale what has been accumulated
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float d = fmaxf(tmax, 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] -= d; cn[r] -= d; }
#pragma unroll
            for (int r = 0; r < 5; ++r) o[r] *= alpha;      // rows 0..8 only: V rows and the ones row (others stay 0)
        }
        uint32_t pk[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2) pk[r >> 1] = pack_bf16x2(__builtin_amdgcn_exp2f(s[r]), __builtin_amdgcn_exp2f(s[r + 1]));
        // PV: MFMA t (t = 0, 1) contracts over the 16 keys {k0 + 16t + 4hh' + (e&3) + 8(e>>2)}, e = 0..7, hh' = 0, 1
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0),
                                                    __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3])), o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1),
                                                    __builtin_bit_cast(bf16x8, make_uint4(pk[4], pk[5], pk[6], pk[7])), o, 0, 0, 0);
    }
    // o rows: reg r of half hh <-> row (r&3) + 8*(r>>2) + 4*hh.  d = 4*hh + r for r < 4; denominator = row 8 = reg 4 of hh 0
    float denom = __shfl(o[4], l31);
    const float inv = 1.0f / denom;
    uint2 ov;
    ov.x = pack_bf16x2(o[0] * inv, o[1] * inv);
    ov.y = pack_bf16x2(o[2] * inv, o[3] * inv);
    if (q0 + l31 < L) *reinterpret_cast<uint2*>(out_bh + (size_t)(q0 + l31) * C + 4 * hh) = ov;
}

// ---- output projection inside the attention launch (round 3) --------------------------------------------------------------------
// y = to_out(o) + bias + x for the 64-pixel block `slot` of image b, by the workgroup that just finished head group `slot`:
// once ALL R head groups of the image have published their 8-channel column slices of o (arrive / wait on the image's counter --
// trunk_seam.h's protocol; the image's workgroups share an XCD, checked), the block's rows of o are complete in the XCD's L2.
// Waves 0 .. 2 C/32 - 1 each own one 32 x 32 output sub-tile (K = C: <= 8 k-steps, weight fragments requested BEFORE the seam);
// the fp32 tile meets the residual in LDS, is rounded, stored with 16-byte rows, and its per-channel (sum, sumsq) become row
// `slot` of y's statistics partials -- conv_small.hip's pointwise epilogue, so the consumers read y like any conv output.
// Replaces a launch of its own (12.4 us in the sampler's graph at 1024 tokens: argument fetch, tile + weight round trips,
// k-group exchange) by one seam and ~4 us of tail.
// SEAM = false: the same tail as a launch of its own behind the attention launch (what the per-layer fall-back of a sampler and
// the A/B switches run: the same code on the same operands in the same order, so the results are identical bit for bit).
template <bool SEAM>
__device__ __forceinline__ void attention_proj_tail(const AttnQkvParams& p, const int b, const int slot, const int R, const int NT) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PB = 64;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = p.C, L = p.L;
    const int nks = C >> 4, nct = C >> 5, c8n = C >> 3;
    const int pt = wave & 1, nt = wave >> 1;
    const bool gemm_wave = nt < nct;
    unsigned* const counter = SEAM ? p.proj_counter + b * 32 : nullptr;
    const unsigned epoch = SEAM ? counter[3] : 0u;        // launches so far (advanced by slot 0 at the very end)
    bf16x8 wf[8];
    f32x16 acc;
    if (gemm_wave) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            if (ks < nks) wf[ks] = *reinterpret_cast<const bf16x8*>(p.proj_w + ((size_t)(nt * nks + ks) * 64 + lane) * 8);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const float4 bv = *reinterpret_cast<const float4*>(p.proj_bias + 32 * nt + 8 * r4 + 4 * hh);
            acc[r4 * 4 + 0] = bv.x; acc[r4 * 4 + 1] = bv.y; acc[r4 * 4 + 2] = bv.z; acc[r4 * 4 + 3] = bv.w;
        }
    }
    if constexpr (SEAM) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        if (slot == 0 && tid == 0) __hip_atomic_store(counter + 1, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        TrunkSeam seam = {};
        seam.counter = counter;
        seam.wait_for = (epoch + 1u) * (unsigned)R;
        seam.has_wait = 1;
        seam.error = p.proj_error;
        trunk_arrive(seam, tid);                          // this head group's rows of o are in the XCD's L2
        trunk_wait(seam, tid);                            // ... and everybody else's
        if (slot != 0 && tid == 0) {
            const unsigned x0 = __hip_atomic_load(counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (x0 != xcc + 1u) *p.proj_error = 2;        // not on slot 0's XCD: the rows read below may be stale
        }
    }
    const int ORS = C * 2 + 16, FRS = C * 4 + 16;
    unsigned char* sO = smem;                             // [64][C] bf16: the block's rows of o, later the rounded output tile
    unsigned char* sF = sO + PB * ORS;                    // [64][C] fp32: projection + bias
    float* sS = reinterpret_cast<float*>(sF + PB * FRS);  // [pixel groups][2][C]
    const size_t row0 = (size_t)b * L + (size_t)slot * PB;
    for (int q = tid; q < PB * c8n; q += NT) {
        const int px = q / c8n, c8 = q - px * c8n;
        *reinterpret_cast<uint4*>(sO + px * ORS + c8 * 16) = *reinterpret_cast<const uint4*>(p.out + (row0 + px) * C + c8 * 8);
    }
    __syncthreads();
    if (gemm_wave) {
        const unsigned char* xrow = sO + (pt * 32 + l31) * ORS + hh * 16;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            if (ks < nks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], *reinterpret_cast<const bf16x8*>(xrow + ks * 32), acc, 0, 0, 0);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
            *reinterpret_cast<float4*>(sF + (pt * 32 + l31) * FRS + (32 * nt + 8 * r4 + 4 * hh) * 4) =
                make_float4(acc[r4 * 4 + 0], acc[r4 * 4 + 1], acc[r4 * 4 + 2], acc[r4 * 4 + 3]);
    }
    __syncthreads();
    for (int q = tid; q < PB * c8n; q += NT) {
        const int px = q / c8n, c8 = q - px * c8n;
        const uint4 rq = *reinterpret_cast<const uint4*>(p.proj_res + (row0 + px) * C + c8 * 8);
        const float4 v0 = *reinterpret_cast<const float4*>(sF + px * FRS + c8 * 32);
        const float4 v1 = *reinterpret_cast<const float4*>(sF + px * FRS + c8 * 32 + 16);
        uint4 o;
        o.x = pack_bf16x2(bf16lo(rq.x) + v0.x, bf16hi(rq.x) + v0.y); o.y = pack_bf16x2(bf16lo(rq.y) + v0.z, bf16hi(rq.y) + v0.w);
        o.z = pack_bf16x2(bf16lo(rq.z) + v1.x, bf16hi(rq.z) + v1.y); o.w = pack_bf16x2(bf16lo(rq.w) + v1.z, bf16hi(rq.w) + v1.w);
        *reinterpret_cast<uint4*>(p.proj_y + (row0 + px) * C + c8 * 8) = o;
        *reinterpret_cast<uint4*>(sO + px * ORS + c8 * 16) = o;
    }
    __syncthreads();
    // statistics of the ROUNDED tile: thread = (channel pair, pixel group); fixed summation order
    const int NCP = C >> 1, NG = NT / NCP, PPG = PB / NG;
    if (PPG >= 1 && tid < NG * NCP) {
        const int cp = tid % NCP, pg = tid / NCP;
        float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
        for (int j = 0; j < PPG; ++j) {
            const uint32_t w2 = *reinterpret_cast<const uint32_t*>(sO + (pg * PPG + j) * ORS + cp * 4);
            const float a0 = bf16lo(w2), a1 = bf16hi(w2);
            s0 += a0; s1 += a1;
            q0 += a0 * a0; q1 += a1 * a1;
        }
        *reinterpret_cast<float2*>(sS + (pg * 2 + 0) * C + cp * 2) = make_float2(s0, s1);
        *reinterpret_cast<float2*>(sS + (pg * 2 + 1) * C + cp * 2) = make_float2(q0, q1);
    }
    __syncthreads();
    if (tid < 2 * C) {
        const int kind = tid / C, c = tid - kind * C;
        float S = 0.f;
        for (int g = 0; g < NG; ++g) S += sS[(g * 2 + kind) * C + c];
        reinterpret_cast<float*>(p.proj_stats + ((size_t)b * R + slot) * C + c)[kind] = S;
    }
    if (SEAM && slot == 0 && tid == 0) counter[3] = epoch + 1u;
}

// =====================================================================================================================
// Second generation of the fused launch (round 2).  Same contract as attention_qkv_d8_kernel; what changed and why
// (per-wave s_memtime stamps of both generations: tools/attn_timeline2.py, docs/history/profiles/round2_attn_timeline.txt):
//   * The prologue of the first generation was FIVE dependent global round trips of ~2 us each (statistics partials ->
//     gamma / beta -> the head's weight fragments -> its bias -> the x rows of the wave's second query tile): 12 of the 44 us of
//     an L = 1024 launch, 6 of the 7 us of an L = 64 one.  None of them depends on another: every one is now requested in the
//     first instructions of the kernel and consumed from registers.
//   * S^T rides on v_mfma_f32_32x32x16_bf16 with the contraction padded from 8 to 16: slot 8 of the key side is 1.0 and slot 8 of
//     the query side is -m (the softmax stabiliser of that query, rounded to bf16 -- the SAME rounded value for every key, so it
//     cancels in the normalisation).  The matrix core then delivers s - m directly with a zero C operand: no 16-register copy of
//     -m per query tile, no per-tile maximum, ballot, branch or rescale in the key loop -- per 32 x 32 score tile the VALU work is
//     16 v_exp_f32 + 8 v_cvt_pk_bf16_f32.
//   * m = the exact maximum over the first 32 keys.  Later keys may beat it by up to ~2^100 before an fp32 accumulator overflows
//     (bf16 has fp32's exponent range, so P = 2^(s-m) > 1 is representable); a wave whose denominators are not finite afterwards
//     (a score beat the first tile's maximum by more than ~100 in log2 units) redoes its tiles with the running-maximum loop of
//     attention_tile.  The denominator is >= 1 by construction (the maximal key of the first tile contributes exactly 1).
//   * a wave runs TWO query tiles against each key tile (L = 1024): the K fragment and the two V^T fragments are read from LDS
//     once for both, and the loop is ROTATED -- S(i+1) of a chain is issued right behind its PV(i) MFMAs, so it executes on the
//     matrix pipe while the VALU does the other chain's exponentials.  In the first generation the oldest wave of a SIMD spent
//     ~240 of its 560 cycles per iteration waiting on its own LDS -> S -> exp -> PV chain, and the younger waves only got the
//     leftover issue slots (the four waves of a SIMD finished one after the other, 12 k cycles apart).
//   * HG heads of one image share a workgroup when a head alone has fewer than 16 query tiles: the GroupNorm fold is done once
//     per workgroup and the heads' x rows come from the CU's L1 after the first head fetched them.
// Layouts (LDS): sK [HG][Lp][8] bf16 (16 B per key), sVt [HG][10][Lp + 8] (as above), sC = one 16-byte row {1.0, 0, ...} (slots
// 8..15 of every key), sW [HG][C/16][64][8], sBp [HG][32].
template <int PAIR, bool TRUNK>
__device__ __forceinline__ void attention_qkv2_body(const AttnQkvParams& p, const int waves, const int Lp, const int HG, const int b,
                                                    const int hg, const TrunkSeam& seam) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef RLDM_ABLATE
    unsigned long long tsv[8];
    int tsn = 0;
    const unsigned long long t_real0 = __builtin_amdgcn_s_memrealtime();
#define RLDM_ASTAMP() if (tsn < 8) tsv[tsn++] = __builtin_amdgcn_s_memtime()
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#else
#define RLDM_ASTAMP()
#endif
    RLDM_ASTAMP();
    int tid_ = threadIdx.x;
    if constexpr (TRUNK) asm volatile("" : "+v"(tid_));     // (opaque per phase: see conv_small_body.h)
    const int tid = tid_, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = waves * 64;
    const int heads = p.C >> 3;
    (void)heads;
    const int l31 = lane & 31, hh = lane >> 5;
    const int C = p.C, L = p.L;
    const int vst = Lp + 8;
    const int ntiles = Lp >> 5;
    const int nks = C >> 4;
    const int wph = waves / HG;                           // waves per head
    const int hl = wave / wph, wl = wave % wph;           // this wave's head (local) and its index among the head's waves
    const int h = hg * HG + hl;

    bf16_t* sK = reinterpret_cast<bf16_t*>(smem);                       // [HG][Lp][8]
    bf16_t* sVt = sK + (size_t)HG * Lp * 8;                             // [HG][10][vst]
    bf16_t* sC = sVt + (size_t)HG * 10 * vst;                           // [8]: {1, 0, 0, 0, 0, 0, 0, 0}
    float* sGa = reinterpret_cast<float*>(sC + 8);                      // [C]
    float* sGs = sGa + C;
    double* sD = reinterpret_cast<double*>(sGs + C);                    // [2][C] scratch (>= HG * nks * 64 floats)
    const int scratch = max(2 * C * 8, HG * nks * 64 * 4);              // bytes of sD / sPart (the partial sums of b' need more for HG > 1)
    bf16_t* sW = reinterpret_cast<bf16_t*>(reinterpret_cast<unsigned char*>(sD) + scratch);    // [HG][C/16][64 lanes][8]
    float* sBp = reinterpret_cast<float*>(sW + (size_t)HG * nks * 512); // [HG][32]
    // x staging tile of this wave: [32 rows][128 bytes + 16 pad] -- 64 channels of a query tile at a time (see the projection)
    constexpr int XRS = kAttnXRowBytes;
    unsigned char* sXw = reinterpret_cast<unsigned char*>(sBp + HG * 32) + (size_t)wave * kAttnXStageBytes;

    // ---- every global read of the prologue, requested up front (one memory round trip instead of five) ----------------------
    constexpr int TPW = PAIR ? 2 : 1;
    constexpr int KB = 8;                                 // k-steps (16 channels each) of x held in registers per tile
    const int T0 = wl;                                    // query tiles of this wave: T0 (and T0 + wph)
    const bool own_ch = tid < C;                          // (C <= 512 <= NT is not guaranteed: channels beyond NT loop below)
    float g_pre = 0.f, b_pre = 0.f;
    const bool prenorm = p.st == nullptr;                 // x arrives normalised (producer-side GroupNorm): W' = W, b' = b
    if (own_ch && !prenorm) { g_pre = p.gamma[tid]; b_pre = p.beta[tid]; }
    const bf16_t* wf_ptr = p.wfrag + (size_t)(hg * HG) * nks * 512;
    const int npieces = HG * nks * 64;
    constexpr int NPW = TRUNK ? 8 : 2;                    // weight pieces held per thread up front (more: loaded in the loop -- one
                                                          // dependent L2 round trip each: a trunk phase holds all 8 of a 4-head group)
    uint4 wpre[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j)
        if (tid + j * NT < npieces) wpre[j] = *reinterpret_cast<const uint4*>(wf_ptr + (size_t)(tid + j * NT) * 8);
    float bias_pre = 0.f;
    if (tid < 32 * HG && (tid & 31) < 24) bias_pre = p.bias[(hg * HG + (tid >> 5)) * 32 + (tid & 31)];
    // (persistent trunk: the weights above do not depend on the previous phase; x does)
    if constexpr (TRUNK) trunk_wait(seam, tid);
    // x rows of this wave's query tiles.  The projection's B operand wants lane = (pixel, 16 bytes): read like that, a wave
    // instruction touches 32 different 128-byte lines for 32 bytes each, every line is visited by four instructions, and the
    // CU's L1 (one line per ~2 clocks) made the prologue of an L = 1024 launch ~24 k cycles for 256 KB of x per workgroup.
    // With C % 64 == 0 the rows are therefore requested COALESCED -- group g = channels [64 g, 64 g + 64) of the 32 rows of a
    // tile = 4 instructions of 8 whole 128-byte segments each -- and transposed to the fragment layout through a 4.5 KB LDS
    // tile of the wave's own (conflict-free both ways; no barrier: a wave's LDS instructions execute in order).
    // (C % 64 != 0 -- toy configurations -- runs on the first-generation kernel: launch_attention_qkv2 declines)
    const int xr8 = lane >> 3, xc8 = lane & 7;           // coalesced piece (i, lane): row 8 i + xr8 of the tile, 16-byte chunk xc8
    bf16x8 xpre[TPW][KB];
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const bf16_t* xt = p.x + ((size_t)b * L) * C + xc8 * 8;
#pragma unroll
        for (int j = 0; j < KB; ++j)
            if ((j & ~3) < nks) {
                const int row = min((T0 + ti * wph) * 32 + (j & 3) * 8 + xr8, L - 1);
                xpre[ti][j] = ld_act_frag<TRUNK>(xt + (size_t)row * C + (j >> 2) * 64);
            }
    }

    // ---- GroupNorm affine of image b (conv_igemm.hip's arithmetic; one fold per workgroup) ----------------------------------
    if (prenorm) {
        // x arrives normalised: W' = W and b' = b exactly (what the fold below computes with a = 1, s = 0), so the head's weight
        // fragments are copied to LDS as they are -- no affine, no per-row bias sums, one barrier instead of three
        if (tid < 8) sC[tid] = tid == 0 ? (bf16_t)0x3f80 : (bf16_t)0;
        for (int q = tid, j = 0; q < npieces; q += NT, ++j) {
            uint4 w;
            bool have = false;
#pragma unroll
            for (int k = 0; k < NPW; ++k)
                if (j == k) { w = wpre[k]; have = true; }
            if (!have) w = *reinterpret_cast<const uint4*>(wf_ptr + (size_t)q * 8);
            *reinterpret_cast<uint4*>(sW + (size_t)q * 8) = w;
        }
        if (tid < 32 * HG) sBp[tid] = bias_pre;
        __syncthreads();
        RLDM_ASTAMP();                                    // 1
        RLDM_ASTAMP();                                    // 2: W, b in LDS
    } else {
        const int cpg = C / p.groups;
        for (int t = tid; t < C; t += NT) {
            const float2* src = p.st + (size_t)b * p.P * C + t;
            double S = 0.0, SS = 0.0;
            int q = 0;
            for (; q + 8 <= p.P; q += 8) {
                float2 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = ld_act8<TRUNK>(src + (size_t)(q + j) * C);
#pragma unroll
                for (int j = 0; j < 8; ++j) { S += (double)v[j].x; SS += (double)v[j].y; }
            }
            for (; q + 4 <= p.P; q += 4) {                // (P = 4 at the 64x4 level: one round trip, not four dependent ones)
                float2 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = ld_act8<TRUNK>(src + (size_t)(q + j) * C);
#pragma unroll
                for (int j = 0; j < 4; ++j) { S += (double)v[j].x; SS += (double)v[j].y; }
            }
            for (; q < p.P; ++q) {
                const float2 v = ld_act8<TRUNK>(src + (size_t)q * C);
                S += (double)v.x;
                SS += (double)v.y;
            }
            sD[t] = S;
            sD[C + t] = SS;
        }
        if (tid < 8) sC[tid] = tid == 0 ? (bf16_t)0x3f80 : (bf16_t)0;
        __syncthreads();
        for (int t = tid; t < C; t += NT) {
            const int g0 = ((t * p.magic_cpg) >> 20) * cpg;
            double S = 0.0, SS = 0.0;
            for (int i = 0; i < cpg; ++i) {
                S += sD[g0 + i];
                SS += sD[C + g0 + i];
            }
            const double mean = S * (double)p.inv_n;
            double var = SS * (double)p.inv_n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const float gm = t == tid ? g_pre : p.gamma[t], bt = t == tid ? b_pre : p.beta[t];
            const float a = gm * __builtin_amdgcn_rsqf((float)var + p.eps);
            sGa[t] = a;
            sGs[t] = bt - (float)mean * a;
        }
        __syncthreads();
    RLDM_ASTAMP();                                        // 1: GroupNorm affine in LDS

    // ---- W' = W_h * diag(a) (bf16, A-fragment order), b' = b_h + W_h * s for the HG heads of this workgroup -----------------
    float* sPart = reinterpret_cast<float*>(sD);
    for (int q = tid, j = 0; q < npieces; q += NT, ++j) { // piece q = (head, k-step, lane): row q & 31, channels 16*ks + 8*(lane >> 5) ..
        const int c0 = ((q >> 6) % nks) * 16 + ((q >> 5) & 1) * 8;
        uint4 w;
        bool have = false;
#pragma unroll
        for (int k = 0; k < NPW; ++k)
            if (j == k) { w = wpre[k]; have = true; }
        if (!have) w = *reinterpret_cast<const uint4*>(wf_ptr + (size_t)q * 8);
        const float4 a0 = *reinterpret_cast<const float4*>(sGa + c0), a1 = *reinterpret_cast<const float4*>(sGa + c0 + 4);
        const float4 s0 = *reinterpret_cast<const float4*>(sGs + c0), s1 = *reinterpret_cast<const float4*>(sGs + c0 + 4);
        uint4 n;
        n.x = pack_bf16x2(bf16lo(w.x) * a0.x, bf16hi(w.x) * a0.y);
        n.y = pack_bf16x2(bf16lo(w.y) * a0.z, bf16hi(w.y) * a0.w);
        n.z = pack_bf16x2(bf16lo(w.z) * a1.x, bf16hi(w.z) * a1.y);
        n.w = pack_bf16x2(bf16lo(w.w) * a1.z, bf16hi(w.w) * a1.w);
        *reinterpret_cast<uint4*>(sW + (size_t)q * 8) = n;
        sPart[q] = bf16lo(w.x) * s0.x + bf16hi(w.x) * s0.y + bf16lo(w.y) * s0.z + bf16hi(w.y) * s0.w +
                   bf16lo(w.z) * s1.x + bf16hi(w.z) * s1.y + bf16lo(w.w) * s1.z + bf16hi(w.w) * s1.w;
    }
    __syncthreads();
    if (tid < 32 * HG) {                                  // fixed summation order
        const int hq = tid >> 5, row = tid & 31;
        float acc_b = bias_pre;
        for (int j = 0; j < 2 * nks; ++j) acc_b += sPart[(hq * 2 * nks + j) * 32 + row];
        sBp[tid] = acc_b;
    }
    __syncthreads();
    RLDM_ASTAMP();                                        // 2: W', b' in LDS
    }

    // ---- projection of this wave's pixel tiles: q stays in registers, k / v go to the head's LDS image ----------------------
    bf16_t* sKh = sK + (size_t)hl * Lp * 8;
    bf16_t* sVh = sVt + (size_t)hl * 10 * vst;
    const bf16_t* sWh = sW + (size_t)hl * nks * 512;
    float binit[12];
#pragma unroll
    for (int r = 0; r < 12; ++r) binit[r] = sBp[hl * 32 + 8 * (r >> 2) + 4 * hh + (r & 3)];
    uint4 qB[TPW];                                        // B operand of S^T: lanes 0..31 q[0..7] of the query, lanes 32..63 {-m, 0...}
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int T = T0 + ti * wph;
        qB[ti] = make_uint4(0u, 0u, 0u, 0u);
        if (T >= ntiles) continue;
        f32x16 acc;                                       // (keys / queries past L: a clamped row, masked later)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = r < 12 ? binit[r] : 0.f;
        for (int k0 = 0; k0 < nks; k0 += KB) {
            bf16x8 xv[KB];
            const bf16_t* xt = p.x + ((size_t)b * L) * C + xc8 * 8;
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                if (k0 == 0) xv[j] = xpre[ti][j];
                else if (k0 + (j & ~3) < nks)
                    xv[j] = ld_act_frag<TRUNK>(xt + (size_t)min(T * 32 + (j & 3) * 8 + xr8, L - 1) * C + ((k0 + j) >> 2) * 64);
            }
#pragma unroll
            for (int g = 0; g < KB / 4; ++g) {
                if (k0 + 4 * g >= nks) break;
                asm volatile("" ::: "memory");            // (the tile is re-used: keep the writes behind the previous group's reads)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<bf16x8*>(sXw + (i * 8 + xr8) * XRS + xc8 * 16) = xv[4 * g + i];
                asm volatile("" ::: "memory");            // same wave, in-order LDS: the reads below see the rows the other lanes wrote
                bf16x8 xf[4], wf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xf[j] = *reinterpret_cast<const bf16x8*>(sXw + l31 * XRS + (2 * j + hh) * 16);
                    wf[j] = *reinterpret_cast<const bf16x8*>(sWh + ((size_t)(k0 + 4 * g + j) * 64 + lane) * 8);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], xf[j], acc, 0, 0, 0);
            }
        }
        uint2 qp, kp, vp;
        qp.x = pack_bf16x2(acc[0], acc[1]); qp.y = pack_bf16x2(acc[2], acc[3]);
        kp.x = pack_bf16x2(acc[4], acc[5]); kp.y = pack_bf16x2(acc[6], acc[7]);
        vp.x = pack_bf16x2(acc[8], acc[9]); vp.y = pack_bf16x2(acc[10], acc[11]);
        // lanes 0..31 gather q[4..7] of their query from the partner half-wave
        const uint32_t px_ = __shfl(qp.x, lane | 32), py_ = __shfl(qp.y, lane | 32);
        if (hh == 0) qB[ti] = make_uint4(qp.x, qp.y, px_, py_);
        const int key = T * 32 + l31;
        *reinterpret_cast<uint2*>(sKh + (size_t)key * 8 + 4 * hh) = kp;
        const int j16 = key & 15;
        const int pos = (key & ~15) + 8 * ((j16 >> 2) & 1) + (j16 & 3) + 4 * (j16 >> 3);
        bf16_t* vcol = sVh + pos;
        vcol[(4 * hh + 0) * vst] = (bf16_t)(vp.x & 0xffffu);
        vcol[(4 * hh + 1) * vst] = (bf16_t)(vp.x >> 16);
        vcol[(4 * hh + 2) * vst] = (bf16_t)(vp.y & 0xffffu);
        vcol[(4 * hh + 3) * vst] = (bf16_t)(vp.y >> 16);
        if (hh == 0) {
            vcol[8 * vst] = (bf16_t)0x3f80;               // 1.0: the PV MFMA's row 8 accumulates the softmax denominator
            vcol[9 * vst] = (bf16_t)0;
        }
    }
    RLDM_ASTAMP();                                        // 3: this wave's tiles projected
    __syncthreads();
    RLDM_ASTAMP();                                        // 4: everyone's

    // ---- key loop -------------------------------------------------------------------------------------------------------------
    const int q0a = T0 * 32, q0b = (T0 + wph) * 32;
    do {                                                  // (one exit: a trunk phase ends in its arrive)
    if (q0a >= L) break;                                  // (no barriers below)
    const bool haveB = PAIR && q0b < L;
    // A operand of S^T: lanes 0..31 the 16-byte K row of their key, lanes 32..63 the constant row {1, 0, ...} (stride 0)
    const bf16_t* kptr = hh == 0 ? sKh + l31 * 8 : sC;
    const int kstep = hh == 0 ? 32 * 8 : 0;
    const bf16_t* vptr = sVh + min(l31, 9) * vst + 8 * hh;
    const bool ragged = (L & 31) != 0;
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;

    // stabiliser: m = exact maximum over the first key tile (raw scores: the -m slot is still zero), rounded to bf16
    uint4 kf = *reinterpret_cast<const uint4*>(kptr);
    kptr += kstep;
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        f32x16 s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qB[ti]), zero, 0, 0, 0);
        if (ragged && 32 > L) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r & 3) + 8 * (r >> 2) + 4 * hh >= L) s0[r] = -1e30f;
        }
        float tmax = fmaxf(fmaxf(s0[0], s0[1]), s0[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) tmax = fmaxf(fmaxf(tmax, s0[r]), s0[r + 1]);
        tmax = fmaxf(tmax, s0[15]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        if (hh == 1) qB[ti].x = pack_bf16x2(-tmax, 0.f);
    }
    RLDM_ASTAMP();                                        // 5: stabilisers

    // rotated loop: on entry of iteration k0, sA / sB hold S(k0) and v0 / v1 the V^T fragments of tile k0; kf holds K(k0 + 32)
    // (the reads past the last tile land in the LDS regions behind sK / sVt and are never used)
    f32x16 oA, oB, sA, sB;
#pragma unroll
    for (int r = 0; r < 16; ++r) { oA[r] = 0.f; oB[r] = 0.f; }
    uint4 v0, v1;
#if RLDM_ATTN_WEAVE
    if constexpr (PAIR) {
        // (round 5) the two chains WOVEN at instruction granularity instead of block by block: in the block form a wave issued
        // [24 VALU][PV][PV][S] per chain and stalled twice per chain at the matrix pipe (PV2 behind PV1 on the same accumulator, S behind
        // PV2: SQ_WAIT_INST_ANY 34 % of the wave cycles, profiles/round5_v1_pmc_valu.txt; 207 cycles per 32 x 32 tile against 144 of VALU).
        // Here no two MFMAs of a wave are closer than four exponentials (32 cycles = one 32x32x16 on the pipe) and every MFMA's operands
        // were produced >= 8 VALU instructions earlier:
        //   expA 0-3 | S_B(k0) | expA 4-7, packs | PV_A1 | expA 8-15, packs | PV_A2 | expB 0-3 | S_A(k0 + 32) | expB 4-7, packs | PV_B1 |
        //   expB 8-15, packs | PV_B2 | (next iteration's expA 0-3 cover PV_B2)
        // On entry of iteration k0: sA = S_A(k0), kf = K(k0) (S_B(k0) is issued inside the iteration), v0 / v1 = V^T(k0).
        // Same MFMAs on the same operands in the same per-accumulator order as the block form: identical results bit for bit.
        sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qB[0]), zero, 0, 0, 0);
        v0 = *reinterpret_cast<const uint4*>(vptr);
        v1 = *reinterpret_cast<const uint4*>(vptr + 16);
        auto ex = [](f32x16& s, const int r0, const int r1) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (r >= r0 && r < r1) s[r] = __builtin_amdgcn_exp2f(s[r]);
        };
        auto pk4 = [](const f32x16& s, const int r0) __attribute__((always_inline)) {
            return make_uint4(pack_bf16x2(s[r0], s[r0 + 1]), pack_bf16x2(s[r0 + 2], s[r0 + 3]), pack_bf16x2(s[r0 + 4], s[r0 + 5]),
                              pack_bf16x2(s[r0 + 6], s[r0 + 7]));
        };
        // (two copies of the loop: with keys past L to mask in the last tile, and without -- the mask's index arithmetic is otherwise hoisted
        //  to the top of EVERY iteration: 18 extra VALU instructions beside 48)
        auto key_loop = [&](auto ragged_c) __attribute__((always_inline)) {
        constexpr bool RG = decltype(ragged_c)::value;
        for (int k0 = 0; k0 < Lp; k0 += 32) {
            const bool last_ragged = RG && k0 + 32 > L;
            if (last_ragged) {                                // last tile: keys >= L get -inf scores
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) sA[r] = -1e30f;
            }
            ex(sA, 0, 4);
            __builtin_amdgcn_sched_barrier(0);
            sB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qB[TPW - 1]), zero, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            kf = *reinterpret_cast<const uint4*>(kptr);      // K(k0 + 32): consumed by S_A(k0 + 32) half an iteration from now
            kptr += kstep;
            ex(sA, 4, 8);
            uint4 pa = pk4(sA, 0);
            __builtin_amdgcn_sched_barrier(0);
            oA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0), __builtin_bit_cast(bf16x8, pa), oA, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            ex(sA, 8, 16);
            pa = pk4(sA, 8);
            __builtin_amdgcn_sched_barrier(0);
            oA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), __builtin_bit_cast(bf16x8, pa), oA, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (last_ragged) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) sB[r] = -1e30f;
            }
            ex(sB, 0, 4);
            __builtin_amdgcn_sched_barrier(0);
            sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qB[0]), zero, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            ex(sB, 4, 8);
            uint4 pb = pk4(sB, 0);
            __builtin_amdgcn_sched_barrier(0);
            oB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0), __builtin_bit_cast(bf16x8, pb), oB, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            ex(sB, 8, 16);
            pb = pk4(sB, 8);
            __builtin_amdgcn_sched_barrier(0);
            oB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), __builtin_bit_cast(bf16x8, pb), oB, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // next iteration's V^T fragments (first used behind eight exponentials of the next iteration)
            v0 = *reinterpret_cast<const uint4*>(vptr + k0 + 32);
            v1 = *reinterpret_cast<const uint4*>(vptr + k0 + 48);
        }
        };
        if (ragged) key_loop(std::true_type{});
        else key_loop(std::false_type{});
    } else
#endif
    {
    sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qB[0]), zero, 0, 0, 0);
    if (PAIR) sB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qB[TPW - 1]), zero, 0, 0, 0);
    kf = *reinterpret_cast<const uint4*>(kptr);
    kptr += kstep;
    v0 = *reinterpret_cast<const uint4*>(vptr);
    v1 = *reinterpret_cast<const uint4*>(vptr + 16);
    for (int k0 = 0; k0 < Lp; k0 += 32) {
        if (ragged && k0 + 32 > L) {                      // last tile: keys >= L get -inf scores
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= L) { sA[r] = -1e30f; if (PAIR) sB[r] = -1e30f; }
        }
        uint32_t pk[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2) pk[r >> 1] = pack_bf16x2(__builtin_amdgcn_exp2f(sA[r]), __builtin_amdgcn_exp2f(sA[r + 1]));
        __builtin_amdgcn_sched_barrier(0);
        oA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0),
                                                     __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3])), oA, 0, 0, 0);
        oA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1),
                                                     __builtin_bit_cast(bf16x8, make_uint4(pk[4], pk[5], pk[6], pk[7])), oA, 0, 0, 0);
        sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qB[0]), zero, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (PAIR) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) pk[r >> 1] = pack_bf16x2(__builtin_amdgcn_exp2f(sB[r]), __builtin_amdgcn_exp2f(sB[r + 1]));
            __builtin_amdgcn_sched_barrier(0);
            oB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0),
                                                         __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3])), oB, 0, 0, 0);
            oB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1),
                                                         __builtin_bit_cast(bf16x8, make_uint4(pk[4], pk[5], pk[6], pk[7])), oB, 0, 0, 0);
            sB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qB[TPW - 1]), zero, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // next iteration's fragments (consumed a whole chain of exponentials from now)
        kf = *reinterpret_cast<const uint4*>(kptr);
        kptr += kstep;
        v0 = *reinterpret_cast<const uint4*>(vptr + k0 + 32);
        v1 = *reinterpret_cast<const uint4*>(vptr + k0 + 48);
    }
    }
    RLDM_ASTAMP();                                        // 6: key loop done
#ifdef RLDM_ABLATE
    if (!TRUNK && p.ts && p.L == p.ts_L && lane == 0) {
        if (blockIdx.x == 0 && wave < 16)
            for (int i = 0; i < 8; ++i) p.ts[wave * 8 + i] = i < tsn ? tsv[i] : 0ull;
        if (blockIdx.x < 2048 && wave == 0) {
            p.ts[256 + 2 * blockIdx.x] = t_real0;
            p.ts[257 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
#undef RLDM_ASTAMP
    // o rows: reg r of half hh <-> row (r&3) + 8*(r>>2) + 4*hh.  d = 4*hh + r for r < 4; denominator = row 8 = reg 4 of hh 0
    bf16_t* out_bh = p.out + ((size_t)b * L) * C + h * 8;
    const float dA = __shfl(oA[4], l31);
    const float dB = PAIR ? __shfl(oB[4], l31) : 1.f;
    // a denominator that is not a finite number below 2^120: some score beat the first tile's maximum by ~100 log2 units ->
    // this wave redoes its tiles with the running-maximum loop (rare; exercised by tests/test_hip_kernels.py)
    const bool bad = !(dA < 1.3e36f) || (haveB && !(dB < 1.3e36f));
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
            if (ti == 1 && !haveB) break;
            // attention_tile's B operand: lane (query l31, half hh) holds q[4hh .. 4hh+3]
            const uint32_t zz = __shfl(qB[ti].z, l31), ww = __shfl(qB[ti].w, l31);   // (all lanes: not under the hh select)
            uint2 qq;
            qq.x = hh ? zz : qB[ti].x;
            qq.y = hh ? ww : qB[ti].y;
            attention_tile(sKh, sVh, vst, L, Lp, C, ti ? q0b : q0a, __builtin_bit_cast(s16x4, qq), out_bh, l31, hh);
        }
        break;
    }
    {
        const float inv = 1.0f / dA;
        uint2 ov;
        ov.x = pack_bf16x2(oA[0] * inv, oA[1] * inv);
        ov.y = pack_bf16x2(oA[2] * inv, oA[3] * inv);
        if (q0a + l31 < L) *reinterpret_cast<uint2*>(out_bh + (size_t)(q0a + l31) * C + 4 * hh) = ov;
    }
    if (haveB) {
        const float inv = 1.0f / dB;
        uint2 ov;
        ov.x = pack_bf16x2(oB[0] * inv, oB[1] * inv);
        ov.y = pack_bf16x2(oB[2] * inv, oB[3] * inv);
        if (q0b + l31 < L) *reinterpret_cast<uint2*>(out_bh + (size_t)(q0b + l31) * C + 4 * hh) = ov;
    }
    } while (false);
    if constexpr (!TRUNK) {
        if (p.proj_w && p.proj_counter) attention_proj_tail<true>(p, b, hg, (C >> 3) / HG, NT);
    }
    if constexpr (TRUNK) {
        trunk_arrive(seam, tid);
#ifdef RLDM_ABLATE
        if (seam.ts && tid == 0) {
            seam.ts[0] = t_entry;
            for (int i = 0; i < 8; ++i) seam.ts[1 + i] = i < tsn ? tsv[i] : 0ull;
            seam.ts[9] = __builtin_amdgcn_s_memtime();
        }
#endif
    }
}

}  // namespace rldm
