// Multi-head self-attention with head_dim = 8 for gfx950 (diffusers Attention/AttnProcessor2_0 as UNet2DModel uses it:
// heads = C/8, softmax(q k^T / sqrt(8)) v; single-head analogue in the reference:
// vae/sgm/modules/diffusionmodules/model.py:391-412).
//
// d = 8 is hostile to the K=16/32 MFMAs, so:
//   S^T = K Q^T  uses v_mfma_f32_32x32x8_bf16 (K = 8 exactly, no padding): one MFMA per 32 keys x 32 queries.  In the
//         result layout every lane owns ONE query (column) and 16 of the 32 keys (rows) -> the online softmax is
//         lane-local except for one exchange with the partner half-wave.
//   O^T += V^T P^T uses v_mfma_f32_32x32x16_bf16 with the contraction index permuted to exactly the key order the
//         lane already holds after S^T (no cross-lane shuffles of P).  Only 8 of the 32 "M" rows carry V; row 8 is
//         all-ones, so the MFMA also produces the softmax denominator for free.  The kernel is VALU(exp)-bound, not
//         MFMA-bound, so the idle MFMA rows cost nothing -- and the VALU work per score is cut to max + exp2 + pack: the
//         running maximum rides in the S^T MFMA's C operand and is only raised (with a rescale) when a score beats it by
//         more than 2^8.
// Data: qkv [B][L][3C] bf16 straight out of the fused GroupNorm+QKV GEMM (q pre-multiplied by log2(e)/sqrt(8) through
// the packed Wq).  A workgroup owns one (image, head): it stages that head's K rows (16 B each) and V TRANSPOSED
// ([d][key], so the PV "A" fragments are 8-byte LDS reads) into LDS once -- 32 bytes per key -- and each of its waves
// then runs 32 queries against all keys out of LDS.  out [B][L][C] bf16.
#include "kernels.h"

namespace rldm {

__global__ void __launch_bounds__(512) attention_d8_kernel(const AttnParams p, const int waves_per_block, const int Lp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = waves_per_block * 64;
    const int heads = p.C >> 3;
    const int qtiles = Lp >> 5;
    const int qblocks = (qtiles + waves_per_block - 1) / waves_per_block;
    int bid = blockIdx.x;
    const int qb = bid % qblocks;
    bid /= qblocks;
    const int h = bid % heads;
    const int b = bid / heads;
    const int l31 = lane & 31, hh = lane >> 5;
    const int ld = 3 * p.C;
    const int vst = Lp + 8;                               // V^T row stride (elements): an odd number of 16-byte slots

    bf16_t* sK = reinterpret_cast<bf16_t*>(smem);         // [Lp][8]
    bf16_t* sVt = sK + (size_t)Lp * 8;                    // [10][vst]: V^T (8 rows), a row of ones, a row of zeros
    const bf16_t* qbase = p.qkv + ((size_t)b * p.L) * ld + h * 8;

    // ---- stage K rows and V^T of this head (keys >= L: zeros).  Within every 16 keys V^T is stored in the order the PV
    // MFMA consumes them (key 4*hh + (e & 3) + 8*(e >> 2) at position 8*hh + e), so its A fragment is one 16-byte read ----
    for (int key = tid; key < Lp; key += NT) {
        uint4 kv = make_uint4(0u, 0u, 0u, 0u), vv = kv;
        if (key < p.L) {
            kv = *reinterpret_cast<const uint4*>(qbase + (size_t)key * ld + p.C);
            vv = *reinterpret_cast<const uint4*>(qbase + (size_t)key * ld + 2 * p.C);
        }
        *reinterpret_cast<uint4*>(sK + (size_t)key * 8) = kv;
        const int j = key & 15;
        const int pos = (key & ~15) + 8 * ((j >> 2) & 1) + (j & 3) + 4 * (j >> 3);
        const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int d = 0; d < 8; ++d) sVt[d * vst + pos] = (bf16_t)((d & 1) ? (w[d >> 1] >> 16) : (w[d >> 1] & 0xffffu));
        sVt[8 * vst + pos] = (bf16_t)0x3f80;              // 1.0: the PV MFMA's row 8 accumulates the softmax denominator
        sVt[9 * vst + pos] = (bf16_t)0;
    }
    __syncthreads();

    const int q0 = (qb * waves_per_block + wave) * 32;
    if (q0 >= p.L) return;                                // no barriers below

    // B operand of S^T: Q^T, lane (query l31, half hh) holds q[query][4*hh .. 4*hh+3]
    const int qrow = min(q0 + l31, p.L - 1);              // rows past L are computed on a clamped row, never stored
    const s16x4 qf = *reinterpret_cast<const s16x4*>(qbase + (size_t)qrow * ld + 4 * hh);

    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;

    // A operand rows of the PV MFMA: row l31 < 8 -> V^T[d = l31], row 8 -> ones, rows 9..31 -> the zero row (one address)
    const bf16_t* vrow_ptr = sVt + min(l31, 9) * vst + 8 * hh;
    const bf16_t* krow_ptr = sK + l31 * 8 + 4 * hh;
    const bool ragged = (p.L & 31) != 0;

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
        if (ragged && 32 > p.L) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r & 3) + 8 * (r >> 2) + 4 * hh >= p.L) s[r] = -1e30f;
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
        if (ragged && k0 + 32 > p.L) {                        // last tile: keys >= L get -inf scores
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (k0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= p.L) s[r] = -1e30f;
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
    if (q0 + l31 < p.L) *reinterpret_cast<uint2*>(p.out + ((size_t)b * p.L + q0 + l31) * p.C + h * 8 + 4 * hh) = ov;
}

int launch_attention(const AttnParams& p, hipStream_t stream) {
    RLDM_REQUIRE(p.L >= 1 && p.L <= 4096, "attention: token count must be in [1, 4096] (K/V of one head are LDS-resident)");
    RLDM_REQUIRE(p.C % 8 == 0, "attention: channels must be a multiple of head_dim 8");
    const int Lp = (p.L + 31) / 32 * 32;
    const int qtiles = Lp / 32;
    const int wpb = qtiles < 8 ? qtiles : 8;
    const int qblocks = (qtiles + wpb - 1) / wpb;
    const int grid = p.B * (p.C / 8) * qblocks;
    const size_t lds = (size_t)Lp * 16 + (size_t)10 * (Lp + 8) * 2;
    static size_t max_set = 0;
    if (lds > max_set) {
        RLDM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_d8_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        max_set = lds;
    }
    hipLaunchKernelGGL(attention_d8_kernel, dim3(grid), dim3(64 * wpb), lds, stream, p, wpb, Lp);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace rldm
