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
//         more than 2^8.  (A two-pass version -- exact maxima first, then exp2 + pack only -- measured 23 % slower in the
//         key loop and -0.6 % end to end: the second S^T MFMA and its LDS reads cost more than the max / ballot they remove.)
// Data: qkv [B][L][3C] bf16 straight out of the fused GroupNorm+QKV GEMM (q pre-multiplied by log2(e)/sqrt(8) through
// the packed Wq).  A workgroup owns one (image, head): it stages that head's K rows (16 B each) and V TRANSPOSED
// ([d][key], so the PV "A" fragments are 8-byte LDS reads) into LDS once -- 32 bytes per key -- and each of its waves
// then runs 32 queries against all keys out of LDS.  out [B][L][C] bf16.
#include "attention_body.h"

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

    attention_tile(sK, sVt, vst, p.L, Lp, p.C, q0, qf, p.out + ((size_t)b * p.L) * p.C + h * 8, l31, hh);
}

// Fused GroupNorm -> q/k/v projection -> attention for one (image, head): the three Linear(C, C) maps split perfectly by
// head (head h needs rows 8h..8h+7 of each), so the workgroup projects its own 24 rows itself and the [B][L][3C] q/k/v
// tensor never exists: D[32 rows][32 pixels] = W_h[32][C] * xn[C][32 pixels] on v_mfma_f32_32x32x16_bf16, W_h streamed in
// A-fragment order from L2 (one 1 KiB fragment per 16 input channels, shared by all tiles), xn = the GroupNorm affine applied
// to 16-byte pieces of x on their way into the B operand.  In the result layout lane (pixel l31, half hh) holds
// q[4hh..4hh+3], k[4hh..4hh+3], v[4hh..4hh+3] of its pixel: q is already the S^T MFMA's B operand, k goes to its LDS row
// with one 8-byte store, v to the transposed image.  Every wave projects the pixel tiles it will later own as queries.
// (diffusers Attention: group_norm -> to_q / to_k / to_v -> softmax(q k^T / sqrt(8)) v; to_out stays a conv launch.)
__global__ void __launch_bounds__(1024) attention_qkv_d8_kernel(const AttnQkvParams p, const int waves, const int Lp) {
    constexpr int TPW = 4;                                // query tiles per wave (L <= 1024)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef RLDM_ABLATE
    unsigned long long tsv[8];
    int tsn = 0;
    const unsigned long long t_real0 = __builtin_amdgcn_s_memrealtime();
#define RLDM_ASTAMP() if (tsn < 8) tsv[tsn++] = __builtin_amdgcn_s_memtime()
#else
#define RLDM_ASTAMP()
#endif
    RLDM_ASTAMP();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = waves * 64;
    const int heads = p.C >> 3;
    const int h = blockIdx.x % heads, b = blockIdx.x / heads;
    const int l31 = lane & 31, hh = lane >> 5;
    const int C = p.C, L = p.L;
    const int vst = Lp + 8;
    const int ntiles = Lp >> 5;

    bf16_t* sK = reinterpret_cast<bf16_t*>(smem);         // [Lp][8]
    bf16_t* sVt = sK + (size_t)Lp * 8;                    // [10][vst]
    float* sGa = reinterpret_cast<float*>(sVt + 10 * vst);         // [C]
    float* sGs = sGa + C;
    double* sD = reinterpret_cast<double*>(sGs + C);      // [2][C] scratch
    bf16_t* sW = reinterpret_cast<bf16_t*>(sD + 2 * C);   // [C/16][64 lanes][8]: this image's W_h * diag(a)
    float* sBp = reinterpret_cast<float*>(sW + (size_t)(C >> 4) * 512);    // [32]: b_h + W_h * s

    // ---- GroupNorm affine of image b (conv_igemm.hip's arithmetic) ----------------------------------------------------
    if (p.st == nullptr) {                                // x arrives normalised (producer-side GroupNorm): W' = W, b' = b
        for (int t = tid; t < C; t += NT) { sGa[t] = 1.f; sGs[t] = 0.f; }
        __syncthreads();
    } else {
        const int cpg = C / p.groups;
        for (int t = tid; t < C; t += NT) {
            const float2* src = p.st + (size_t)b * p.P * C + t;
            double S = 0.0, SS = 0.0;
            int q = 0;
            for (; q + 8 <= p.P; q += 8) {
                float2 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(q + j) * C];
#pragma unroll
                for (int j = 0; j < 8; ++j) { S += (double)v[j].x; SS += (double)v[j].y; }
            }
            for (; q < p.P; ++q) {
                const float2 v = src[(size_t)q * C];
                S += (double)v.x;
                SS += (double)v.y;
            }
            sD[t] = S;
            sD[C + t] = SS;
        }
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
            const float a = p.gamma[t] * __builtin_amdgcn_rsqf((float)var + p.eps);
            sGa[t] = a;
            sGs[t] = p.beta[t] - (float)mean * a;
        }
        __syncthreads();
    }

    RLDM_ASTAMP();                                        // GroupNorm affine in LDS
    // ---- the GroupNorm affine folded into the head's weights: W' = W_h * diag(a) (bf16, A-fragment order, LDS),
    // b' = b_h + W_h * s -- the pixel fragments then go from global memory into the MFMA untouched -----------------------
    const int nks = C >> 4;
    const bf16_t* wf_ptr = p.wfrag + (size_t)h * nks * 512;
    for (int q = tid; q < nks * 64; q += NT) {            // piece q = (k-step, lane): row q & 31, channels 16*ks + 8*(lane >> 5) ..
        const int c0 = (q >> 6) * 16 + ((q >> 5) & 1) * 8;
        const uint4 w = *reinterpret_cast<const uint4*>(wf_ptr + (size_t)q * 8);
        const float4 a0 = *reinterpret_cast<const float4*>(sGa + c0), a1 = *reinterpret_cast<const float4*>(sGa + c0 + 4);
        const float4 s0 = *reinterpret_cast<const float4*>(sGs + c0), s1 = *reinterpret_cast<const float4*>(sGs + c0 + 4);
        uint4 n;
        n.x = pack_bf16x2(bf16lo(w.x) * a0.x, bf16hi(w.x) * a0.y);
        n.y = pack_bf16x2(bf16lo(w.y) * a0.z, bf16hi(w.y) * a0.w);
        n.z = pack_bf16x2(bf16lo(w.z) * a1.x, bf16hi(w.z) * a1.y);
        n.w = pack_bf16x2(bf16lo(w.w) * a1.z, bf16hi(w.w) * a1.w);
        *reinterpret_cast<uint4*>(sW + (size_t)q * 8) = n;
        // partial of W_h * s for row q & 31 over these 8 channels -> scratch [piece]
        reinterpret_cast<float*>(sD)[q] = bf16lo(w.x) * s0.x + bf16hi(w.x) * s0.y + bf16lo(w.y) * s0.z + bf16hi(w.y) * s0.w +
                                          bf16lo(w.z) * s1.x + bf16hi(w.z) * s1.y + bf16lo(w.w) * s1.z + bf16hi(w.w) * s1.w;
    }
    __syncthreads();
    if (tid < 32) {                                       // fixed summation order
        float acc_b = tid < 24 ? p.bias[h * 32 + tid] : 0.f;
        for (int j = 0; j < 2 * nks; ++j) acc_b += reinterpret_cast<float*>(sD)[j * 32 + tid];
        sBp[tid] = acc_b;
    }
    __syncthreads();

    RLDM_ASTAMP();                                        // W', b' in LDS
    // ---- projection of this wave's pixel tiles T = wave, wave + waves, ...: one tile at a time, its rows of x requested
    // whole (a lane's 16-byte pieces of one 128-byte line are issued back to back: every line is fetched once) ------------
    float binit[12];
#pragma unroll
    for (int r = 0; r < 12; ++r) binit[r] = sBp[8 * (r >> 2) + 4 * hh + (r & 3)];
    s16x4 qf[TPW];
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int T = wave + ti * waves;
        if (T >= ntiles) continue;
        const int px = min(T * 32 + l31, L - 1);          // keys / queries past L: a clamped row, masked later
        const bf16_t* xrow = p.x + ((size_t)b * L + px) * C + 8 * hh;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = r < 12 ? binit[r] : 0.f;
        constexpr int KB = 16;                            // k-steps (16 channels each) per batch of loads
        for (int k0 = 0; k0 < nks; k0 += KB) {
            bf16x8 xv[KB];
#pragma unroll
            for (int j = 0; j < KB; ++j)
                if (k0 + j < nks) xv[j] = *reinterpret_cast<const bf16x8*>(xrow + (k0 + j) * 16);
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                if (k0 + j >= nks) break;
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(sW + ((size_t)(k0 + j) * 64 + lane) * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xv[j], acc, 0, 0, 0);
            }
        }
        // q stays in registers (the S^T MFMA's B operand), k / v go to LDS
        uint2 qp, kp, vp;
        qp.x = pack_bf16x2(acc[0], acc[1]); qp.y = pack_bf16x2(acc[2], acc[3]);
        kp.x = pack_bf16x2(acc[4], acc[5]); kp.y = pack_bf16x2(acc[6], acc[7]);
        vp.x = pack_bf16x2(acc[8], acc[9]); vp.y = pack_bf16x2(acc[10], acc[11]);
        qf[ti] = __builtin_bit_cast(s16x4, qp);
        const int key = T * 32 + l31;
        *reinterpret_cast<uint2*>(sK + (size_t)key * 8 + 4 * hh) = kp;
        const int j16 = key & 15;
        const int pos = (key & ~15) + 8 * ((j16 >> 2) & 1) + (j16 & 3) + 4 * (j16 >> 3);
        bf16_t* vcol = sVt + pos;
        vcol[(4 * hh + 0) * vst] = (bf16_t)(vp.x & 0xffffu);
        vcol[(4 * hh + 1) * vst] = (bf16_t)(vp.x >> 16);
        vcol[(4 * hh + 2) * vst] = (bf16_t)(vp.y & 0xffffu);
        vcol[(4 * hh + 3) * vst] = (bf16_t)(vp.y >> 16);
        if (hh == 0) {
            vcol[8 * vst] = (bf16_t)0x3f80;               // 1.0: the PV MFMA's row 8 accumulates the softmax denominator
            vcol[9 * vst] = (bf16_t)0;
        }
    }
    RLDM_ASTAMP();                                        // this wave's tiles projected
    __syncthreads();
    RLDM_ASTAMP();                                        // ... everyone's
    bf16_t* out_bh = p.out + ((size_t)b * L) * C + h * 8;
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int q0 = (wave + ti * waves) * 32;
        if (q0 < L) attention_tile(sK, sVt, vst, L, Lp, C, q0, qf[ti], out_bh, l31, hh);
    }
    RLDM_ASTAMP();                                        // attention of this wave's tiles
#ifdef RLDM_ABLATE
    if (p.ts && p.L == p.ts_L && tid == 0) {
        if (blockIdx.x == 0)
            for (int i = 0; i < 8; ++i) p.ts[i] = i < tsn ? tsv[i] : 0ull;
        if (blockIdx.x < 2048) {
            p.ts[256 + 2 * blockIdx.x] = t_real0;
            p.ts[257 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
#undef RLDM_ASTAMP
}


template <int PAIR>
__global__ void __launch_bounds__(1024) attention_qkv2_d8_kernel(const AttnQkvParams p, const int waves, const int Lp, const int HG) {
    // workgroups land on XCD (blockIdx % 8): give every XCD a contiguous range of (image, head group) ids, so all the heads of
    // an image read its x rows through ONE L2 instead of eight (HBM-side fetch of an L = 1024 launch: 38 MB -> one copy of x)
    const int hgroups = (p.C >> 3) / HG;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const TrunkSeam none = {};
    attention_qkv2_body<PAIR, false>(p, waves, Lp, HG, bid / hgroups, bid % hgroups, none);
}

static int g_attn_old = getenv("RLDM_ATTN_OLD") ? atoi(getenv("RLDM_ATTN_OLD")) : 0;
static int g_attn_hg = getenv("RLDM_ATTN_HG") ? atoi(getenv("RLDM_ATTN_HG")) : 0;      // force heads per workgroup (A/B runs)

int attention_qkv2_geometry(int B, int L, int C, int* HG_out, int* waves_out) {
    const int Lp = (L + 31) / 32 * 32;
    const int ntiles = Lp / 32;
    const int heads = C / 8;
    // a head's query tiles go two to a wave when it has >= 32 of them (L = 1024: 16 waves x 2 tiles), else one to a wave; heads
    // with few tiles share a workgroup (HG heads) as long as the grid still covers the chip once
    const int pair = ntiles >= 32 ? 1 : 0;
    const int wph = pair ? (ntiles + 1) / 2 : ntiles;     // waves per head
    int HG = 1;
    while (HG * 2 * wph <= 16 && heads % (HG * 2) == 0 && (long long)B * (heads / (HG * 2)) >= 256) HG *= 2;
    if (g_attn_hg > 0 && g_attn_hg * wph <= 16 && heads % g_attn_hg == 0) HG = g_attn_hg;
    const int waves = HG * wph;
    if (waves > 16 || L > 1024 || C % 64 != 0) return -1;          // (caller falls back to the first-generation kernel; C % 64: the
                                                                    //  projection stages x in groups of 64 channels)
    if (attention_qkv2_lds_bytes(L, C, HG, waves) > 160 * 1024) return -1;
    *HG_out = HG;
    *waves_out = waves;
    return 0;
}

// The output projection rides in the attention launch when: the geometry above holds; an image's R = heads / HG workgroups each
// take one 64-pixel block of it (L == 64 R); C <= 128 (the tail keeps a wave's 8 weight fragments in registers; 2 x C/32 output
// sub-tiles <= waves); the XCD-contiguous id ranges (xcd_remap) never cut an image: grid % 8 == 0 and (grid / 8) % R == 0; and every
// workgroup is resident at once (1024 threads, > 80 KB of LDS: one per CU).
bool attention_proj_fusable(int B, int L, int C, int cus) {     // (cus < 0: the shape only -- the tail as a launch of its own)
    int HG = 0, waves = 0;
    if (g_attn_old || attention_qkv2_geometry(B, L, C, &HG, &waves)) return false;
    const int R = (C / 8) / HG, grid = B * R;
    if (cus < 0) cus = grid;
    const int ng = waves * 128 / C;                       // pixel groups of the tail's statistics pass (threads / channel pairs)
    return L == 64 * R && C <= 128 && 2 * (C / 32) <= waves && (waves * 128) % C == 0 && ng <= 64 && 64 % ng == 0 &&
           grid % 8 == 0 && (grid / 8) % R == 0 && grid <= cus;
}

// the output projection as a launch of its own (attention_proj_tail<false>): one workgroup per 64-pixel block of an image
__global__ void __launch_bounds__(1024) attention_proj_kernel(const AttnQkvParams p, const int R) {
    attention_proj_tail<false>(p, blockIdx.x / R, blockIdx.x % R, R, blockDim.x);
}

int launch_attention_proj(const AttnQkvParams& p, hipStream_t stream) {
    int HG = 0, waves = 0;
    RLDM_REQUIRE(p.proj_w && attention_qkv2_geometry(p.B, p.L, p.C, &HG, &waves) == 0, "attention_proj: unsupported shape");
    const int R = (p.C / 8) / HG;
    const size_t lds = (size_t)64 * (p.C * 2 + 16) + (size_t)64 * (p.C * 4 + 16) + (size_t)(waves * 128 / p.C) * 2 * p.C * 4;
    static DynLdsLimit lds_limit;
    RLDM_HIP_CHECK(lds_limit.ensure(reinterpret_cast<const void*>(attention_proj_kernel), lds));
    hipLaunchKernelGGL(attention_proj_kernel, dim3(p.B * R), dim3(64 * waves), lds, stream, p, R);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_attention_qkv2(const AttnQkvParams& p, hipStream_t stream) {
    const int Lp = (p.L + 31) / 32 * 32;
    const int pair = Lp / 32 >= 32 ? 1 : 0;
    int HG = 0, waves = 0;
    if (attention_qkv2_geometry(p.B, p.L, p.C, &HG, &waves)) return -1;
    const size_t lds = attention_qkv2_lds_bytes(p.L, p.C, HG, waves);
    auto kern = pair ? attention_qkv2_d8_kernel<1> : attention_qkv2_d8_kernel<0>;
    static DynLdsLimit lds_limit[2];             // per instantiation, per device
    RLDM_HIP_CHECK(lds_limit[pair].ensure(reinterpret_cast<const void*>(kern), lds));
    hipLaunchKernelGGL(kern, dim3(p.B * ((p.C / 8) / HG)), dim3(64 * waves), lds, stream, p, waves, Lp, HG);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_attention_qkv(const AttnQkvParams& p, hipStream_t stream) {
    RLDM_REQUIRE(p.L >= 1 && p.L <= 1024, "attention_qkv: token count must be in [1, 1024]");
    RLDM_REQUIRE(p.C % 16 == 0 && p.C <= 512 && p.C % p.groups == 0, "attention_qkv: channels must be a multiple of 16, <= 512");
    if (!g_attn_old) {
        const int rc = launch_attention_qkv2(p, stream);
        if (rc >= 0) return rc;
    }
    RLDM_REQUIRE(p.proj_w == nullptr, "attention_qkv: the fused output projection needs the second-generation launch");
    const int Lp = (p.L + 31) / 32 * 32;
    const int ntiles = Lp / 32;
    // up to 16 waves per (image, head): with 1024 tokens that is 4 waves per SIMD on one workgroup per CU
    const int waves = ntiles <= 4 ? 4 : (ntiles <= 8 ? ntiles : 16);       // (>= 4: the affine / weight fold uses every thread)
    const size_t lds = (size_t)Lp * 16 + (size_t)10 * (Lp + 8) * 2 + (size_t)p.C * 8 + (size_t)2 * p.C * 8 + (size_t)p.C * 64 + 128;
    static DynLdsLimit lds_limit;                // per device, thread safe
    RLDM_HIP_CHECK(lds_limit.ensure(reinterpret_cast<const void*>(attention_qkv_d8_kernel), lds));
    hipLaunchKernelGGL(attention_qkv_d8_kernel, dim3(p.B * (p.C / 8)), dim3(64 * waves), lds, stream, p, waves, Lp);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
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
    static DynLdsLimit lds_limit;                // per device, thread safe
    RLDM_HIP_CHECK(lds_limit.ensure(reinterpret_cast<const void*>(attention_d8_kernel), lds));
    hipLaunchKernelGGL(attention_d8_kernel, dim3(grid), dim3(64 * wpb), lds, stream, p, wpb, Lp);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace rldm
