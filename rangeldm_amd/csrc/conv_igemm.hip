// Circular implicit-GEMM convolution for gfx950 (MI355X), bf16 in / fp32 accumulate on v_mfma_f32_32x32x16_bf16.
//
// Replaces the reference's `F.pad(circular W) + F.pad(zero H) + F.conv2d` (ldm/utils.py:46-49, dup
// vae/sgm/modules/diffusionmodules/model.py:99-102), the stride-2 down-samplers (ldm/utils.py:107-116, model.py:164-172),
// nearest-x2 + conv up-samplers (model.py:120-125), the 1x1 shortcuts and the attention Linear layers, with the
// GroupNorm+SiLU that precedes every conv folded into the tile loader and bias / time-embedding / residual folded
// into the epilogue.
//
// Mapping (one 256-thread workgroup = 4 waves):
//   GEMM view  D[n][m] = sum_k W[n][k] * X[k][m],  n = output channel, m = output pixel, k = (tap, input channel).
//   The MFMA "A" operand is the weight tile (32 channels x 16 k), the "B" operand the pixel tile (16 k x 32 pixels),
//   so each lane ends up holding 4 consecutive output channels of ONE pixel per register quad -> 8-byte channels-last
//   stores, per-pixel residual / per-channel bias adds without any cross-lane traffic.
//   Input: for every CK-channel chunk the block stages the (TW*s+2) x (TH*s+2) input HALO of its TW x TH output tile
//   in LDS once -- wrap-around along W (azimuth), zeros along H (beams), nearest-x2 and channel-concat resolved in the
//   address computation, GroupNorm affine + SiLU applied on the way in -- and all 9 taps read it back with a
//   per-tap constant offset: 9x fewer global/L2 reads and 9x fewer normalisation FLOPs than im2col-per-tap.
//   Weights: packed on the host per (n-tile, chunk, tap) in exactly the padded LDS image (row = 2*CK + 16 bytes: the
//   odd 16-byte-slot stride makes every ds_read_b128 lane group conflict-free), register-prefetched one tap ahead.
#include "kernels.h"

namespace rldm {

template <int BM, int BN, int WM, int WN, int CK, int TAPS>
__global__ void __launch_bounds__(256, 2) conv_igemm_kernel(const ConvParams p) {
    constexpr int MI = BM / (32 * WM);        // 32-pixel MFMA tiles per wave
    constexpr int NI = BN / (32 * WN);        // 32-channel MFMA tiles per wave
    constexpr int RS = CK * 2 + 16;           // LDS row stride in bytes
    constexpr int KS = CK / 16;               // MFMA k-steps per (tap, chunk)
    constexpr int KW = (TAPS == 9) ? 3 : 1;
    constexpr int C8 = CK / 8;                // 16-byte pieces per LDS row
    constexpr int WCH = BN * RS / 16;         // 16-byte pieces per weight tile
    constexpr int WFULL = WCH / 256;          // full 256-thread passes over a weight tile
    constexpr int WTAIL = WCH % 256;          // pieces left for a partial pass
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(MI >= 1 && NI >= 1, "tile too small for the wave grid");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int kh = lane >> 5, l31 = lane & 31;

    // ---- which tile -----------------------------------------------------------------------------------------
    const int tiles_h = p.Hout / p.TH;
    const int tiles_img = (p.Wout / p.TW) * tiles_h;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = lid % p.ntile_n;
    int mt = lid / p.ntile_n;
    const int b = mt / tiles_img;
    mt -= b * tiles_img;
    const int tw = mt / tiles_h, th = mt - tw * tiles_h;
    const int w0 = tw * p.TW, h0 = th * p.TH;
    const int npx = p.TW * p.TH;

    const int Cin = p.C0 + p.C1;
    const int NCC = Cin / CK;
    const int THv = (p.TH - 1) * p.stride + KW;
    const int TWv = (p.TW - 1) * p.stride + KW;
    const int nslots = TWv * THv;
    const int Wv = p.Win * p.up, Hv = p.Hin * p.up;
    const int upshift = p.up - 1;             // up in {1,2}

    unsigned char* sW = smem;
    unsigned char* sA = smem + 2 * BN * RS;
    float* sGa = reinterpret_cast<float*>(sA + ((nslots * RS + 15) & ~15));
    float* sGs = sGa + Cin;

    // ---- GroupNorm finalize: (sum, sumsq) partials -> per-channel affine a*x + s ---------------------------------
    const bool gn = p.gn_part != nullptr;
    if (gn) {
        const int cpg = Cin / p.gn_groups;
        const double inv_n = 1.0 / ((double)p.Win * (double)p.Hin * (double)cpg);
        for (int c = tid; c < Cin; c += 256) {
            const int g = c / cpg;
            double S = 0.0, SS = 0.0;
            for (int q = 0; q < p.gn_P; ++q) {
                const float2 v = p.gn_part[((size_t)b * p.gn_P + q) * p.gn_groups + g];
                S += (double)v.x;
                SS += (double)v.y;
            }
            const double mean = S * inv_n;
            double var = SS * inv_n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const float rstd = (float)(1.0 / sqrt(var + (double)p.gn_eps));
            const float a = p.gn_gamma[c] * rstd;
            sGa[c] = a;
            sGs[c] = p.gn_beta[c] - (float)mean * a;
        }
    }

    // ---- per-lane LDS byte offsets of the MFMA operands ----------------------------------------------------------
    int xoff[MI], woff[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        int pidx = wm * (MI * 32) + mi * 32 + l31;
        if (pidx >= npx) pidx = 0;            // masked at the store
        const int pw = pidx / p.TH, ph = pidx - pw * p.TH;
        xoff[mi] = ((pw * p.stride) * THv + ph * p.stride) * RS + kh * 16;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) woff[ni] = (wn * (NI * 32) + ni * 32 + l31) * RS + kh * 16;

    f32x16 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

    const uint4* wsrc_base = reinterpret_cast<const uint4*>(p.wpk) + (size_t)nt * NCC * TAPS * WCH;

    for (int cc = 0; cc < NCC; ++cc) {
        __syncthreads();   // previous chunk's MFMAs are done with sA / sW (also orders sGa writes on cc == 0)
        // ---- stage the input halo for channels [cc*CK, cc*CK + CK) ------------------------------------------------
        {
            const int total = nslots * C8;
            for (int q = tid; q < total; q += 256) {
                const int slot = q / C8, c8 = q - slot * C8;
                const int vwl = slot / THv, vhl = slot - vwl * THv;
                const int vh = h0 * p.stride - p.pad_lo + vhl;
                int vw = w0 * p.stride - p.pad_lo + vwl;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (vh >= 0 && vh < Hv) {                       // beams: zero padding
                    vw = vw < 0 ? vw + Wv : (vw >= Wv ? vw - Wv : vw);   // azimuth: wrap-around
                    const int sw = vw >> upshift, sh = vh >> upshift;
                    const int c = cc * CK + c8 * 8;
                    const size_t pix = ((size_t)b * p.Win + sw) * p.Hin + sh;
                    const bf16_t* src = (c < p.C0) ? p.x0 + pix * p.C0 + c : p.x1 + pix * p.C1 + (c - p.C0);
                    v = *reinterpret_cast<const uint4*>(src);
                    if (gn) {
                        const float4 a0 = *reinterpret_cast<const float4*>(sGa + c);
                        const float4 a1 = *reinterpret_cast<const float4*>(sGa + c + 4);
                        const float4 s0 = *reinterpret_cast<const float4*>(sGs + c);
                        const float4 s1 = *reinterpret_cast<const float4*>(sGs + c + 4);
                        float f0 = bf16lo(v.x) * a0.x + s0.x, f1 = bf16hi(v.x) * a0.y + s0.y;
                        float f2 = bf16lo(v.y) * a0.z + s0.z, f3 = bf16hi(v.y) * a0.w + s0.w;
                        float f4 = bf16lo(v.z) * a1.x + s1.x, f5 = bf16hi(v.z) * a1.y + s1.y;
                        float f6 = bf16lo(v.w) * a1.z + s1.z, f7 = bf16hi(v.w) * a1.w + s1.w;
                        if (p.silu) {
                            f0 = silu_f(f0); f1 = silu_f(f1); f2 = silu_f(f2); f3 = silu_f(f3);
                            f4 = silu_f(f4); f5 = silu_f(f5); f6 = silu_f(f6); f7 = silu_f(f7);
                        }
                        v.x = pack_bf16x2(f0, f1); v.y = pack_bf16x2(f2, f3);
                        v.z = pack_bf16x2(f4, f5); v.w = pack_bf16x2(f6, f7);
                    }
                }
                *reinterpret_cast<uint4*>(sA + slot * RS + c8 * 16) = v;
            }
        }
        // ---- first weight tile of the chunk ----------------------------------------------------------------------
        {
            const uint4* src = wsrc_base + (size_t)cc * TAPS * WCH;
            uint4* dst = reinterpret_cast<uint4*>(sW);
#pragma unroll
            for (int i = 0; i < WFULL; ++i) dst[tid + i * 256] = src[tid + i * 256];
            if constexpr (WTAIL > 0) {
                if (tid < WTAIL) dst[WFULL * 256 + tid] = src[WFULL * 256 + tid];
            }
        }
        __syncthreads();

        auto compute = [&](int tap) {
            const int ti = tap / 3, tj = tap - ti * 3;
            const int tapoff = (ti * THv + tj) * RS;
            const unsigned char* wb = sW + (tap & 1) * (BN * RS);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 wf[NI], xf[MI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) wf[ni] = *reinterpret_cast<const bf16x8*>(wb + woff[ni] + ks * 32);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    xf[mi] = *reinterpret_cast<const bf16x8*>(sA + xoff[mi] + tapoff + ks * 32);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
            }
        };
#pragma unroll 1
        for (int tap = 0; tap < TAPS - 1; ++tap) {
            // register-prefetch the next tap's weights (unconditional clamped loads keep wreg in VGPRs)
            // (named scalars, not an array: hipcc leaves dead scratch stores behind for a uint4[] here)
            static_assert(WFULL <= 5, "weight tile too large for the register prefetch");
            uint4 w0 = make_uint4(0u, 0u, 0u, 0u), w1 = w0, w2 = w0, w3 = w0, w4 = w0, wtail = w0;
            const uint4* src = wsrc_base + ((size_t)cc * TAPS + tap + 1) * WCH + tid;
            if constexpr (WFULL > 0) w0 = src[0];
            if constexpr (WFULL > 1) w1 = src[256];
            if constexpr (WFULL > 2) w2 = src[512];
            if constexpr (WFULL > 3) w3 = src[768];
            if constexpr (WFULL > 4) w4 = src[1024];
            if constexpr (WTAIL > 0) wtail = src[tid < WTAIL ? WFULL * 256 : 0];
            compute(tap);
            uint4* dst = reinterpret_cast<uint4*>(sW + ((tap + 1) & 1) * (BN * RS)) + tid;
            if constexpr (WFULL > 0) dst[0] = w0;
            if constexpr (WFULL > 1) dst[256] = w1;
            if constexpr (WFULL > 2) dst[512] = w2;
            if constexpr (WFULL > 3) dst[768] = w3;
            if constexpr (WFULL > 4) dst[1024] = w4;
            if constexpr (WTAIL > 0) {
                if (tid < WTAIL) dst[WFULL * 256] = wtail;
            }
            __syncthreads();
        }
        compute(TAPS - 1);
    }

    // ---- epilogue: bias (+ time embedding) (+ residual) -> bf16 channels-last (or fp32 NCHW / transposed V) ----------
    const float* temb_row = nullptr;
    if (p.temb) {
        const int step = p.step_ptr ? *p.step_ptr : 0;
        temb_row = p.temb + (size_t)(step * p.temb_rows_per_step + (p.temb_per_sample ? b : 0)) * p.temb_ld;
    }
    const int L = p.Wout * p.Hout;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int pidx = wm * (MI * 32) + mi * 32 + l31;
        const bool pvalid = pidx < npx;
        const int pw = pidx / p.TH, ph = pidx - pw * p.TH;
        const int ow = w0 + pw, oh = h0 + ph;
        const size_t pix = ((size_t)b * p.Wout + ow) * p.Hout + oh;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int ch = nt * BN + wn * (NI * 32) + ni * 32 + 8 * r4 + 4 * kh;
                const float4 bv = *reinterpret_cast<const float4*>(p.bias + ch);
                float v0 = acc[ni][mi][r4 * 4 + 0] + bv.x, v1 = acc[ni][mi][r4 * 4 + 1] + bv.y;
                float v2 = acc[ni][mi][r4 * 4 + 2] + bv.z, v3 = acc[ni][mi][r4 * 4 + 3] + bv.w;
                if (!pvalid || ch >= p.N) continue;
                if (p.y_nchw) {
                    const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ch + e < p.N) p.y_nchw[(((size_t)b * p.N + ch + e) * p.Wout + ow) * p.Hout + oh] = vv[e];
                    continue;
                }
                if (temb_row) {
                    const float4 tv = *reinterpret_cast<const float4*>(temb_row + ch);
                    v0 += tv.x; v1 += tv.y; v2 += tv.z; v3 += tv.w;
                }
                if (ch < p.n_store) {
                    if (ch + 3 < p.n_store) {
                        if (p.res) {
                            const uint2 rv = *reinterpret_cast<const uint2*>(p.res + pix * p.N + ch);
                            v0 += bf16lo(rv.x); v1 += bf16hi(rv.x); v2 += bf16lo(rv.y); v3 += bf16hi(rv.y);
                        }
                        uint2 o;
                        o.x = pack_bf16x2(v0, v1);
                        o.y = pack_bf16x2(v2, v3);
                        *reinterpret_cast<uint2*>(p.y + pix * p.y_ld + ch) = o;
                    } else {                                   // channel count not a multiple of 4: scalar tail
                        const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (ch + e < p.n_store) {
                                float t = vv[e];
                                if (p.res) t += bf16_to_f32(p.res[pix * p.N + ch + e]);
                                p.y[pix * p.y_ld + ch + e] = f32_to_bf16(t);
                            }
                    }
                } else {
                    // attention V, stored transposed per head: vt[b][head][d][token]
                    const int cv = ch - p.n_store;
                    const int heads = (p.N - p.n_store) >> 3;
                    bf16_t* dst = p.vt + (((size_t)b * heads + (cv >> 3)) * 8 + (cv & 7)) * L + (ow * p.Hout + oh);
                    dst[0] = f32_to_bf16(v0);
                    dst[(size_t)L] = f32_to_bf16(v1);
                    dst[(size_t)2 * L] = f32_to_bf16(v2);
                    dst[(size_t)3 * L] = f32_to_bf16(v3);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
size_t conv_lds_bytes(const ConvTile& t, const ConvParams& p) {
    const int RS = conv_row_bytes(t.CK);
    const int KW = t.taps == 9 ? 3 : 1;
    const int THv = (p.TH - 1) * p.stride + KW, TWv = (p.TW - 1) * p.stride + KW;
    size_t a = ((size_t)TWv * THv * RS + 15) & ~(size_t)15;
    size_t g = p.gn_part ? (size_t)(p.C0 + p.C1) * 8 : 0;
    return (size_t)2 * t.BN * RS + a + g;
}

template <int BM, int BN, int WM, int WN, int CK, int TAPS>
static int launch_inst(const ConvParams& p, int grid, size_t lds, hipStream_t stream) {
    auto kern = conv_igemm_kernel<BM, BN, WM, WN, CK, TAPS>;
    static size_t max_set = 0;
    if (lds > max_set) {
        RLDM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        max_set = lds;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

#define RLDM_CONV_INSTANCES(X)                                                                                  \
    X(128, 128, 2, 2, 64, 9) X(128, 128, 2, 2, 64, 1) X(64, 64, 2, 2, 64, 9) X(64, 64, 2, 2, 64, 1)             \
    X(128, 64, 2, 2, 64, 9) X(128, 64, 2, 2, 64, 1) X(128, 32, 4, 1, 64, 9)                                     \
    X(128, 128, 2, 2, 16, 9) X(128, 64, 2, 2, 16, 9) X(64, 64, 2, 2, 16, 9) X(64, 64, 2, 2, 16, 1)              \
    X(128, 32, 4, 1, 16, 9)

bool conv_tile_supported(const ConvTile& t) {
#define X(bm_, bn_, wm_, wn_, ck_, taps_) \
    if (t.BM == bm_ && t.BN == bn_ && t.CK == ck_ && t.taps == taps_) return true;
    RLDM_CONV_INSTANCES(X)
#undef X
    return false;
}

int launch_conv(const ConvTile& t, const ConvParams& p, hipStream_t stream) {
    RLDM_REQUIRE(p.Wout % p.TW == 0 && p.Hout % p.TH == 0, "conv: output size is not a multiple of the pixel tile");
    RLDM_REQUIRE(p.TW * p.TH <= t.BM, "conv: pixel tile larger than BM");
    RLDM_REQUIRE((p.C0 + p.C1) % t.CK == 0, "conv: input channels not a multiple of CK");
    RLDM_REQUIRE(p.C1 == 0 || p.C0 % 8 == 0, "conv: concat boundary must be a multiple of 8 channels");
    RLDM_REQUIRE(p.up == 1 || p.up == 2, "conv: upsample factor must be 1 or 2");
    RLDM_REQUIRE(p.Win * p.up >= 2 || t.taps == 1, "conv: azimuth extent too small for wrap-around");
    const int grid = p.B * (p.Wout / p.TW) * (p.Hout / p.TH) * p.ntile_n;
    const size_t lds = conv_lds_bytes(t, p);
    RLDM_REQUIRE(lds <= 160 * 1024, "conv: LDS footprint exceeds 160 KiB");
#define X(bm_, bn_, wm_, wn_, ck_, taps_) \
    if (t.BM == bm_ && t.BN == bn_ && t.CK == ck_ && t.taps == taps_) \
        return launch_inst<bm_, bn_, wm_, wn_, ck_, taps_>(p, grid, lds, stream);
    RLDM_CONV_INSTANCES(X)
#undef X
    set_error("conv: no kernel instance for tile BM=" + std::to_string(t.BM) + " BN=" + std::to_string(t.BN) +
              " CK=" + std::to_string(t.CK) + " taps=" + std::to_string(t.taps));
    return 1;
}

}  // namespace rldm
