// conv_stream's 256-pixel x 128-channel tile with SPECIALISED waves (round 4): the four waves of the workgroup's first half only run
// the matrix pipes (one per SIMD: 256 pixels x 32 channels each, 8 MFMAs per weight fragment), the four waves of the second half only
// stage the halo chunks (global -> GroupNorm + SiLU -> LDS).  Why: stamps of the symmetric kernel (every wave does both, conv_stream_body.h)
// put a chunk at 13.6 k cycles for 9.2 k cycles of matrix-pipe time -- a wave that normalises its share of the next chunk is not issuing
// MFMAs meanwhile, and its ~360 VALU instructions per chunk come out of its SIMD partner's MFMA stream when both are mixed
// (MI355X_MICROARCH.md, "Two waves per SIMD": a matrix-only wave and a VALU-only wave on one SIMD run concurrently; moving work between
// two mixed waves is zero-sum).  Here the MFMA wave of a SIMD never leaves its K loop: per k-step 8 MFMAs (256 cycles), 8 ds_read_b128,
// one 1-KiB weight fragment from L2 -- ~2.4 other instructions per MFMA -- while its partner (waves w and w + 4 share a SIMD) does all
// the VALU work of the tile.  Same ConvParams, same weight image (get_streampacked, one k-group), same results bit for bit as
// conv_stream_body<2, 4>: the summation order per output element is unchanged (bias + temb, chunks in order, taps in order, k-steps in
// order, then the residual chunks).
// Reference arithmetic: ldm/utils.py:40-58,107-116; vae/sgm/modules/diffusionmodules/model.py:93-125,342-362.
#pragma once
#include "conv_stream_body.h"

namespace rldm {

template <bool TRUNK>
__device__ __forceinline__ void conv_stream_spec_body(const ConvParams& p, const int nt, const int mt, const int b, const TrunkSeam& seam) {
    constexpr int NT = 512, SNT = 256, CK = 64, MI = 8, WN = 4;
    constexpr int BM = 256, BN = 32 * WN;
    constexpr int RS = CK * 2 + 16;            // halo row stride (bytes): 9 16-byte slots
    constexpr int C8 = CK / 8;
    constexpr int HALO_PX = (BM / 8 + 2) * 10;                 // 34 x 10
    constexpr int ACH = (HALO_PX * C8 + SNT - 1) / SNT;        // 11 pieces per staging thread and chunk
    constexpr int SPT = 4, ROW = 3 * SPT, CST = 9 * SPT, G = ROW;
    constexpr int PFX = 1;                     // (8 MFMAs per k-step: the read behind MFMA mi of step c has the other 7 MFMAs = 224+ cycles of lead for step c + 1)
    constexpr int ERS = BN * 2 + 16, NC8 = BN / 8;
    static_assert(G <= 16, "wave grid");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tid_ = threadIdx.x;
    if constexpr (TRUNK) asm volatile("" : "+v"(tid_));          // (opaque per phase: conv_small_body.h)
    const int tid = tid_, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;           // (waves w and w + 4 sit on one SIMD: one matrix wave and one staging wave each)
    const int wn = wave & 3;
    const int stid = tid - 256;                // staging thread (producers)
    const int kh = lane >> 5, l31 = lane & 31;
#ifdef RLDM_ABLATE
    unsigned long long tsv[12];
    int tsn = 0;
#define RLDM_STAMP() if (tsn < 12) tsv[tsn++] = __builtin_amdgcn_s_memtime()
    const unsigned long long t_real0 = __builtin_amdgcn_s_memrealtime();
#else
#define RLDM_STAMP()
#endif
    RLDM_STAMP();

    const int tiles_h = p.tiles_h, tiles_img = p.tiles_img;       // tiles_h is a power of two
    const int tw = mt >> (31 - __builtin_clz(tiles_h)), th = mt & (tiles_h - 1);
    const int w0 = tw * p.TW, h0 = th * p.TH;

    const float* const temb_tab = TRUNK ? seam.temb : p.temb;
    const int* const step_ptr = TRUNK ? seam.step_ptr : p.step_ptr;
    const int temb_rps = TRUNK ? seam.temb_rows_per_step : p.temb_rows_per_step;
    const int temb_ps = TRUNK ? seam.temb_per_sample : p.temb_per_sample;
    const int temb_ld = TRUNK ? seam.temb_ld : p.temb_ld;
    const int temb_step = (temb_tab && step_ptr && tid < BN) ? load_step_vector(step_ptr) : 0;

    const int Cin = p.C0 + p.C1;
    const int NCC = Cin / CK;                  // main-phase chunks: 9 taps x 4 k-steps
    const int NCB = (p.R0 + p.R1) / CK;        // residual-phase chunks: centre tap, 4 k-steps, raw input
    const int NCT = NCC + NCB;
    const int THv = p.TH + 2, TWv = p.TW + 2;
    const int colb = p.colb;
    const int abytes = TWv * colb;
    const int Wv = p.Win * p.up, Hv = p.Hin * p.up;
    const int upshift = p.up - 1;

    unsigned char* sA = smem;                                  // 2 * abytes
    float* sGa = reinterpret_cast<float*>(sA + 2 * abytes);    // Cin
    float* sGs = sGa + Cin;
    float* sBias = sGs + Cin;                                  // BN

    const bool gn = p.st0 != nullptr;
    const bf16_t* const gx0 = p.x0;
    const bf16_t* const gx1 = p.x1;
    const bf16_t* const gr0 = p.r0;
    const bf16_t* const gr1 = p.r1;
    const int nC0 = p.C0, nC1 = p.C1, nR0 = p.R0, nR1 = p.R1;
    RLDM_STAMP();
    if constexpr (TRUNK) trunk_wait(seam, tid);

    // ---- GroupNorm statistics partials of channel `tid` (all 512 threads; Cin <= 512), requested before anything else ------------------
    double gS = 0.0, gSS = 0.0;
    float g_gamma = 0.f, g_beta = 0.f;
    if (gn && tid < Cin) {
        const float2* const gs0p = p.st0;
        const float2* const gs1p = p.st1;
        const int nP0 = p.P0, nP1 = p.P1;
        const bool first = tid < nC0;
        const int c = first ? tid : tid - nC0;
        const int C = first ? nC0 : nC1;
        const int P = first ? nP0 : nP1;
        const float2* src = (first ? gs0p : gs1p) + (size_t)b * P * C + c;
        g_gamma = p.gn_gamma[tid];
        g_beta = p.gn_beta[tid];
        int q = 0;
        for (; q + 16 <= P; q += 16) {
            float2 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = ld_act8<TRUNK>(src + (size_t)(q + j) * C);
#pragma unroll
            for (int j = 0; j < 16; ++j) { gS += (double)v[j].x; gSS += (double)v[j].y; }
        }
        for (; q + 4 <= P; q += 4) {
            float2 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ld_act8<TRUNK>(src + (size_t)(q + j) * C);
#pragma unroll
            for (int j = 0; j < 4; ++j) { gS += (double)v[j].x; gSS += (double)v[j].y; }
        }
        for (; q < P; ++q) {
            const float2 v = ld_act8<TRUNK>(src + (size_t)q * C);
            gS += (double)v.x;
            gSS += (double)v.y;
        }
    }
    float bias_v = 0.f;
    if (tid < BN) bias_v = p.bias[nt * BN + tid];

    // ---- staging state (producers): source pixel of each of the thread's ACH 16-byte halo pieces ------------------------------------------
    const int atotal = TWv * THv * C8;
    int apix[ACH];
    uint4 areg[ACH];
    const int my_c8 = (stid & (C8 - 1)) * 8;
    auto load_a = [&](int cs) __attribute__((always_inline)) {
        const int c = cs * CK + my_c8;
        const bool first = c < nC0;
        const bf16_t* base = first ? gx0 + c : gx1 + (c - nC0);
        const int ld = first ? nC0 : nC1;
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int pix = apix[i] < 0 ? 0 : apix[i];
            areg[i] = ld_act16<TRUNK>(base + (size_t)pix * ld);
        }
    };
    auto store_a = [&](int cs) __attribute__((always_inline)) {                 // GroupNorm + SiLU -> LDS
        unsigned char* dstbuf = sA + (cs & 1) * abytes;
        float4 ga0, ga1, gs0, gs1;
        if (gn) {
            const int c = cs * CK + my_c8;
            ga0 = *reinterpret_cast<const float4*>(sGa + c);
            ga1 = *reinterpret_cast<const float4*>(sGa + c + 4);
            gs0 = *reinterpret_cast<const float4*>(sGs + c);
            gs1 = *reinterpret_cast<const float4*>(sGs + c + 4);
        }
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            uint4 v = areg[i];
            if (apix[i] < 0) {
                v = make_uint4(0u, 0u, 0u, 0u);
            } else if (gn) {
                float f0 = bf16lo(v.x) * ga0.x + gs0.x, f1 = bf16hi(v.x) * ga0.y + gs0.y;
                float f2 = bf16lo(v.y) * ga0.z + gs0.z, f3 = bf16hi(v.y) * ga0.w + gs0.w;
                float f4 = bf16lo(v.z) * ga1.x + gs1.x, f5 = bf16hi(v.z) * ga1.y + gs1.y;
                float f6 = bf16lo(v.w) * ga1.z + gs1.z, f7 = bf16hi(v.w) * ga1.w + gs1.w;
                if (p.silu) {
                    silu_x8(f0, f1, f2, f3, f4, f5, f6, f7);
                }
                v.x = pack_bf16x2(f0, f1); v.y = pack_bf16x2(f2, f3);
                v.z = pack_bf16x2(f4, f5); v.w = pack_bf16x2(f6, f7);
            }
            const int q = stid + i * SNT;
            const int slot = q / C8, c8 = q - slot * C8;
            const int vwl = (slot * p.magic_thv) >> 20, vhl = slot - vwl * THv;
            if (q < atotal) *reinterpret_cast<uint4*>(dstbuf + vwl * colb + vhl * RS + c8 * 16) = v;
        }
    };
    // residual chunks (raw block input under the centre tap) at the tile's own pixels: 8 pieces per staging thread
    constexpr int RCH = BM * C8 / SNT;
    constexpr int RSTEP = SNT / C8;                                       // tile pixels between a thread's pieces (4 columns of 8)
    auto load_r = [&](int cs, uint4* rr) __attribute__((always_inline)) {
        const int cb = (cs - NCC) * CK;
        const bool first = cb < nR0;
        const bf16_t* base = (first ? gr0 + cb : gr1 + (cb - nR0)) + my_c8;
        const int ld = first ? nR0 : nR1;
        const int pidx = stid / C8, pw = pidx >> p.th_shift, ph = pidx - (pw << p.th_shift);
        const bf16_t* src = base + (size_t)((b * p.Wout + w0 + pw) * p.Hout + h0 + ph) * ld;
        const size_t step = (size_t)((RSTEP >> p.th_shift) * p.Hout) * ld;
#pragma unroll
        for (int i = 0; i < RCH; ++i) rr[i] = ld_act16<TRUNK>(src + i * step);
    };
    auto store_r = [&](int cs, const uint4* rr) __attribute__((always_inline)) {
        const int pidx = stid / C8, pw = pidx >> p.th_shift, ph = pidx - (pw << p.th_shift);
        unsigned char* dst = sA + (cs & 1) * abytes + (pw + 1) * colb + (ph + 1) * RS + (stid & (C8 - 1)) * 16;
        const int step = (RSTEP >> p.th_shift) * colb;
#pragma unroll
        for (int i = 0; i < RCH; ++i) *reinterpret_cast<uint4*>(dst + i * step) = rr[i];
    };

    // ---- this matrix wave's weight stream (channel tile 4 * nt + wn): [NCC][9 taps][4 k-steps] then [NCB][4], 1 KiB each ------------------
    const int nsteps = NCC * CST + NCB * SPT;
    const unsigned char* wptr = reinterpret_cast<const unsigned char*>(p.wpk) + (size_t)(nt * WN + wn) * nsteps * 1024;
    const unsigned woff = lane * 16 + 4096;
    auto w_load = [&](const unsigned char* base, int idx) __attribute__((always_inline)) {      // fragment idx in [0, 16)
        return *reinterpret_cast<const bf16x8*>(base + (idx / 8) * 8192 + woff + ((idx % 8) * 1024 - 4096));
    };
    bf16x8 wr[G];

    // ---- stage A: the staging waves request the first halo chunk, the matrix waves their first row of weight fragments ---------------------
    if (producer) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int q = stid + i * SNT;
            const int slot = q / C8;
            const int vwl = (slot * p.magic_thv) >> 20, vhl = slot - vwl * THv;
            const int vh = h0 - 1 + vhl;
            int vw = w0 - 1 + vwl;
            vw = vw < 0 ? vw + Wv : (vw >= Wv ? vw - Wv : vw);
            const bool ok = q < atotal && vh >= 0 && vh < Hv;
            apix[i] = ok ? ((b * p.Win + (vw >> upshift)) * p.Hin + (vh >> upshift)) : -1;
        }
        if (NCT > 0) load_a(0);
    } else {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            wr[j] = w_load(wptr, j);
            __builtin_amdgcn_sched_barrier(0);      // issued here and in this order: the counted waits below rely on it
        }
        wptr += G * 1024;
        if (temb_tab && tid < BN)
            bias_v += temb_tab[(size_t)(temb_step * temb_rps + (temb_ps ? b : 0)) * temb_ld + nt * BN + tid];
    }
    RLDM_STAMP();
    // ---- stage B (all 512 threads): every channel's thread folds its own group -------------------------------------------------------------
    if (gn) {
        double* sD = reinterpret_cast<double*>(sA);             // scratch: [2][Cin] doubles (the halo is not written yet)
        const int cpg = Cin / p.gn_groups;
        if (tid < Cin) {
            sD[tid] = gS;
            sD[Cin + tid] = gSS;
        }
        __syncthreads();
        float ga = 0.f, gs = 0.f;
        if (tid < Cin) {
            const int g0 = ((tid * p.magic_cpg) >> 20) * cpg;
            double S = 0.0, SS = 0.0;
            for (int i = 0; i < cpg; ++i) {
                S += sD[g0 + i];
                SS += sD[Cin + g0 + i];
            }
            const double inv_n = (double)p.gn_inv_n;
            const double mean = S * inv_n;
            double var = SS * inv_n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            ga = g_gamma * __builtin_amdgcn_rsqf((float)var + p.gn_eps);
            gs = g_beta - (float)mean * ga;
            sGa[tid] = ga;
            sGs[tid] = gs;
        }
        __syncthreads();                        // affine visible; sD fully consumed before the halo is written
    }
    RLDM_STAMP();
    if (tid < BN) sBias[tid] = bias_v;

    f32x16 acc[MI];
    int xoff[MI];
    // (chunk k of the sequence main, residual through the halo registers: the first residual chunk only; later ones use rreg below)
    auto load_next = [&](int k) __attribute__((always_inline)) {
        if (RLDM_TDBG(p, 16384)) return;
        if (k < NCC) load_a(k); else load_r(k, areg);
    };
    auto store_next = [&](int k) __attribute__((always_inline)) {
        if (RLDM_TDBG(p, 8192)) return;
        if (k < NCC) store_a(k); else store_r(k, areg);
    };
    if (producer) {
        if (NCT > 0) store_a(0);
    } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int pidx = mi * 32 + l31;
            const int pw = pidx >> p.th_shift, ph = pidx - (pw << p.th_shift);
            xoff[mi] = pw * colb + ph * RS + kh * 16;
        }
    }
    RLDM_STAMP();
    lds_barrier_b();                            // sBias and halo chunk 0 are written
    RLDM_STAMP();

    if (producer) {
        // ---- staging waves: chunk cs + 1 while the matrix waves consume chunk cs; one barrier per chunk on both sides -------------------------
        for (int cs = 0; cs < NCC; ++cs) {
            // (requesting chunk cs + 2 here and storing chunk cs + 1 from the previous request measured SLOWER: 58 spilled registers)
            if (cs + 1 < NCT) { load_next(cs + 1); store_next(cs + 1); }
            lds_barrier_b();
        }
        // residual chunks are 32 MFMAs per matrix wave (~1 k cycles): three of them in flight in registers
        uint4 rreg[3][RCH];
        if (1 < NCB) load_r(NCC + 1, rreg[1]);
        if (2 < NCB) load_r(NCC + 2, rreg[2]);
        if (3 < NCB) load_r(NCC + 3, rreg[0]);
        for (int rc0 = 0; rc0 < NCB; rc0 += 3) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int rc = rc0 + r;
                if (rc < NCB) {
                    if (rc + 1 < NCB) store_r(NCC + rc + 1, rreg[(r + 1) % 3]);
                    if (rc + 4 < NCB) load_r(NCC + rc + 4, rreg[(r + 1) % 3]);
                    lds_barrier_b();
                }
            }
        }
    } else {
        // ---- matrix waves ---------------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const float4 bv = *reinterpret_cast<const float4*>(sBias + wn * 32 + 8 * r4 + 4 * kh);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                acc[mi][r4 * 4 + 0] = bv.x; acc[mi][r4 * 4 + 1] = bv.y;
                acc[mi][r4 * 4 + 2] = bv.z; acc[mi][r4 * 4 + 3] = bv.w;
            }
        }
        // One set of pixel fragments: the fragment of (step c + 1, mi) is requested right behind the MFMA that consumed (step c, mi) -- the
        // other seven MFMAs of the step (224+ cycles) are its lead.  `cur` = per-lane LDS address of the pixel at the current row of taps.
        bf16x8 xr[MI];
        auto tap_row = [&](int (&cur)[MI], int ti) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < ROW; ++j) {
                const int c = ti * ROW + j, slot = c % G;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[slot], xr[mi], acc[mi], 0, 0, 0);
                    if (c + 1 < CST) {
                        if (j == ROW - 1) cur[mi] += colb;                  // (the next row of taps: one halo column on)
                        const int r = (j + 1) % ROW;
                        xr[mi] = *reinterpret_cast<const bf16x8*>(smem + cur[mi] + (r / SPT) * RS + (r % SPT) * 32);
                    }
                }
                wr[slot] = w_load(wptr, slot);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (c + 1 < CST) {
                        if (j == ROW - 1) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((ti * ROW + ROW) % G == 0) wptr += G * 1024;
        };
        for (int cs = 0; cs < NCC; ++cs) {
            int cur[MI];
            const int boff = (cs & 1) * abytes;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                cur[mi] = xoff[mi] + boff;
                xr[mi] = *reinterpret_cast<const bf16x8*>(smem + cur[mi]);
            }
            __builtin_amdgcn_sched_barrier(0);
            tap_row(cur, 0);
            tap_row(cur, 1);
            tap_row(cur, 2);
            lds_barrier_b();                    // chunk cs consumed, chunk cs + 1 written
        }
        constexpr int RCR = G / SPT;            // residual chunks per ring revolution (3)
        for (int rc0 = 0; rc0 < NCB; rc0 += RCR) {
#pragma unroll
            for (int r = 0; r < RCR; ++r) {
                const int rc = rc0 + r;
                if (rc < NCB) {
                    const int cs = NCC + rc;
                    int xc[MI];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) xc[mi] = xoff[mi] + (cs & 1) * abytes + colb + RS;
#pragma unroll
                    for (int ks = 0; ks < SPT; ++ks) {
                        bf16x8 xf[MI];
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) xf[mi] = *reinterpret_cast<const bf16x8*>(smem + xc[mi] + ks * 32);
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
                            acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[r * SPT + ks], xf[mi], acc[mi], 0, 0, 0);
                        wr[r * SPT + ks] = w_load(wptr, r * SPT + ks);
                    }
                    lds_barrier_b();
                }
                if (r == RCR - 1) wptr += G * 1024;
            }
        }
    }
    RLDM_STAMP();

    // ---- epilogue: bf16 -> LDS [pixel][channel] (matrix waves) -> 16-byte coalesced stores + statistics (all 512 threads) ------------------
    unsigned char* sE = smem;
    if (!producer) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int pidx = mi * 32 + l31;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int chl = wn * 32 + 8 * r4 + 4 * kh;
                uint2 o;
                o.x = pack_bf16x2(acc[mi][r4 * 4 + 0], acc[mi][r4 * 4 + 1]);
                o.y = pack_bf16x2(acc[mi][r4 * 4 + 2], acc[mi][r4 * 4 + 3]);
                *reinterpret_cast<uint2*>(sE + pidx * ERS + chl * 2) = o;
            }
        }
    }
    lds_barrier_b();
    RLDM_STAMP();
    const int c8 = tid % NC8;
    const int chg = nt * BN + c8 * 8;
    {
        const int g = tid / NC8;                                            // pixel 0..31 of the pass: (pw, ph) = (g >> 3, g & 7)
        bf16_t* yp = p.y + (((size_t)b * p.Wout + w0 + (g >> 3)) * p.Hout + h0 + (g & 7)) * p.y_ld + chg;
        const size_t ystep = (size_t)4 * p.Hout * p.y_ld;
#pragma unroll
        for (int i = 0; i < BM / (NT / NC8); ++i) {
            *reinterpret_cast<uint4*>(yp) = *reinterpret_cast<const uint4*>(sE + (g + i * (NT / NC8)) * ERS + c8 * 16);
            yp += ystep;
        }
    }
    if (p.y_stats) {
        constexpr int NCP = BN / 2, NG = NT / NCP, PPG = BM / NG;          // 64 channel pairs x 8 pixel groups of 32
        const int cp = tid % NCP, pg = tid / NCP;
        float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll 8
        for (int j = 0; j < PPG; ++j) {
            const uint32_t w2 = *reinterpret_cast<const uint32_t*>(sE + (pg * PPG + j) * ERS + cp * 4);
            const float a0 = bf16lo(w2), a1 = bf16hi(w2);
            s0 += a0; s1 += a1;
            q0 += a0 * a0; q1 += a1 * a1;
        }
        float* sS = reinterpret_cast<float*>(sE + BM * ERS);                // [8 waves][2][BN]
        *reinterpret_cast<float2*>(sS + (wave * 2 + 0) * BN + cp * 2) = make_float2(s0, s1);
        *reinterpret_cast<float2*>(sS + (wave * 2 + 1) * BN + cp * 2) = make_float2(q0, q1);
        lds_barrier_b();
        if (tid < 2 * BN) {
            const int kind = tid / BN, c = tid - kind * BN;
            float S = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) S += sS[(w * 2 + kind) * BN + c];
            reinterpret_cast<float*>(p.y_stats + ((size_t)b * tiles_img + mt) * p.N + nt * BN + c)[kind] = S;
        }
    }
    RLDM_STAMP();
    if constexpr (TRUNK) trunk_arrive(seam, tid);
#ifdef RLDM_ABLATE
    if (TRUNK && seam.ts && tid == 0) {
        RLDM_STAMP();
        for (int i = 0; i < 12; ++i) seam.ts[i] = i < tsn ? tsv[i] : 0ull;
    }
    {
        const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const int total = gridDim.x * gridDim.y * gridDim.z;
        const int slot = lin < 2 ? lin : (lin - total / 2 >= 0 && lin - total / 2 < 2 ? 2 + lin - total / 2 : -1);
        if (!TRUNK && p.ts && slot >= 0 && tid == 0)
            for (int i = 0; i < 12; ++i) p.ts[slot * 64 + i] = i < tsn ? tsv[i] : 0ull;
        if (!TRUNK && p.ts && tid == 0 && lin < 2048) {
            unsigned xcc, hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 8, 8)" : "=s"(hw));
            p.ts[256 + 2 * lin] = t_real0;
            p.ts[257 + 2 * lin] = __builtin_amdgcn_s_memrealtime() | ((unsigned long long)((xcc << 8) | hw) << 48);
        }
    }
#endif
#undef RLDM_STAMP
}

}  // namespace rldm
