// Body of conv_small_kernel (conv_small.hip) as a device function, shared with the persistent trunk kernel (trunk.hip).
// TRUNK = false: one launch = one conv (arguments from the kernel-argument segment).
// TRUNK = true : one PHASE of a persistent launch: the arguments live in device memory, the weight ring arrives prefetched, the
//                activations another workgroup of the image's cluster published are read past the L1 (nontemporal loads), and the
//                phase ends by publishing (arrive on the cluster's counter) instead of by a kernel boundary.
#pragma once
#include "kernels.h"
#include "trunk_seam.h"

#ifndef RLDM_H16_NB
#define RLDM_H16_NB 3               /* k-steps of pixel fragments per LDS block of the 16-channel K loop (x 4 fragments each) */
#endif

namespace rldm {

__device__ __forceinline__ void lds_barrier_s() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// BM = 32 * MI pixels (64: the 64x4 / 32x2 levels and the pointwise convs; 128: the 128x8 level; 32: 32x1 images -- the lowest nuScenes level), BN = 32 * NWN channels, C_in = 16 * KG * CPT channels (CPT = a k-group's steps per tap), 512 threads.
// TAPS == 9: 3x3 over a pre-activated input.  TAPS == 1: pointwise (attention q/k/v and output projections); there the
// GroupNorm affine (no separate launch: one FMA per element while the tile is on its way to LDS) is folded in.
// (round 5) H16: a 16-channel tile -- an image of the 32x2 level is then 16 workgroups, the level runs on all 256 CUs at batch 16 and a
// workgroup's K loop, weight stream and epilogue halve.  v_mfma_f32_16x16x32_bf16: a k-step is 32 channels of a tap (k-group kg owns the
// 32-channel groups kg, kg + 8, ...), a weight fragment (16 channels x 32 k = 1 KiB, lane l: channel l & 15, k = 8 (l >> 4) .. + 8) feeds the
// tile's four 16-pixel fragments; only image-owning 64-pixel tiles (NWN == 1, MI == 2), C_in = 256 * CPT.
template <int NWN, int CPT, int TAPS, int MI, bool TRUNK, int PF = kTrunkPrefetch, bool H16 = false>
__device__ __forceinline__ void conv_small_body(const ConvParams& p, const int nt, const int mt, const int b, bf16x8 (&wpf)[kTrunkPrefetch],
                                                const TrunkSeam& seam) {
    static_assert(!H16 || (NWN == 1 && MI == 2 && CPT <= 2), "16-channel tiles: image-owning 64-pixel tiles over 256 / 512 input channels");
    constexpr int NT = 512, KG = 8 / NWN, BM = 32 * MI, BN = H16 ? 16 : 32 * NWN;
    constexpr int KSC = H16 ? 32 : 16;          // input channels per k-step
    constexpr int CIN = KSC * KG * CPT, C8 = CIN / 8;
    constexpr int RSM = CIN * 2 + 16;          // image row stride (bytes): C8 + 1 16-byte slots, odd
    constexpr int HALO = TAPS == 9 ? 1 : 0;
    // taps per unrolled group: all nine when a k-group has <= 2 steps per tap (the ring then holds the wave's WHOLE main stream --
    // 9 / 18 fragments requested at kernel entry, no weight wait inside the K loop), a tap row, or one tap for wide inputs
    constexpr int TPG = TAPS == 1 ? 1 : ((CPT <= 2 && MI <= 2) ? 9 : (CPT <= 4 ? 3 : 1));
    constexpr int G = TPG * CPT;               // k-steps per group = weight fragments in flight per wave
    constexpr int NGRP = TAPS / TPG;
    constexpr int PFX = (G % 3 == 0) ? 3 : 2;  // pixel fragments read ahead (LDS); divides G
    constexpr int RMAX = G < 8 ? G : 8;        // residual-phase steps per k-group (<= G: they arrive in the ring)
    // accumulator sets: consecutive MFMAs never share an accumulator (64 pixels x 32 channels -- the instances the persistent
    // trunk carries: one set, the two pixel tiles alternate, and three instances + a prefetched ring fit 256 registers)
    constexpr int AS = (MI >= 4 || (NWN == 1 && MI == 2)) ? 1 : 2;     // (the same in both modes: identical results bit for bit)
    constexpr int HB = BM < 64 ? BM : 64;      // epilogue half-tile: pixels exchanged through LDS at a time (32-pixel tiles: all of it)
    constexpr int FRS = BN * 4 + 16, NC8 = BN / 8;    // fp32 partial-sum image: [k-group][HB pixels][FRS bytes]
    constexpr int LPS = C8 <= 16 ? 16 : (C8 <= 32 ? 32 : 64), SPI = 64 / LPS;   // lanes per halo slot, slots per instruction
    // staging batch: columns per wave x row groups in flight (128-pixel tiles are 16 wide: 18 halo columns, 3 per wave,
    // and all of a wave's pieces are requested before the first is stored)
    constexpr int NCW = MI >= 4 ? 3 : 5, KB = MI >= 4 ? (10 + SPI - 1) / SPI : 4;
    static_assert(C8 <= 64 && TAPS % TPG == 0 && (H16 || (G % PFX == 0 && PFX <= G)), "shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tid_ = threadIdx.x;
    // (a phase of the persistent launch: the thread index is made opaque per phase, or every lane-dependent constant of every
    //  instance is hoisted out of the phase loop and lives in registers across all of them)
    if constexpr (TRUNK) asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % NWN, kg = wave / NWN;
    const int kh = lane >> 5, l31 = lane & 31;
#ifdef RLDM_ABLATE
    unsigned long long tsv[16];
    int tsn = 0;
#define RLDM_STAMP() if (tsn < 16) tsv[tsn++] = __builtin_amdgcn_s_memtime()
#else
#define RLDM_STAMP()
#endif
    RLDM_STAMP();
#ifdef RLDM_ABLATE
    const unsigned long long t_real0 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- which tile ---------------------------------------------------------------------------------------------
    // grid = (channel tiles, pixel tiles of an image, images).  Workgroups go to the XCDs round-robin in x-fastest order, so
    // with 4 or 8 channel tiles all blocks that stream the same weight slice share an XCD (one L2 copy of it)
    const int tiles_h = p.tiles_h, tiles_img = p.tiles_img;       // tiles_h is a power of two
#ifdef RLDM_ABLATE
    asm volatile("s_nop 0" :: "s"(tiles_h));                      // (the stamp below sits behind the kernel-argument wait)
#endif
    RLDM_STAMP();                               // kernel arguments arrived
    const int tw = mt >> (31 - __builtin_clz(tiles_h)), th = mt & (tiles_h - 1);
    const int w0 = tw * p.TW, h0 = th * p.TH;
    const int npx = p.TW * p.TH;               // == BM
    const int R = p.R0 + p.R1, R8 = R >> 3;
    const int RSR = R * 2 + 16;                // residual image row stride
    const int THv = p.TH + 2 * HALO, TWv = p.TW + 2 * HALO;
    const int colb = p.colb;
    const int abytes = TWv * colb;

    unsigned char* sA = smem;                                   // [TWv][colb]: activated input + halo
    unsigned char* sR = sA + abytes;                            // [npx][RSR]: raw residual-phase input
    float* sBias = reinterpret_cast<float*>(sR + npx * RSR);    // BN

    // ---- bias (+ time embedding row): fetched now, parked in LDS after the staging loop ---------------------------------
    // (the sampler's step index selects the time-embedding row: requested first, as a vector load -- common.h -- and used behind
    //  the weight ring; the dependent row load then costs wave 0 nothing ahead of its ring)
    const float* const temb_tab = TRUNK ? seam.temb : p.temb;
    const int* const step_ptr = TRUNK ? seam.step_ptr : p.step_ptr;
    const int temb_rps = TRUNK ? seam.temb_rows_per_step : p.temb_rows_per_step;
    const int temb_ps = TRUNK ? seam.temb_per_sample : p.temb_per_sample;
    const int temb_ld = TRUNK ? seam.temb_ld : p.temb_ld;
    const int temb_step = (temb_tab && step_ptr && tid < BN) ? load_step_vector(step_ptr) : 0;
    float bias_v = 0.f, temb_v = 0.f;
    if (tid < BN) bias_v = p.bias[nt * BN + tid];

    // ---- producer-side GroupNorm: the consumers' gamma / beta of channel tid % BN (view tid / BN), requested now ---------
    float nv_gamma = 0.f, nv_beta = 0.f;
#pragma unroll
    for (int v = 0; v < 3; ++v)         // (constant indices: a run-time index into the by-value argument would copy it to scratch)
        if (v < p.nviews && tid / BN == v) {
            nv_gamma = p.nv[v].gamma[nt * BN + tid % BN];
            nv_beta = p.nv[v].beta[nt * BN + tid % BN];
        }

    // ---- persistent trunk: everything above is independent of the previous phase; everything below reads what it published ----
    if constexpr (TRUNK) {
        RLDM_STAMP();                           // (trunk) small requests issued, waiting for the cluster
        trunk_wait(seam, tid);
        RLDM_STAMP();                           // (trunk) the previous phase has been published
    }

    // ---- TAPS == 1: the GroupNorm inputs of channel `tid` (statistics partials of the producer, gamma, beta), requested now
    // (3x3: single-input convs only; concatenated inputs come pre-activated.  Trunk phases: pre-activated inputs only)
    const bool gn = (TRUNK && NWN == 1) ? false : p.st0 != nullptr;       // (image-owning trunk phases: pre-activated inputs only)
    double gS = 0.0, gSS = 0.0;
    float g_gamma = 0.f, g_beta = 0.f;
    // (round 4) a CONCATENATED 3x3 input -- an up-block's conv1: cat[x0, x1] with C0 + C1 == CIN -- is normalised here as well instead of by a
    // gn_apply launch / phase in front of the conv: channel tid belongs to x0 (statistics st0, P0 partials) or to x1 (st1, P1)
    const int nC0 = p.C1 != 0 ? p.C0 : CIN, nC1 = p.C1;
    if (gn && tid < CIN) {
        const float2* const gs0p = p.st0;       // (locals: selecting between fields of `p` by address would copy it to scratch)
        const float2* const gs1p = p.st1;
        const int nP0 = p.P0, nP1 = p.P1;
        const bool first_t = tid < nC0;
        const int Ct = first_t ? nC0 : nC1;
        const int P = first_t ? nP0 : nP1;
        const float2* src = (first_t ? gs0p : gs1p) + (size_t)b * P * Ct + (first_t ? tid : tid - nC0);
        int q = 0;
        for (; q + 8 <= P; q += 8) {            // (8 pixel tiles per image -- 128x4 / 128x8 inputs: one round trip instead of two)
            float2 u[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) u[j] = ld_act8<TRUNK>(src + (size_t)(q + j) * Ct);
#pragma unroll
            for (int j = 0; j < 8; ++j) { gS += (double)u[j].x; gSS += (double)u[j].y; }
        }
        for (; q + 4 <= P; q += 4) {
            float2 u[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j] = ld_act8<TRUNK>(src + (size_t)(q + j) * Ct);
#pragma unroll
            for (int j = 0; j < 4; ++j) { gS += (double)u[j].x; gSS += (double)u[j].y; }
        }
        for (; q < P; ++q) {
            const float2 u = ld_act8<TRUNK>(src + (size_t)q * Ct);
            gS += (double)u.x;
            gSS += (double)u.y;
        }
        g_gamma = p.gn_gamma[tid];
        g_beta = p.gn_beta[tid];
    }

    RLDM_STAMP();                               // small requests (bias, statistics partials, affines) issued
    // ---- this wave's weight stream: [9 * CPT main steps (tap-major)][RPT residual steps], 1 KiB each; lane l holds channel
    // l & 31, k = 8 * (l >> 5) .. + 8 of the step.  The first G fragments are requested before anything else.
    const int RPT = (R / KSC) / KG;             // residual steps of this k-group (<= RMAX)
    const int nmine = TAPS * CPT + RPT;
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.wpk) +
                                 ((size_t)((nt * NWN + wn) * KG + kg) * nmine) * 1024;      // uniform
    unsigned woff[(G + 7) / 8];                 // lane offsets: immediates of +-4 KiB around them reach 8 fragments each
#pragma unroll
    for (int q = 0; q < (G + 7) / 8; ++q) woff[q] = lane * 16 + 4096 + q * 8192;
    auto w_load = [&](const unsigned char* base, int idx) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8*>(base + woff[idx / 8] + ((idx % 8) * 1024 - 4096));
    };
    bf16x8 wr[G];
    if constexpr (TRUNK) {                      // requested by the previous phase, behind its K loop
#pragma unroll
        for (int j = 0; j < G; ++j) {
            if (j < PF) wr[j] = wpf[j];
            else wr[j] = w_load(wbase, j);      // (the rest of a long ring: in flight during the seam and the gather; PF == 0 --
                                                //  multi-tile clusters, whose instances leave no registers to carry a ring across
                                                //  phases: the whole ring, requested here like a stand-alone launch's)
        }
    } else {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            wr[j] = w_load(wbase, j);
            __builtin_amdgcn_sched_barrier(0);  // issued here and in this order: the K loop's counted waits rely on it
        }
    }

    if (temb_tab && tid < BN)
        temb_v = temb_tab[(size_t)(temb_step * temb_rps + (temb_ps ? b : 0)) * temb_ld + nt * BN + tid];

    RLDM_STAMP();
    // ---- the input tile, once: global -> LDS; wrap on W, zeros on H.  Wave w copies halo columns w, w + 8, ...; a wave
    // instruction moves SPI rows x C8 16-byte pieces of a column, so a lane's row, channel and both offsets never change.
    // TAPS == 1 with statistics: the GroupNorm affine of the image is derived while the first loads are in flight and
    // applied on the way into LDS ----
    const bool whole_image = (TRUNK && NWN == 1) || (!TRUNK && tiles_img == 1 && p.up == 1 && !gn && p.TW == p.Win && p.TH == p.Hin && 2 * p.TH * C8 <= NT);
    if (whole_image) {
        // the tile IS the image and arrives ready (pre-activated, or no norm): [npx][CIN] is one contiguous block -- a linear copy
        // (thread-constant piece index -> pixel / channel by constant divisions, no per-piece predicates), the two wrap-around halo
        // columns re-read from the image's last / first column, zero rows written without a load
        constexpr int NPIECE = BM * C8, NLD = (NPIECE + NT - 1) / NT;
        const unsigned char* img = reinterpret_cast<const unsigned char*>(p.x0) + (size_t)b * npx * (CIN * 2);
        uint4 v[NLD], hv[HALO ? 1 : 1];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int q = tid + i * NT;
            if (NPIECE % NT == 0 || q < NPIECE) v[i] = ld_act16<TRUNK>(img + (size_t)q * 16);
        }
        // halo columns (3x3): piece hq = (side, row, c8); side 0 -> LDS column 0 <- image column W - 1, side 1 -> column TW + 1 <- 0
        const int nhalo = HALO ? 2 * p.TH * C8 : 0;
        const int hside = tid / (p.TH * C8), hrem = tid - hside * (p.TH * C8);
        if (HALO && tid < nhalo)
            hv[0] = ld_act16<TRUNK>(img + ((size_t)(hside ? 0 : (p.TW - 1) * p.TH) * C8 + hrem) * 16);
        RLDM_STAMP();                           // tile loads issued
        if (HALO) {                             // zero rows 0 and TH + 1 of every halo column
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            for (int q = tid; q < TWv * 2 * C8; q += NT) {
                const int col = q / (2 * C8), r = q - col * (2 * C8);
                const int row = r < C8 ? 0 : THv - 1, c8 = r < C8 ? r : r - C8;
                *reinterpret_cast<uint4*>(sA + col * colb + row * RSM + c8 * 16) = z;
            }
        }
        RLDM_STAMP();
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int q = tid + i * NT;
            const int pix = q / C8, c8 = q - pix * C8;
            const int pw = pix >> p.th_shift, ph = pix - (pw << p.th_shift);
            if (NPIECE % NT == 0 || q < NPIECE)
                *reinterpret_cast<uint4*>(sA + (pw + HALO) * colb + (ph + HALO) * RSM + c8 * 16) = v[i];
        }
        if (HALO && tid < nhalo) {
            const int row = hrem / C8, c8 = hrem - row * C8;
            *reinterpret_cast<uint4*>(sA + (hside ? TWv - 1 : 0) * colb + (row + 1) * RSM + c8 * 16) = hv[0];
        }
    } else if constexpr (!TRUNK || NWN > 1) {
        const int c8 = lane & (LPS - 1), rsub = lane / LPS;
        const bool laneok = c8 < C8;
        // this lane's 8 channels live in x0 or in x1 (a concatenated input; one tensor: always x0): per-lane base, pixel pitch, channel offset
        const bf16_t* const gx0 = p.x0;
        const bf16_t* const gx1 = p.x1;
        const bool first_l = c8 * 8 < nC0;
        const unsigned char* xg = reinterpret_cast<const unsigned char*>(first_l ? gx0 : gx1);
        const unsigned pitch_l = (unsigned)(first_l ? nC0 : nC1) * 2u;
        const unsigned choff_l = (unsigned)(first_l ? c8 * 8 : c8 * 8 - nC0) * 2u;
        const int KC = (THv + SPI - 1) / SPI;   // instructions per column
        const int ups = p.up - 1;               // nearest x2 folded into the source indexing
        float ga[8], gs[8];
        for (int k0 = 0; k0 < KC; k0 += KB) {
            uint4 v[KB][NCW];
            int ldo[KB];
            bool rowok[KB], inimg[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                // slot -> halo row, rotated by one (3x3): a wave instruction moves SPI consecutive slots, and with rows
                // 1, 2, ... first the zero rows above and below a full-height tile (32x2 / 64x4 images: rows 0 and THv - 1)
                // share an instruction instead of each wasting half of one on lanes that skip the GroupNorm + SiLU math
                rowok[kb] = false;
                inimg[kb] = false;
                ldo[kb] = 0;
                if ((k0 + kb) >= KC) continue;              // (uniform: a row group this tile does not have costs a branch,
                                                            //  not the predicate / zero-fill code of its NCW slots)
                const int slot = (k0 + kb) * SPI + rsub;
                const int vhl = HALO ? (slot + 1 >= THv ? slot + 1 - THv : slot + 1) : slot;
                const int vh = h0 - HALO + vhl;                 // row / column of the (nearest-x2: virtual) input image
                rowok[kb] = laneok && slot < THv;
                inimg[kb] = rowok[kb] && vh >= 0 && vh < (p.Hin << ups);
                const unsigned goff = (unsigned)(vh >> ups) * pitch_l + choff_l;
                ldo[kb] = vhl * RSM + c8 * 16;
#pragma unroll
                for (int j = 0; j < NCW; ++j) {
                    const int col = wave + 8 * j;
                    if (col >= TWv) continue;               // (uniform per wave)
                    v[kb][j] = make_uint4(0u, 0u, 0u, 0u);
                    int vw = w0 - HALO + col;
                    const int Wv = p.Win << ups;
                    vw = vw < 0 ? vw + Wv : (vw >= Wv ? vw - Wv : vw);
                    // 32-bit byte offset of the column (uniform; the tensors on this route are far below 4 GiB): scalar base +
                    // one VGPR offset per load instead of a 64-bit pointer per column
                    const unsigned coff = (unsigned)((b * p.Win + (vw >> ups)) * p.Hin) * pitch_l;
                    if (inimg[kb]) v[kb][j] = ld_act16<TRUNK>(xg + (coff + goff));
                }
            }
            if (k0 == 0) { RLDM_STAMP(); }      // tile loads issued
            if (gn && k0 == 0) {
                // per-channel sums -> (every channel's thread folds its own group: no serial phase) mean / rstd -> a*x + s
                double* sD = reinterpret_cast<double*>(smem + abytes + 64);      // [2*CIN], behind the image
                float* sG = reinterpret_cast<float*>(sD + 2 * CIN);              // [2][CIN]
                const int cpg = CIN / p.gn_groups;
                if (tid < CIN) {
                    sD[tid] = gS;
                    sD[CIN + tid] = gSS;
                }
                __syncthreads();
                if (tid < CIN) {
                    const int g0 = ((tid * p.magic_cpg) >> 20) * cpg;
                    double S = 0.0, SS = 0.0;
                    for (int i = 0; i < cpg; ++i) {
                        S += sD[g0 + i];
                        SS += sD[CIN + g0 + i];
                    }
                    const double inv_n = (double)p.gn_inv_n;
                    const double mean = S * inv_n;
                    double var = SS * inv_n - mean * mean;
                    var = var < 0.0 ? 0.0 : var;
                    const float a = g_gamma * __builtin_amdgcn_rsqf((float)var + p.gn_eps);
                    sG[tid] = a;
                    sG[CIN + tid] = g_beta - (float)mean * a;
                }
                __syncthreads();
                if (laneok) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ga[e] = sG[c8 * 8 + e]; gs[e] = sG[CIN + c8 * 8 + e]; }
                }
            }
            if (k0 == 0) { RLDM_STAMP(); }      // affine ready
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                if ((k0 + kb) >= KC) continue;
#pragma unroll
                for (int j = 0; j < NCW; ++j) {
                    const int col = wave + 8 * j;
                    if (col >= TWv) continue;
                    uint4 o = v[kb][j];
                    if (gn && inimg[kb]) {
                        float f0 = bf16lo(o.x) * ga[0] + gs[0], f1 = bf16hi(o.x) * ga[1] + gs[1];
                        float f2 = bf16lo(o.y) * ga[2] + gs[2], f3 = bf16hi(o.y) * ga[3] + gs[3];
                        float f4 = bf16lo(o.z) * ga[4] + gs[4], f5 = bf16hi(o.z) * ga[5] + gs[5];
                        float f6 = bf16lo(o.w) * ga[6] + gs[6], f7 = bf16hi(o.w) * ga[7] + gs[7];
                        if (p.silu) {
                            silu_x8(f0, f1, f2, f3, f4, f5, f6, f7);
                        }
                        o.x = pack_bf16x2(f0, f1); o.y = pack_bf16x2(f2, f3);
                        o.z = pack_bf16x2(f4, f5); o.w = pack_bf16x2(f6, f7);
                    }
                    if (rowok[kb]) *reinterpret_cast<uint4*>(sA + col * colb + ldo[kb]) = o;
                }
            }
        }
        RLDM_STAMP();                           // tile normalised and stored
    }
    {
        // residual-phase input: raw cat[r0, r1], the tile's own pixels only; a wave instruction moves 64 / LPR pixels
        if (R8 > 0) {
            const int lgr = R8 <= 16 ? 4 : (R8 <= 32 ? 5 : 6);          // log2(lanes per pixel)
            const int rc8 = lane & ((1 << lgr) - 1), psub = lane >> lgr;
            const int ppi = 64 >> lgr;                                   // pixels per instruction
            const int c = rc8 * 8;
            const bf16_t* gr0 = p.r0;           // (locals: selecting between fields of `p` by address would copy it to scratch)
            const bf16_t* gr1 = p.r1;
            const int nR0 = p.R0, nR1 = p.R1;
            const bool first = c < nR0;
            const unsigned char* lbase = reinterpret_cast<const unsigned char*>(first ? gr0 + c : gr1 + (c - nR0));
            const unsigned ld2 = (unsigned)(first ? nR0 : nR1) * 2u;
            const int pix0 = (b * p.Win + w0) * p.Hin + h0;
            constexpr int NBR = 8;              // <= 8 instructions per wave and 64 pixels (R <= 512)
            for (int hp = 0; hp < (BM + 63) / 64; ++hp) {
                uint4 rv[NBR];
#pragma unroll
                for (int u = 0; u < NBR; ++u) {
                    if ((wave + 8 * u) * ppi >= HB) continue;       // (uniform: instructions past the half-tile cost a branch)
                    const int pl = (wave + 8 * u) * ppi + psub, pidx = hp * 64 + pl;
                    const int pw = pidx >> p.th_shift, ph = pidx - (pw << p.th_shift);
                    rv[u] = make_uint4(0u, 0u, 0u, 0u);
                    if (pl < HB && rc8 < R8)
                        rv[u] = ld_act16<TRUNK>(lbase + (size_t)(unsigned)(pix0 + pw * p.Hin + ph) * ld2);
                }
#pragma unroll
                for (int u = 0; u < NBR; ++u) {
                    if ((wave + 8 * u) * ppi >= HB) continue;
                    const int pl = (wave + 8 * u) * ppi + psub, pidx = hp * 64 + pl;
                    if (pl < HB && rc8 < R8) *reinterpret_cast<uint4*>(sR + pidx * RSR + rc8 * 16) = rv[u];
                }
            }
        }
    }
    if (tid < BN) sBias[tid] = bias_v + temb_v;
    lds_barrier_s();
    RLDM_STAMP();

    // ---- accumulators: k-group 0 carries bias + temb ---------------------------------------------------------------
    f32x16 acc[H16 ? 1 : AS][H16 ? 1 : MI];
    constexpr int PT = 2 * MI;                  // (H16) 16-pixel fragments of the tile
    f32x4 acc16[H16 ? PT : 1];
    const int l15 = lane & 15, k4 = lane >> 4;  // (H16) lane = (pixel / channel l15, 8-wide k slot / 4-channel row group k4)
    if constexpr (H16) {
        float4 bv = *reinterpret_cast<const float4*>(sBias + 4 * k4);
        if (kg != 0) bv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) { acc16[pt][0] = bv.x; acc16[pt][1] = bv.y; acc16[pt][2] = bv.z; acc16[pt][3] = bv.w; }
    } else {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        float4 bv = *reinterpret_cast<const float4*>(sBias + wn * 32 + 8 * r4 + 4 * kh);
        if (kg != 0) bv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int a = 0; a < AS; ++a) {
                const float z = a == 0 ? 1.f : 0.f;
                acc[a][mi][r4 * 4 + 0] = bv.x * z; acc[a][mi][r4 * 4 + 1] = bv.y * z;
                acc[a][mi][r4 * 4 + 2] = bv.z * z; acc[a][mi][r4 * 4 + 3] = bv.w * z;
            }
    }
    }

    // ---- identity residual (y = conv + x): fetched now, added in fp32 in the epilogue ------------------------------------
    constexpr int NHALF = BM / HB;
    constexpr int NPASS = (HB * NC8 + NT - 1) / NT;             // epilogue items (pixel, 8 channels) per thread and half-tile
    uint4 resv[NHALF][NPASS];
#pragma unroll
    for (int hp = 0; hp < NHALF; ++hp)
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            const int pl = tid / NC8 + q * (NT / NC8), pidx = hp * HB + pl;
            const int pw = pidx >> p.th_shift, ph = pidx - (pw << p.th_shift);
            resv[hp][q] = make_uint4(0u, 0u, 0u, 0u);
            if (p.res && pl < HB)
                resv[hp][q] = ld_act16<TRUNK>(
                    p.res + (((size_t)b * p.Wout + (w0 + pw)) * p.Hout + (h0 + ph)) * p.N + nt * BN + (tid % NC8) * 8);
        }

    // ---- barrier-free K loop ------------------------------------------------------------------------------------------
    if constexpr (H16) {
        // the ring holds the wave's WHOLE main stream (9 * CPT <= 18 fragments: requested by the previous phase / at entry), so the loop is
        // one straight run: per k-step four MFMAs on one weight fragment, the pixel fragments at column base + immediate
        int xc[PT][TAPS == 9 ? 3 : 1], xres16[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int pidx = pt * 16 + l15;
            const int pw = pidx >> p.th_shift, ph = pidx - (pw << p.th_shift);
            const int x0 = pw * colb + ph * RSM + k4 * 16 + kg * 64;
#pragma unroll
            for (int j = 0; j < (TAPS == 9 ? 3 : 1); ++j) xc[pt][j] = x0 + j * colb;
            xres16[pt] = abytes + pidx * RSR + k4 * 16 + kg * 64;
        }
        const unsigned char* wres = wbase + TAPS * CPT * 1024;
        // pixel fragments in blocks of NB k-steps, double-buffered: 16 ds_read_b128 per wave in flight while the previous block's 16 MFMAs
        // run (the loop is LDS-bound -- 64 pixels x K x 2 B = 295 KB per workgroup and phase -- and the LDS only reaches its rate with
        // >= 16 reads per wait, MI355X_MICROARCH.md; with 2-3 reads ahead the K loop + barrier took 2.9 k cycles whatever the MFMA count)
        constexpr int NB = RLDM_H16_NB, NBLK = (G + NB - 1) / NB;
        bf16x8 xb[2][NB * PT];
        auto issue = [&](const int blk) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int idx = blk * NB + j;
                if (idx >= G) continue;
                const int tap = idx / CPT, cs = idx % CPT;
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    xb[blk & 1][j * PT + pt] = *reinterpret_cast<const bf16x8*>(smem + xc[pt][TAPS == 9 ? tap / 3 : 0] +
                                                                                 (TAPS == 9 ? (tap % 3) * RSM : 0) + cs * (KG * 64));
            }
        };
        issue(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            if (blk + 1 < NBLK) issue(blk + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int idx = blk * NB + j;
                if (idx >= G) continue;
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    acc16[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[idx], xb[blk & 1][j * PT + pt], acc16[pt], 0, 0, 0);
                if (idx < RMAX) wr[idx] = *reinterpret_cast<const bf16x8*>(wres + (unsigned)(lane * 16 + max(min(idx, RPT - 1), 0) * 1024));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int idx = 0; idx < RMAX; ++idx) {
            if (idx < RPT) {
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    const bf16x8 xv = *reinterpret_cast<const bf16x8*>(smem + xres16[pt] + idx * (KG * 64));
                    acc16[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[idx], xv, acc16[pt], 0, 0, 0);
                }
            }
        }
    } else {
    // NGRP groups of TPG taps; step idx of a group = tap idx / CPT of the group, 16-channel group kg + (idx % CPT) * KG.
    // Per-lane LDS addresses: xa[mi][t] = pixel (mi, lane) at tap t of the current group, xn = the same for the next group.
    int xa[MI][TPG], xn[MI][TPG], xres[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int pidx = mi * 32 + l31;
        const int pw = pidx >> p.th_shift, ph = pidx - (pw << p.th_shift);
        const int x0 = pw * colb + ph * RSM + kh * 16 + kg * 32;
#pragma unroll
        for (int t = 0; t < TPG; ++t) {
            xa[mi][t] = TPG == 9 ? x0 + (t / 3) * colb + (t % 3) * RSM : x0 + t * RSM;   // group 0: all taps | taps (0, t)
            xn[mi][t] = TPG == 3 ? x0 + colb + t * RSM : x0 + RSM;         // group 1: taps (1, t) | tap (0, 1) (TPG == 9: none)
        }
        xres[mi] = abytes + pidx * RSR + kh * 16 + kg * 32;
    }
    auto x_read = [&](const int (&base)[MI][TPG], int idx, bf16x8 (&dst)[MI]) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
            dst[mi] = *reinterpret_cast<const bf16x8*>(smem + base[mi][idx / CPT] + (idx % CPT) * (KG * 32));
    };
    auto x_read_res = [&](int idx, bf16x8 (&dst)[MI]) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) dst[mi] = *reinterpret_cast<const bf16x8*>(smem + xres[mi] + idx * (KG * 32));
    };
    bf16x8 xr[PFX][MI];
#pragma unroll
    for (int j = 0; j < PFX; ++j) {
        x_read(xa, j, xr[j]);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned char* wnext = wbase + G * 1024;          // fragments of the next group
    int tj = 0;                                             // TPG == 1: beam offset of the current tap
#pragma unroll 1
    for (int g = 0; g < NGRP - 1; ++g) {
#pragma unroll
        for (int idx = 0; idx < G; ++idx) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                acc[idx % AS][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[idx], xr[idx % PFX][mi], acc[idx % AS][mi], 0, 0, 0);
            wr[idx] = w_load(wnext, idx);
            if (idx + PFX < G) x_read(xa, idx + PFX, xr[idx % PFX]);
            else x_read(xn, idx + PFX - G, xr[idx % PFX]);
            __builtin_amdgcn_sched_barrier(0);  // steps stay in program order: every wait then leaves G - 1 loads in flight
        }
        wnext += G * 1024;
        // next group's addresses
        if (TPG == 3) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int t = 0; t < TPG; ++t) { xa[mi][t] = xn[mi][t]; xn[mi][t] += colb; }
        } else {
            tj = tj == 2 ? 0 : tj + 1;                      // beam offset of the tap that just became current
            const int d = tj == 2 ? colb - 2 * RSM : RSM;   // ... and the step to the one after it
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) { xa[mi][0] = xn[mi][0]; xn[mi][0] += d; }
        }
    }
    // last main group: its refills are the residual steps (clamped: a k-group without that many re-reads a fragment)
    {
        const unsigned char* wres = wbase + TAPS * CPT * 1024;
#pragma unroll
        for (int idx = 0; idx < G; ++idx) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                acc[idx % AS][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[idx], xr[idx % PFX][mi], acc[idx % AS][mi], 0, 0, 0);
            if (idx < RMAX) wr[idx] = *reinterpret_cast<const bf16x8*>(wres + (unsigned)(lane * 16 + max(min(idx, RPT - 1), 0) * 1024));
            if (idx + PFX < G) x_read(xa, idx + PFX, xr[idx % PFX]);
            else x_read_res(idx + PFX - G, xr[idx % PFX]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int idx = 0; idx < RMAX; ++idx) {
        if (idx < RPT) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                acc[idx % AS][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[idx], xr[idx % PFX][mi], acc[idx % AS][mi], 0, 0, 0);
        }
        x_read_res(idx + PFX, xr[idx % PFX]);
    }
#pragma unroll
    for (int a = 1; a < AS; ++a)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][mi][r] += acc[a][mi][r];
    }
    RLDM_STAMP();
    // (multi-tile clusters: the next phase's weights are touched into the XCD's L2 now -- trunk_seam.h)
    TrunkWarm warm;
    if constexpr (TRUNK && (NWN > 1 || RLDM_TRUNK_WARM0)) trunk_warm_next(seam, tid, NT, warm);
    lds_barrier_s();                            // everyone is done with the input images: LDS is reused below
    RLDM_STAMP();

    // ---- epilogue, 64 pixels at a time: every k-group parks its fp32 partial half-tile in LDS as [k-group][pixel][channel];
    // then all 512 threads sum the k-groups for one (pixel, 8 channels) item each, round to bf16, store 16 bytes, and keep
    // (sum, sumsq) of the ROUNDED values for the next GroupNorm (same contract as conv_igemm.hip's epilogue) ------------
    unsigned char* sE = smem;
    constexpr int TRS = BN * 2 + 16;            // rounded half-tile [pixel][channel] bf16, for the statistics
    unsigned char* sT = sE + KG * HB * FRS;
    const int c8 = tid % NC8;
    const int chg = nt * BN + c8 * 8;
    // statistics: lane = channel pair (conflict-free 4-byte reads down the pixels), NT / (BN/2) pixel groups
    constexpr int NCP = BN / 2, NG = NT / NCP, PPG = HB / NG;
    const int cp = tid % NCP, pg = tid / NCP;
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int hp = 0; hp < NHALF; ++hp) {
        if (hp > 0) lds_barrier_s();            // the previous half-tile has been consumed
        if constexpr (H16) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)     // lane (pixel pt * 16 + l15) holds channels 4 k4 .. + 3
                *reinterpret_cast<float4*>(sE + (kg * HB + pt * 16 + l15) * FRS + k4 * 16) =
                    make_float4(acc16[pt][0], acc16[pt][1], acc16[pt][2], acc16[pt][3]);
        } else {
#pragma unroll
        for (int m2 = 0; m2 < (MI < 2 ? 1 : 2); ++m2) {
            const int mi = hp * 2 + m2, pl = m2 * 32 + l31;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int chl = wn * 32 + 8 * r4 + 4 * kh;
                *reinterpret_cast<float4*>(sE + (kg * HB + pl) * FRS + chl * 4) =
                    make_float4(acc[0][mi][r4 * 4 + 0], acc[0][mi][r4 * 4 + 1], acc[0][mi][r4 * 4 + 2], acc[0][mi][r4 * 4 + 3]);
            }
        }
        }
        lds_barrier_s();
        if (hp == 0) { RLDM_STAMP(); }
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            const int pl = tid / NC8 + q * (NT / NC8), pidx = hp * HB + pl;
            if (pl >= HB) break;
            const uint4 rq = resv[hp][q];
            float f[8] = {bf16lo(rq.x), bf16hi(rq.x), bf16lo(rq.y), bf16hi(rq.y),
                          bf16lo(rq.z), bf16hi(rq.z), bf16lo(rq.w), bf16hi(rq.w)};
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                const float4 v0 = *reinterpret_cast<const float4*>(sE + (g * HB + pl) * FRS + c8 * 32);
                const float4 v1 = *reinterpret_cast<const float4*>(sE + (g * HB + pl) * FRS + c8 * 32 + 16);
                f[0] += v0.x; f[1] += v0.y; f[2] += v0.z; f[3] += v0.w;
                f[4] += v1.x; f[5] += v1.y; f[6] += v1.z; f[7] += v1.w;
            }
            uint4 v;
            v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
            v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
            const int pw = pidx >> p.th_shift, ph = pidx - (pw << p.th_shift);
            const size_t pix = ((size_t)b * p.Wout + (w0 + pw)) * p.Hout + (h0 + ph);
            *reinterpret_cast<uint4*>(p.y + pix * p.y_ld + chg) = v;             // N % BN == 0 on this path
            *reinterpret_cast<uint4*>(sT + pl * TRS + c8 * 16) = v;
        }
        if (p.y_stats) {
            lds_barrier_s();
#pragma unroll
            for (int j = 0; j < PPG; ++j) {
                const uint32_t w2 = *reinterpret_cast<const uint32_t*>(sT + (pg * PPG + j) * TRS + cp * 4);
                const float a0 = bf16lo(w2), a1 = bf16hi(w2);
                s0 += a0; s1 += a1;
                q0 += a0 * a0; q1 += a1 * a1;
            }
        }
    }
    if (p.y_stats) {
        // the pixel groups of one wave fold by lane shuffles, the 8 waves through LDS
        float* sS = reinterpret_cast<float*>(sT + HB * TRS);                // [8 waves][2][BN]
        float* sC = sS + 16 * BN;                                           // [2][BN]: the tile's per-channel (sum, sumsq)
#pragma unroll
        for (int d = NCP; d < 64; d <<= 1) {
            s0 += __shfl_xor(s0, d); s1 += __shfl_xor(s1, d);
            q0 += __shfl_xor(q0, d); q1 += __shfl_xor(q1, d);
        }
        if (lane < NCP) {
            *reinterpret_cast<float2*>(sS + (wave * 2 + 0) * BN + cp * 2) = make_float2(s0, s1);
            *reinterpret_cast<float2*>(sS + (wave * 2 + 1) * BN + cp * 2) = make_float2(q0, q1);
        }
        lds_barrier_s();
        if (tid < 2 * BN) {
            const int kind = tid / BN, c = tid - kind * BN;
            float S = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) S += sS[(w * 2 + kind) * BN + c];
            reinterpret_cast<float*>(p.y_stats + ((size_t)b * tiles_img + mt) * p.N + nt * BN + c)[kind] = S;
            sC[tid] = S;
        }
        // ---- the tile IS the image (one tile per image): the statistics of its channels' groups are complete here, so the
        // GroupNorm (+ SiLU) of every consumer is applied now, once, to the rounded tile still in LDS, instead of by each of
        // the consumer's channel tiles on its way into LDS (statistics round trip + fold + 9 VALU instructions per element
        // in front of every consumer's K loop).  Same arithmetic as the consumer-side fold: per-channel fp32 sums of the
        // rounded values -> group sums in double -> a = gamma * rsq(var + eps), s = beta - mean * a -> a * x + s (-> SiLU) ----
        RLDM_STAMP();                           // output and statistics written
        if (NHALF == 1 && p.nviews > 0) {
            float* sAff = sC + 2 * BN;                                      // [view][2][BN]
            lds_barrier_s();
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                if (v >= p.nviews || tid / BN != v) continue;
                const NormView nv = p.nv[v];
                const int c = tid % BN, cpg = 1 << nv.cpg_shift, g0 = (c >> nv.cpg_shift) << nv.cpg_shift;
                double S = 0.0, SS = 0.0;
                for (int i = 0; i < cpg; ++i) {
                    S += (double)sC[g0 + i];
                    SS += (double)sC[BN + g0 + i];
                }
                const double inv_n = (double)nv.inv_n;
                const double mean = S * inv_n;
                double var = SS * inv_n - mean * mean;
                var = var < 0.0 ? 0.0 : var;
                const float a = nv_gamma * __builtin_amdgcn_rsqf((float)var + nv.eps);
                sAff[(v * 2 + 0) * BN + c] = a;
                sAff[(v * 2 + 1) * BN + c] = nv_beta - (float)mean * a;
            }
            lds_barrier_s();
            // item = (pixel, 4 channels): 8-byte pieces, 64-byte rows per pixel
            constexpr int NC4 = BN / 4, VPASS = (HB * NC4 + NT - 1) / NT;
#pragma unroll
            for (int q = 0; q < VPASS; ++q) {
                const int pl = tid / NC4 + q * (NT / NC4), c4 = tid % NC4;
                if (pl >= HB) break;
                const uint2 rq = *reinterpret_cast<const uint2*>(sT + pl * TRS + c4 * 8);
                const float x[4] = {bf16lo(rq.x), bf16hi(rq.x), bf16lo(rq.y), bf16hi(rq.y)};
                const int pw = pl >> p.th_shift, ph = pl - (pw << p.th_shift);
                const size_t pix = ((size_t)b * p.Wout + (w0 + pw)) * p.Hout + (h0 + ph);
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    if (v >= p.nviews) continue;
                    const NormView nv = p.nv[v];
                    const float4 av = *reinterpret_cast<const float4*>(sAff + (v * 2 + 0) * BN + c4 * 4);
                    const float4 sv = *reinterpret_cast<const float4*>(sAff + (v * 2 + 1) * BN + c4 * 4);
                    float f0 = x[0] * av.x + sv.x, f1 = x[1] * av.y + sv.y, f2 = x[2] * av.z + sv.z, f3 = x[3] * av.w + sv.w;
                    if (nv.silu) { f0 = silu_f(f0); f1 = silu_f(f1); f2 = silu_f(f2); f3 = silu_f(f3); }
                    uint2 o;
                    o.x = pack_bf16x2(f0, f1); o.y = pack_bf16x2(f2, f3);
                    *reinterpret_cast<uint2*>(nv.y + pix * nv.ld + nt * BN + c4 * 4) = o;
                }
            }
        }
    }
    RLDM_STAMP();
    if constexpr (TRUNK && (NWN > 1 || RLDM_TRUNK_WARM0)) trunk_warm_done(warm);
    if constexpr (TRUNK) trunk_arrive(seam, tid);
    if constexpr (TRUNK && PF > 0) {            // the next phase's first fragments: requested behind the arrive (its vmcnt(0) must not
                                                // wait for them), in flight during the seam and the next gather
        const unsigned nr = seam.next_rec;
        const int next_g = __builtin_amdgcn_readlane((int)nr, TW_G);
        const unsigned long long wp = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)nr, TW_WPK + 1) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)nr, TW_WPK);
        // (a GLOBAL-address-space pointer: built from the integer alone it would be a flat one, and flat loads also count on lgkmcnt)
        typedef __attribute__((address_space(1))) const unsigned char* global_bytes_t;
        const unsigned char* next_w = (const unsigned char*)(global_bytes_t)wp +
                                      ((size_t)seam.next_rank_kg * (unsigned)__builtin_amdgcn_readlane((int)nr, TW_NMINE)) * 1024;
#pragma unroll
        for (int j = 0; j < PF; ++j)
            if (j < next_g) wpf[j] = *reinterpret_cast<const bf16x8*>(next_w + (unsigned)(j * 1024 + lane * 16));
    }
#ifdef RLDM_ABLATE
    if (TRUNK && seam.ts && tid == 0) {
        RLDM_STAMP();
        for (int i = 0; i < 16; ++i) seam.ts[i] = i < tsn ? tsv[i] : 0ull;
    }
    if (!TRUNK && p.ts && blockIdx.x < 4 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0)
        for (int i = 0; i < 16; ++i) p.ts[blockIdx.x * 64 + i] = i < tsn ? tsv[i] : 0ull;
    {
        const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (!TRUNK && p.ts && tid == 0 && lin < 2048) {
            p.ts[256 + 2 * lin] = t_real0;
            p.ts[257 + 2 * lin] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
#undef RLDM_STAMP
}

}  // namespace rldm
