// Body of conv_stream_kernel (conv_stream.hip) as a device function, shared with the persistent trunk kernel (trunk.hip).
#pragma once
#include "kernels.h"
#include "trunk_seam.h"

namespace rldm {

#ifdef RLDM_ABLATE
#define RLDM_TDBG(p, bit) (((p).dbg & (bit)) != 0)
#else
#define RLDM_TDBG(p, bit) false
#endif

namespace {
__device__ __forceinline__ void lds_barrier_b() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
}  // namespace

// WM pixel parts (128 pixels each) x WN 32-channel tiles x KG k-groups = NW waves (8, or 4: round 4's half-size workgroups, two of
// which are resident on a CU -- independent tiles at different points of their prologue -> K loop -> epilogue chains, so that one's
// MFMAs run under the other's round trips; the per-wave K loop is the 8-wave instance's)
// TRUNK = true: a phase of the persistent launch (trunk.hip): arguments from a phase record, activations and statistics another
// workgroup of the image's cluster published are read past the L1, the phase ends by arriving on the cluster's counter.
// MI = 32-pixel fragments per wave (4; 2: round 4's 64-pixel x 128-channel x 2-k-group tile for the 128x8 level -- an 8 x 8 tile's halo is
// 100 positions against 180 for 16 x 8, it is normalised once for all 128 output channels instead of once per 64, and two k-groups exchange half
// the partial sums of four)
// STR = stride (1; 2: the UNet's Downsample2D, pad 1, on the 64-pixel tile: output (w, h) reads inputs (2w - 1 + i, 2h - 1 + j) -- a 17 x 17 halo)
// SUB = true: nearest x2 + 3x3 as FOUR 2x2 convolutions over the input (sub-pixel form).  Output pixel (2w + a, 2h + c) of the 3x3 over the
// up-sampled image reads only the inputs (w - 1 + a .. w + a) x (h - 1 + c .. h + c), each through the SUM of the taps that land on it
// (parity a = 0: {k[0]}, {k[1] + k[2]}; a = 1: {k[0] + k[1]}, {k[2]} along each axis; zero rows above / below the image stay zero, the wrap
// is the input's): 4 taps instead of 9 per output pixel.  The tiles are INPUT tiles; `nt_` = channel tile * 4 + parity; the weight stream of a
// (32-channel tile, parity) is [chunks][2 x 2 taps][4 k-steps] of summed weights (ConvLayer::get_subpixpacked)
// FH = true (round 4): tiles as tall as the image (8 x 16 on 16-beam images, 16 x 8 on 8-beam ones): the halo rows above and below the tile are the image's zero padding, ALWAYS --
// they are zeroed once per launch / phase and never staged, and the staged part (10 x 16 positions) is exactly five pieces per thread where the 16 x 8
// tile's 18 x 10 were six with 18 of them padding (the staging is VALU-issue bound: section 3.10)
template <int WM, int WN, bool TRUNK, int NW = 8, int MI = 4, int STR = 1, bool SUB = false, bool T4 = false, bool FH = false>
__device__ __forceinline__ void conv_stream_body(const ConvParams& p, const int nt_, const int mt, const int b, const TrunkSeam& seam) {
    static_assert(!SUB || (STR == 1 && NW / (WM * WN) == 1), "sub-pixel form: the one-k-group instances");
    static_assert(!FH || (STR == 1 && !SUB && NW == 4 && WM == 1), "full-height tiles: the 4-wave stride-1 instance");
    constexpr int TAPW = SUB ? 2 : 3;          // taps per row / rows of taps
    const int nt = SUB ? nt_ >> 2 : nt_;
    const int par_w = SUB ? (nt_ >> 1) & 1 : 0, par_h = SUB ? nt_ & 1 : 0;
    constexpr int NT = 64 * NW, CK = 64, KG = NW / (WM * WN);
    constexpr int BM = 32 * MI * WM, BN = 32 * WN;
    constexpr int RS = CK * 2 + 16;            // halo row stride (bytes): 9 16-byte slots
    constexpr int C8 = CK / 8;
    // 16-byte halo pieces per thread and chunk: 34 x 10 pixels (6); the 128-pixel instance 18 x 10 = 16 x 8 tiles, or 34 x 6 = 32 x 4
    // tiles for images of 4 beams (nuScenes' 128 x 4 level) (4)
    // (the 4-wave instances take 16 x 8 tiles only: 180 positions, 6 pieces per thread like the 256-pixel instance)
    // (T4: the sub-pixel instance on 32 x 4 tiles -- inputs of 4 beams: 34 x 6 = 204 positions, 7 pieces per thread)
    constexpr int HALO_PX = MI == 2 ? (STR == 2 ? 17 * 17 : (8 + 2) * 10) : (WM == 1 ? (NW == 4 ? (T4 ? (32 + 2) * (4 + 2) : (FH ? (8 + 2) * 16 : (16 + 2) * 10)) : (32 + 2) * (4 + 2)) : (BM / 8 + 2) * 10);
    constexpr int ACH = (HALO_PX * C8 + NT - 1) / NT;
    constexpr int SPT = 4 / KG;                // k-steps per tap of this wave
    constexpr int ROW = TAPW * SPT;            // ... per row of taps
    constexpr int CST = TAPW * TAPW * SPT;     // ... per chunk
    // weight fragments in flight per wave (ring): a chunk (9) | a row of taps (12 | 6); MI == 2: a chunk (18) -- a k-step is 64 cycles of MFMAs
    constexpr int G = (KG == 4 || MI == 2) ? CST : ROW;
    constexpr int PFX = KG == 1 ? 2 : 3;       // pixel fragments read ahead; divides CST
    constexpr int ERS = BN * 2 + 16, NC8 = BN / 8;
    static_assert(WM * WN * KG == NW && (NW == 4 || NW == 8) && CST % PFX == 0 && G <= 18 && (MI == 4 || (MI == 2 && WM == 1 && NW == 8)), "wave grid");
    constexpr int CPT = 512 / NT;              // channels per thread in the GroupNorm fold (Cin <= 512)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tid_ = threadIdx.x;
    if constexpr (TRUNK) asm volatile("" : "+v"(tid_));          // (opaque per phase: conv_small_body.h)
    const int tid = tid_, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / (WM * WN);
    const int wm = (wave % (WM * WN)) / WN, wn = wave % WN;
    const int grp = wave >> 2;                 // the two waves of a SIMD are in different halves of the workgroup (4 waves: one half)
    const int kh = lane >> 5, l31 = lane & 31;
#ifdef RLDM_ABLATE
    unsigned long long tsv[12];
    int tsn = 0;
#define RLDM_STAMP() if (tsn < 12) tsv[tsn++] = __builtin_amdgcn_s_memtime()
#else
#define RLDM_STAMP()
#endif
    RLDM_STAMP();
#ifdef RLDM_ABLATE
    const unsigned long long t_real0 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- which tile: grid = (channel tiles, pixel tiles of an image, images) -------------------------------------------
    const int tiles_h = p.tiles_h, tiles_img = p.tiles_img;       // tiles_h is a power of two
    const int tw = mt >> (31 - __builtin_clz(tiles_h)), th = mt & (tiles_h - 1);
    const int w0 = tw * p.TW, h0 = th * p.TH;

    // the sampler's step index selects the time-embedding row: requested first, used after everything else is in flight
    const float* const temb_tab = TRUNK ? seam.temb : p.temb;
    const int* const step_ptr = TRUNK ? seam.step_ptr : p.step_ptr;
    const int temb_rps = TRUNK ? seam.temb_rows_per_step : p.temb_rows_per_step;
    const int temb_ps = TRUNK ? seam.temb_per_sample : p.temb_per_sample;
    const int temb_ld = TRUNK ? seam.temb_ld : p.temb_ld;
    const int temb_step = (temb_tab && step_ptr && tid < BN) ? load_step_vector(step_ptr) : 0;

    const int Cin = p.C0 + p.C1;
    const int NCC = Cin / CK;                  // main-phase chunks: 9 taps x 4 k-steps
    const int NCBw = (p.R0 + p.R1) / CK;       // residual-phase chunks: centre tap, 4 k-steps, raw input
    const int NCB = RLDM_EXP_NORES ? 0 : NCBw;
    const int NCT = NCC + NCB;
    const int THv = (p.TH - 1) * STR + 3, TWv = (p.TW - 1) * STR + 3;
    static_assert(STR == 1 || (STR == 2 && MI == 2), "stride 2: the 64-pixel instance only");
    const int colb = p.colb;
    const int abytes = TWv * colb;
    const int Wv = p.Win * p.up, Hv = p.Hin * p.up;
    const int upshift = p.up - 1;

    unsigned char* sA = smem;                                  // 2 * abytes
    float* sGa = reinterpret_cast<float*>(sA + 2 * abytes);    // Cin
    float* sGs = sGa + Cin;
    float* sBias = sGs + Cin;                                  // BN

    // ---- this wave's weight stream (channel tile WN*nt + wn, k-group kg): [NCC][9 taps][SPT k-steps] then [NCB][SPT], 1 KiB each
    const int nsteps = NCC * CST + NCBw * SPT;
    const unsigned char* wptr = reinterpret_cast<const unsigned char*>(p.wpk) +
                                (size_t)(SUB ? (nt * WN + wn) * 4 + (par_w * 2 + par_h) : (nt * WN + wn) * KG + kg) * nsteps * 1024;
    const unsigned woff = lane * 16 + 4096;     // lane offset: immediates of +-4 KiB around it reach 8 fragments
    auto w_load = [&](const unsigned char* base, int idx) __attribute__((always_inline)) {      // fragment idx in [0, 16)
        return *reinterpret_cast<const bf16x8*>(base + (idx / 8) * 8192 + woff + ((idx % 8) * 1024 - 4096));
    };
    RLDM_STAMP();
    // ---- halo staging: thread-constant source pixel of each of its ACH 16-byte pieces ---------------------------------
    const int THs = FH ? p.TH : THv;            // staged rows per halo column (FH: the tile's own rows -- p.magic_thv divides by THs)
    const int atotal = TWv * THs * C8;
    int apix[ACH];
    const int my_c8 = (tid % C8) * 8;
    // halo pieces of a chunk in registers between its request and its store: one set, or two for the short-chunk instance (KG == 4: a
    // chunk is 9 k-steps per wave, ~2.5 k cycles -- less than the pieces' round trip, so chunk cs + 2 is requested during chunk cs and
    // stored during chunk cs + 1; round 4: the 128x8 convs' K loops ran 5.5 k cycles per chunk for 2.3 k of matrix-pipe time, most of the
    // rest was store_next waiting for loads issued ~400 cycles earlier)
    constexpr bool PF2 = (KG == 4 && RLDM_STREAM_PF2) || (MI == 2 && RLDM_STREAM_PF2_MI2);     // (MI == 2: two pieces per thread, 8 registers)
    uint4 areg[ACH];
    uint4 areg2[ACH];                          // (second set: PF2 only -- never touched otherwise)
    // (the sets are selected by a compile-time tag inside the lambdas: passing an array of uint4 by reference sends it to scratch)
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    const bool gn = p.st0 != nullptr;
    const bf16_t* const gx0 = p.x0;             // (locals: selecting between fields of `p` by address would copy it to scratch)
    const bf16_t* const gx1 = p.x1;
    const bf16_t* const gr0 = p.r0;
    const bf16_t* const gr1 = p.r1;
    const int nC0 = p.C0, nC1 = p.C1, nR0 = p.R0, nR1 = p.R1;
    auto load_a = [&](int cs, auto W) __attribute__((always_inline)) {            // chunk cs of the sequence main, residual
        const bool main_phase = cs < NCC;
        const int c = (main_phase ? cs : cs - NCC) * CK + my_c8;
        const int split = main_phase ? nC0 : nR0;
        const bool first = c < split;
        const bf16_t* t0 = main_phase ? gx0 : gr0;
        const bf16_t* t1 = main_phase ? gx1 : gr1;
        const bf16_t* base = first ? t0 + c : t1 + (c - split);
        const int ld = first ? split : (main_phase ? nC1 : nR1);
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int pix = apix[i] < 0 ? 0 : apix[i];
            const uint4 v = ld_act16<TRUNK>(base + (size_t)pix * ld);
            if constexpr (decltype(W)::value == 0) areg[i] = v; else areg2[i] = v;
        }
    };
    auto store_a = [&](int cs, auto W) __attribute__((always_inline)) {           // GroupNorm + SiLU (main phase) -> LDS
        unsigned char* dstbuf = sA + (cs & 1) * abytes;
        const bool anorm = gn && cs < NCC && !RLDM_EXP_NONORM;
        float4 ga0, ga1, gs0, gs1;
        if (anorm) {
            const int c = cs * CK + my_c8;
            ga0 = *reinterpret_cast<const float4*>(sGa + c);
            ga1 = *reinterpret_cast<const float4*>(sGa + c + 4);
            gs0 = *reinterpret_cast<const float4*>(sGs + c);
            gs1 = *reinterpret_cast<const float4*>(sGs + c + 4);
        }
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            uint4 v;
            if constexpr (decltype(W)::value == 0) v = areg[i]; else v = areg2[i];
            if (apix[i] < 0) {
                v = make_uint4(0u, 0u, 0u, 0u);
            } else if (anorm) {
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
            const int q = tid + i * NT;
            const int slot = q / C8, c8 = q - slot * C8;
            const int vwl = (slot * p.magic_thv) >> 20, vhl = slot - vwl * THs + (FH ? 1 : 0);
            if (q < atotal) *reinterpret_cast<uint4*>(dstbuf + vwl * colb + vhl * RS + c8 * 16) = v;
        }
    };
    // A RESIDUAL chunk (the raw block input under the centre tap: identity, or the 1x1 shortcut) is read at the tile's own pixels only:
    // a thread-linear mapping of the BM x 8 pieces (4 | 2 per thread instead of the halo's 6 | 4, a quarter fewer bytes), addresses
    // recomputed per chunk (shifts and adds), the pieces parked in the halo registers.  Round 3: the residual phase of the
    // full-resolution convs is a burst of loads with 16 MFMAs per wave between two of them -- removing it altogether measured +4.0 %
    // end to end (an upper bound), the halo-shaped fetch with its unused ring was the most avoidable part of it.
    constexpr int RCH = (BM * C8 + NT - 1) / NT;
    static_assert(RCH <= ACH && (BM * C8) % NT == 0, "residual pieces fit the halo registers");
    constexpr int RSTEP = NT / C8;                                        // tile pixels between a thread's pieces (a whole number of columns)
    auto load_r = [&](int cs, uint4* rr) __attribute__((always_inline)) {
        const int cb = (cs - NCC) * CK;                                   // (uniform: the chunk lies in ONE of the two tensors)
        const bool first = cb < nR0;
        const bf16_t* base = (first ? gr0 + cb : gr1 + (cb - nR0)) + my_c8;
        const int ld = first ? nR0 : nR1;
        const int pidx = tid / C8, pw = pidx >> p.th_shift, ph = pidx - (pw << p.th_shift);
        const bf16_t* src = base + (size_t)((b * p.Wout + w0 + pw) * p.Hout + h0 + ph) * ld;
        const size_t step = (size_t)((RSTEP >> p.th_shift) * p.Hout) * ld;
#pragma unroll
        for (int i = 0; i < RCH; ++i) rr[i] = ld_act16<TRUNK>(src + i * step);
    };
    auto store_r = [&](int cs, const uint4* rr) __attribute__((always_inline)) {
        const int pidx = tid / C8, pw = pidx >> p.th_shift, ph = pidx - (pw << p.th_shift);
        unsigned char* dst = sA + (cs & 1) * abytes + (pw + 1) * colb + (ph + 1) * RS + (tid % C8) * 16;
        const int step = (RSTEP >> p.th_shift) * colb;
#pragma unroll
        for (int i = 0; i < RCH; ++i) *reinterpret_cast<uint4*>(dst + i * step) = rr[i];
    };
    auto load_next = [&](int cs, auto W) __attribute__((always_inline)) {
        if (cs < NCC) load_a(cs, W);
        else if constexpr (decltype(W)::value == 0) load_r(cs, areg);
        else load_r(cs, areg2);
    };
    auto store_next = [&](int cs, auto W) __attribute__((always_inline)) {
        if (cs < NCC) store_a(cs, W);
        else if constexpr (decltype(W)::value == 0) store_r(cs, areg);
        else store_r(cs, areg2);
    };
    RLDM_STAMP();
    if constexpr (TRUNK) trunk_wait(seam, tid);     // (everything above is independent of the previous phase)
    // ---- GroupNorm: the statistics partials of channel `tid`, gamma and beta are requested first, then the first halo
    // chunk; the fold runs while they are all in flight.  Every channel's thread folds its own group (no serial phase).
    double gS[CPT], gSS[CPT];
    float g_gamma[CPT], g_beta[CPT];
#pragma unroll
    for (int sl = 0; sl < CPT; ++sl) {
        gS[sl] = 0.0; gSS[sl] = 0.0; g_gamma[sl] = 0.f; g_beta[sl] = 0.f;
        const int ch = tid + sl * NT;
        if (gn && ch < Cin) {
        const float2* const gs0p = p.st0;
        const float2* const gs1p = p.st1;
        const int nP0 = p.P0, nP1 = p.P1;
        const bool first = ch < nC0;
        const int c = first ? ch : ch - nC0;
        const int C = first ? nC0 : nC1;
        const int P = RLDM_EXP_NOSTATS ? 0 : (first ? nP0 : nP1);
        if (RLDM_EXP_NOSTATS) gSS[sl] = (double)(1.0f / p.gn_inv_n) / (Cin / p.gn_groups);   // (mean 0, variance 1: finite garbage)
        const float2* src = (first ? gs0p : gs1p) + (size_t)b * P * C + c;
        g_gamma[sl] = p.gn_gamma[ch];
        g_beta[sl] = p.gn_beta[ch];
        int q = 0;
        if constexpr (NW == 4) {                // (128-pixel tiles: 32 partials per image at the 256 x 16 level -- one round trip)
            for (; q + 32 <= P; q += 32) {
                float2 v[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) v[j] = ld_act8<TRUNK>(src + (size_t)(q + j) * C);
#pragma unroll
                for (int j = 0; j < 32; ++j) { gS[sl] += (double)v[j].x; gSS[sl] += (double)v[j].y; }
            }
        }
        for (; q + 16 <= P; q += 16) {          // 16 partials per round trip (P = pixel tiles per image of the producer)
            float2 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = ld_act8<TRUNK>(src + (size_t)(q + j) * C);
#pragma unroll
            for (int j = 0; j < 16; ++j) { gS[sl] += (double)v[j].x; gSS[sl] += (double)v[j].y; }
        }
        for (; q + 4 <= P; q += 4) {
            float2 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ld_act8<TRUNK>(src + (size_t)(q + j) * C);
#pragma unroll
            for (int j = 0; j < 4; ++j) { gS[sl] += (double)v[j].x; gSS[sl] += (double)v[j].y; }
        }
        for (; q < P; ++q) {
            const float2 v = ld_act8<TRUNK>(src + (size_t)q * C);
            gS[sl] += (double)v.x;
            gSS[sl] += (double)v.y;
        }
        }
    }
    float bias_v = 0.f;
    if (tid < BN) bias_v = p.bias[nt * BN + tid];
    // Request order = the order the data is needed in (s_waitcnt vmcnt counts in order): statistics partials (the fold), the first
    // halo chunk (GroupNorm + SiLU of chunk 0), the weight ring (the K loop).  The address arithmetic of the halo pieces (integer
    // multiplies) runs while the partials are in flight.  Round 3: the halo chunk used to be requested BEHIND the ring -- its wait then
    // also covered the ring's 12 KB per wave (96 KB per CU through a 64 B/clk L1), ~2 k cycles of every launch / phase.
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        const int q = tid + i * NT;
        const int slot = q / C8;
        const int vwl = (slot * p.magic_thv) >> 20, vhl = slot - vwl * THs + (FH ? 1 : 0);
        const int vh = h0 * STR - 1 + vhl;
        int vw = w0 * STR - 1 + vwl;
        vw = vw < 0 ? vw + Wv : (vw >= Wv ? vw - Wv : vw);
        const bool ok = q < atotal && vh >= 0 && vh < Hv;
        apix[i] = ok ? ((b * p.Win + (vw >> upshift)) * p.Hin + (vh >> upshift)) : -1;
    }
    if (NCT > 0) load_a(0, Set0());
    RLDM_STAMP();
    bf16x8 wr[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        wr[j] = w_load(wptr, j);
        __builtin_amdgcn_sched_barrier(0);      // issued here and in this order: the counted waits below rely on it
    }
    wptr += G * 1024;                           // -> the fragments the first row of taps refills
    if (temb_tab && tid < BN)
        bias_v += temb_tab[(size_t)(temb_step * temb_rps + (temb_ps ? b : 0)) * temb_ld + nt * BN + tid];
    if (gn) {
        double* sD = reinterpret_cast<double*>(sA);             // scratch: [2][Cin] doubles (the halo is not written yet)
        const int cpg = Cin / p.gn_groups;
#pragma unroll
        for (int sl = 0; sl < CPT; ++sl) {
            const int ch = tid + sl * NT;
            if (ch < Cin) {
                sD[ch] = gS[sl];
                sD[Cin + ch] = gSS[sl];
            }
        }
        __syncthreads();
        float ga[CPT], gs[CPT];                 // Cin <= CPT * NT
#pragma unroll
        for (int sl = 0; sl < CPT; ++sl) {
            const int ch = tid + sl * NT;
            ga[sl] = 0.f; gs[sl] = 0.f;
            if (ch < Cin) {
                const int g0 = ((ch * p.magic_cpg) >> 20) * cpg;
                double S = 0.0, SS = 0.0;
                for (int i = 0; i < cpg; ++i) {
                    S += sD[g0 + i];
                    SS += sD[Cin + g0 + i];
                }
                const double inv_n = (double)p.gn_inv_n;
                const double mean = S * inv_n;
                double var = SS * inv_n - mean * mean;
                var = var < 0.0 ? 0.0 : var;
                ga[sl] = g_gamma[sl] * __builtin_amdgcn_rsqf((float)var + p.gn_eps);
                gs[sl] = g_beta[sl] - (float)mean * ga[sl];
            }
        }
#pragma unroll
        for (int sl = 0; sl < CPT; ++sl) {
            const int ch = tid + sl * NT;
            if (ch < Cin) {
                sGa[ch] = ga[sl];               // (sGa / sGs sit behind both halo buffers: disjoint from the scratch)
                sGs[ch] = gs[sl];
            }
        }
        __syncthreads();                        // affine visible; sD fully consumed before the halo is written
    }
    RLDM_STAMP();
    if (tid < BN) sBias[tid] = bias_v;
    if constexpr (FH) {
        // the zero rows above and below the image, both buffers: TWv columns x 2 rows x 2 buffers x 9 slots of 16 bytes (never written again;
        // the k-group-less epilogue's staging overwrites them only after the last chunk)
        const int nz = TWv * 2 * 2 * (RS / 16);
        for (int q = tid; q < nz; q += NT) {
            const int sl = q % (RS / 16), r = q / (RS / 16);
            const int col = r >> 2, which = r & 3;                       // (buffer, top / bottom)
            *reinterpret_cast<uint4*>(sA + (which >> 1) * abytes + col * colb + ((which & 1) ? (THv - 1) * RS : 0) + sl * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    if (NCT > 0) store_a(0, Set0());
    if constexpr (PF2) { if (1 < NCT) load_next(1, Set1()); }   // (requested a chunk ahead from the start)
    RLDM_STAMP();

    // ---- per-lane LDS offsets of the pixel fragments; accumulators start at bias + temb ---------------------------------
    int xoff[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int pidx = wm * (MI * 32) + mi * 32 + l31;
        const int pw = pidx >> p.th_shift, ph = pidx - (pw << p.th_shift);
        xoff[mi] = (pw * STR + par_w) * colb + (ph * STR + par_h) * RS + kh * 16 + kg * (SPT * 32);
    }
    lds_barrier_b();                            // sBias and halo chunk 0 are written
    f32x16 acc[MI];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        float4 bv = *reinterpret_cast<const float4*>(sBias + wn * 32 + 8 * r4 + 4 * kh);
        if (kg != 0) bv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            acc[mi][r4 * 4 + 0] = bv.x; acc[mi][r4 * 4 + 1] = bv.y;
            acc[mi][r4 * 4 + 2] = bv.z; acc[mi][r4 * 4 + 3] = bv.w;
        }
    }

    // ---- K loop ----------------------------------------------------------------------------------------------------------
    // chunk step c = ROW * ti + SPT * tj + ks; its fragment sits in ring slot c % G and is refilled with the fragment G
    // steps ahead (next row / next chunk / residual phase: the stream is linear) right after its MFMAs
    bf16x8 xr[PFX][MI];
    // cur / nxt: per-lane LDS addresses of the pixel at tap (ti, 0) / (ti + 1, 0); r = step within the row (may run into the next)
    auto x_read = [&](const int (&cur)[MI], const int (&nxt)[MI], int r, bf16x8 (&dst)[MI]) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
            dst[mi] = *reinterpret_cast<const bf16x8*>(smem + (r < ROW ? cur[mi] : nxt[mi]) + ((r % ROW) / SPT) * RS + (r % SPT) * 32);
    };
    auto tap_row = [&](const int (&cur)[MI], const int (&nxt)[MI], int ti) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < ROW; ++j) {
            const int c = ti * ROW + j, slot = c % G;
#if RLDM_STREAM_ILV
            // (round 4) each pixel fragment of step c + PFX is requested right behind the MFMA that consumed its register, not behind
            // the step's last MFMA: ~100 cycles more lead per read.  A wave ALONE on its SIMD (the 4-wave instances while the other
            // workgroup of the CU is outside its K loop) ran 236 cycles per k-step against 128 of matrix-pipe time -- waiting on LDS.
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[slot], xr[c % PFX][mi], acc[mi], 0, 0, 0);
                if (c + PFX < CST && !RLDM_TDBG(p, 65536)) {
                    const int r = j + PFX;
                    xr[c % PFX][mi] = *reinterpret_cast<const bf16x8*>(smem + (r < ROW ? cur[mi] : nxt[mi]) + ((r % ROW) / SPT) * RS + (r % SPT) * 32);
                }
            }
            if (!RLDM_TDBG(p, 32768)) wr[slot] = w_load(wptr, slot);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (c + PFX < CST) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#else
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[slot], xr[c % PFX][mi], acc[mi], 0, 0, 0);
            wr[slot] = w_load(wptr, slot);
            if (c + PFX < CST) x_read(cur, nxt, j + PFX, xr[c % PFX]);   // (no read-ahead across the chunk's barrier)
#endif
            __builtin_amdgcn_sched_barrier(0);  // steps stay in program order: every wait then leaves G - 1 loads in flight
        }
        if ((ti * ROW + ROW) % G == 0) wptr += G * 1024;
    };
    static_assert(PFX <= ROW, "the read-ahead reaches at most into the next row of taps");
    RLDM_STAMP();
    // (round-4 experiments on how the two workgroups of a CU share its matrix pipes; p.exp bits, stand-alone launches)
    //   1 static priority for the first-dispatched half of the grid during its K loop; 2 a per-CU lock around the K loop;
    //   4 priority by progress
    int* cu_lock = nullptr;
    if constexpr (NW == 4 && !TRUNK) {
        if (p.exp >> 4) {                       // initial skew: the second workgroup of every CU starts (p.exp >> 4) x 4096 cycles late
            const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            if (lin >= 256 && lin < 512)
                for (int i = 0; i < 4 * (p.exp >> 4); ++i) __builtin_amdgcn_s_sleep(16);
        }
        if (p.exp & 1) {
            const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            if (2 * lin < (int)(gridDim.x * gridDim.y * gridDim.z)) __builtin_amdgcn_s_setprio(3);
        }
        if (p.exp & 4) __builtin_amdgcn_s_setprio(1);
        if ((p.exp & 2) && p.cu_lock) {
            unsigned xcc, hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 8, 8)" : "=s"(hw));
            cu_lock = p.cu_lock + ((xcc << 8) | hw);
            if (tid == 0) {
                for (int tries = 0; tries < 200000; ++tries) {          // (bounded: a lost lock only costs the overlap)
                    int expect = 0;
                    if (__hip_atomic_compare_exchange_strong(cu_lock, &expect, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            __syncthreads();
        }
    }
    // one main chunk: `held` carries chunk cs + 1's pieces if they were requested earlier (PF2), `fresh` receives this iteration's request
    auto main_chunk = [&](int cs, auto held, auto fresh) __attribute__((always_inline)) {
        if constexpr (NW == 4 && !TRUNK) { if ((p.exp & 4) && 2 * cs >= NCC) __builtin_amdgcn_s_setprio(2); }
        if constexpr (PF2) {
            // chunk cs + 2 (main chunks and the FIRST residual chunk travel through these registers); chunk cs + 1 is in `held`
            if (cs + 2 < NCT && cs + 2 <= NCC && !RLDM_TDBG(p, 16384)) load_next(cs + 2, fresh);
        } else {
            if (cs + 1 < NCT && !RLDM_TDBG(p, 16384)) load_next(cs + 1, held);   // next chunk (main or first residual): requested now, written below
        }
        int cur[MI], nxt[MI];
        const int boff = (cs & 1) * abytes;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) { cur[mi] = xoff[mi] + boff; nxt[mi] = cur[mi] + colb; }
#pragma unroll
        for (int j = 0; j < PFX; ++j) {
            x_read(cur, nxt, j, xr[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        tap_row(cur, nxt, 0);
        if (grp == 0 && cs + 1 < NCT && !RLDM_TDBG(p, 8192)) store_next(cs + 1, held);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) { cur[mi] = nxt[mi]; nxt[mi] += colb; }
        tap_row(cur, nxt, 1);
        if (grp == 1 && cs + 1 < NCT && !RLDM_TDBG(p, 8192)) store_next(cs + 1, held);
        if constexpr (TAPW == 3) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) cur[mi] = nxt[mi];
            tap_row(cur, nxt, 2);
        }
        lds_barrier_b();                        // chunk cs consumed by everyone, chunk cs + 1 written by everyone
    };
    if constexpr (PF2) {
        for (int cs = 0; cs < NCC; cs += 2) {   // (two chunks per trip: the register sets swap roles with static indices)
            main_chunk(cs, Set1(), Set0());
            if (cs + 1 < NCC) main_chunk(cs + 1, Set0(), Set1());
        }
    } else {
        for (int cs = 0; cs < NCC; ++cs) main_chunk(cs, Set0(), Set1());
    }
    // residual phase: centre tap of the raw block input, SPT k-steps per chunk; ring slots continue (CST % G == 0).  A chunk is 16
    // MFMAs per wave -- far less than a round trip to the L2 -- so RD chunks are kept in flight in registers (the main loop's
    // read-ahead registers are dead here): chunk k lands in set k % RD, requested RD iterations before it is written to LDS.
    constexpr int RCR = G / SPT;                // residual chunks per ring revolution
    constexpr int RD = RLDM_RES_DEPTH, RUN = RCR * RD;
    uint4 rreg[RD][RCH];
#pragma unroll
    for (int k = 1; k < RD; ++k)
        if (k < NCB) load_r(NCC + k, rreg[k % RD]);             // (chunk 0 came through the halo registers during the last main chunk)
    for (int rc0 = 0; rc0 < NCB; rc0 += RUN) {
#pragma unroll
        for (int r = 0; r < RUN; ++r) {
            const int rc = rc0 + r;
            if (rc < NCB) {
                const int cs = NCC + rc;
                if (rc + RD < NCB) load_r(cs + RD, rreg[r % RD]);
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
                        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[(r % RCR) * SPT + ks], xf[mi], acc[mi], 0, 0, 0);
                    wr[(r % RCR) * SPT + ks] = w_load(wptr, (r % RCR) * SPT + ks);
                }
                if (rc + 1 < NCB) store_r(cs + 1, rreg[(r + 1) % RD]);
                lds_barrier_b();
            }
            if (r % RCR == RCR - 1) wptr += G * 1024;
        }
    }
    RLDM_STAMP();
    if constexpr (NW == 4 && !TRUNK) {
        if (p.exp & 1) __builtin_amdgcn_s_setprio(0);
        if (p.exp & 4) __builtin_amdgcn_s_setprio(3);
        if (cu_lock && tid == 0) __hip_atomic_store(cu_lock, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    if constexpr (KG > 1) {
    // ---- epilogue of the k-group instance (conv_small.hip's), 64 pixels at a time: fp32 partials [k-group][pixel][channel] in
    // LDS -> all threads sum the k-groups of one (pixel, 8 channels) item each, round, store 16 bytes, statistics ----------
    constexpr int HB = 64, NHALF = BM / HB, FRS = BN * 4 + 16, TRS = BN * 2 + 16;
    constexpr int NPASS = (HB * NC8 + NT - 1) / NT;
    unsigned char* sE = smem;
    unsigned char* sT = sE + KG * HB * FRS;
    const int c8 = tid % NC8;
    const int chg = nt * BN + c8 * 8;
    constexpr int NCP = BN / 2, NG = NT / NCP, PPG = HB / NG;
    const int cp = tid % NCP, pg = tid / NCP;
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int hp = 0; hp < NHALF; ++hp) {
        if (hp > 0) lds_barrier_b();            // the previous half-tile has been consumed
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
            const int mi = (hp % 2) * 2 + m2, pl = m2 * 32 + l31;       // (the waves of pixel part hp / 2 hold this pass's pixels)
            if (WM > 1 && wm != hp / 2) continue;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int chl = wn * 32 + 8 * r4 + 4 * kh;
                *reinterpret_cast<float4*>(sE + (kg * HB + pl) * FRS + chl * 4) =
                    make_float4(acc[mi][r4 * 4 + 0], acc[mi][r4 * 4 + 1], acc[mi][r4 * 4 + 2], acc[mi][r4 * 4 + 3]);
            }
        }
        lds_barrier_b();
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            const int pl = tid / NC8 + q * (NT / NC8), pidx = hp * HB + pl;
            if (pl >= HB) break;
            float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
            *reinterpret_cast<uint4*>(p.y + pix * p.y_ld + chg) = v;
            *reinterpret_cast<uint4*>(sT + pl * TRS + c8 * 16) = v;
        }
        if (p.y_stats) {
            lds_barrier_b();
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
        float* sS = reinterpret_cast<float*>(sT + HB * TRS);                // [NW waves][2][BN]
#pragma unroll
        for (int d = NCP; d < 64; d <<= 1) {
            s0 += __shfl_xor(s0, d); s1 += __shfl_xor(s1, d);
            q0 += __shfl_xor(q0, d); q1 += __shfl_xor(q1, d);
        }
        if (lane < NCP) {
            *reinterpret_cast<float2*>(sS + (wave * 2 + 0) * BN + cp * 2) = make_float2(s0, s1);
            *reinterpret_cast<float2*>(sS + (wave * 2 + 1) * BN + cp * 2) = make_float2(q0, q1);
        }
        lds_barrier_b();
        if (tid < 2 * BN) {
            const int kind = tid / BN, c = tid - kind * BN;
            float S = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) S += sS[(w * 2 + kind) * BN + c];
            reinterpret_cast<float*>(p.y_stats + ((size_t)b * tiles_img + mt) * p.N + nt * BN + c)[kind] = S;
        }
    }
    } else {
    // ---- epilogue (conv_igemm.hip's): bf16 -> LDS [pixel][channel] -> 16-byte coalesced stores + statistics -----------------
    unsigned char* sE = smem;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int pidx = wm * (MI * 32) + mi * 32 + l31;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int chl = wn * 32 + 8 * r4 + 4 * kh;
            uint2 o;
            o.x = pack_bf16x2(acc[mi][r4 * 4 + 0], acc[mi][r4 * 4 + 1]);
            o.y = pack_bf16x2(acc[mi][r4 * 4 + 2], acc[mi][r4 * 4 + 3]);
            *reinterpret_cast<uint2*>(sE + pidx * ERS + chl * 2) = o;
        }
    }
    lds_barrier_b();
    RLDM_STAMP();
    // thread (g = tid / 16, c8 = tid % 16) stores 16 bytes of pixels g, g + 32, ... (4 halo columns apart: a constant
    // address step; 4 waves: g + 16, ..., 2 columns), then the statistics of the ROUNDED tile: lane = channel pair, 8 pixel groups, LDS fold over the waves
    const int c8 = tid % NC8;
    const int chg = nt * BN + c8 * 8;
    {
        const int g = tid / NC8;                                            // pixel 0..31 of the pass: (pw, ph) = (g >> 3, g & 7) on tiles of 8 rows
        const int us = SUB ? 2 : 1;                                          // (sub-pixel form: input pixel (w, h) -> output (2w + parity, 2h + parity))
        const int ts = T4 ? 2 : (FH ? p.th_shift : 3);                       // (... on tiles of 4 rows; full-height tiles of 16 or 8 rows)
        bf16_t* yp = p.y + (((size_t)b * p.Wout + (w0 + (g >> ts)) * us + par_w) * p.Hout + (h0 + (g & ((1 << ts) - 1))) * us + par_h) * p.y_ld + chg;
        const size_t ystep = (size_t)((NT / NC8) >> ts) * us * p.Hout * p.y_ld;
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
        float* sS = reinterpret_cast<float*>(sE + BM * ERS);                // [NW waves][2][BN]
        *reinterpret_cast<float2*>(sS + (wave * 2 + 0) * BN + cp * 2) = make_float2(s0, s1);
        *reinterpret_cast<float2*>(sS + (wave * 2 + 1) * BN + cp * 2) = make_float2(q0, q1);
        lds_barrier_b();
        if (tid < 2 * BN) {
            const int kind = tid / BN, c = tid - kind * BN;
            float S = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) S += sS[(w * 2 + kind) * BN + c];
            // (sub-pixel form: one partial per (tile, parity))
            const size_t part = SUB ? ((size_t)b * tiles_img + mt) * 4 + (par_w * 2 + par_h) : (size_t)b * tiles_img + mt;
            reinterpret_cast<float*>(p.y_stats + part * p.N + nt * BN + c)[kind] = S;
        }
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
        // stamps of four workgroups: dispatch ids 0, 1 (first round) and total / 2, total / 2 + 1 (with two workgroups per CU: the
        // younger half); every workgroup's [start, end] on the shared 100 MHz counter, its CU (XCC id, HW_ID's SE / SH / CU) in the
        // top 16 bits of `end`
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
