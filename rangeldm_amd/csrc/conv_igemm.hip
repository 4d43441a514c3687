// Circular implicit-GEMM convolution for gfx950 (MI355X), bf16 in / fp32 accumulate on v_mfma_f32_32x32x16_bf16.
//
// Replaces the reference's `F.pad(circular W) + F.pad(zero H) + F.conv2d` (ldm/utils.py:46-49, dup
// vae/sgm/modules/diffusionmodules/model.py:99-102), the stride-2 down-samplers (ldm/utils.py:107-116, model.py:164-172),
// nearest-x2 + conv up-samplers (model.py:120-125), the 1x1 shortcuts and the attention Linear layers.  One launch is a
// whole "GroupNorm -> SiLU -> conv (+ time embedding) (+ shortcut / residual)" unit of ResnetBlock2D (model.py:342-362):
//   * GroupNorm + SiLU of the INPUT are applied while the input tile is staged into LDS (the per-channel sums come from
//     the producing launch, see "statistics" below) -- the normalised tensor never exists in HBM;
//   * bias and the per-sample time-embedding projection are the initial value of the accumulators;
//   * the residual branch is a second K-phase of the same GEMM: extra 1x1 "chunks" read the block input x (or the
//     concatenation [h, skip]) and multiply it with the conv_shortcut weights -- or with an identity matrix when the
//     block has no shortcut conv -- so `x + h` costs no epilogue traffic and is rounded to bf16 exactly once;
//   * statistics: the epilogue emits per-(image, pixel-tile, channel) partial (sum, sum of squares) of the bf16 OUTPUT,
//     which the next launch's prologue folds into its GroupNorm.  Fixed summation order everywhere: bit-reproducible.
//
// Mapping (one workgroup = NW waves of 64):
//   GEMM view  D[n][m] = sum_k W[n][k] * X[k][m],  n = output channel, m = output pixel, k = (tap, input channel).
//   The MFMA "A" operand is the weight tile (32 channels x 16 k), the "B" operand the pixel tile (16 k x 32 pixels),
//   so each lane holds 4 consecutive output channels of ONE pixel per register quad.
//   Input: for every CK-channel chunk the block stages the (TW*s+2) x (TH*s+2) input HALO of its TW x TH output tile
//   in LDS once -- wrap-around along W (azimuth), zeros along H (beams), nearest-x2 and channel-concat resolved in the
//   address computation -- and all 9 taps read it back with a per-tap constant offset.  Chunk c+1 (and c+2) are
//   prefetched into registers / the other half of a double buffer while chunk c computes.
//   Weights: packed on the host per (n-tile, stage) in exactly the padded LDS image (row = 2*CK + 16 bytes: the odd
//   16-byte-slot stride makes every ds_read_b128 lane group conflict-free) and streamed by LDS-DMA
//   (global_load_lds_dwordx4, no VGPR round trip) into a 3-deep ring, two stages ahead of the MFMAs, with counted
//   s_waitcnt vmcnt and ONE raw s_barrier per stage (a stage = one tap of one chunk).
//   Output: accumulators -> bf16 -> LDS [pixel][channel] -> 16-byte fully coalesced channels-last stores.
//   Split-K (small pixel counts): `ksplit` workgroups share an output tile, each sums a slice of the chunks, parks its
//   fp32 accumulators in a slab, and the last one to arrive (agent-scope release/acquire around a ticket counter)
//   adds the slabs in slice order and runs the epilogue.
#include "kernels.h"
#include <type_traits>

namespace rldm {

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to LDS [lds_addr, lds_addr + 1024).  Inline asm so hipcc
// neither counts it in its own vmcnt bookkeeping nor drains it before ds_reads (guide 5.7); m0 is saved/restored
// inside the statement.  lds_addr must be wave-uniform.
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_addr)
        : "memory");
}
// Hot-loop ablation switches (ConvParams::dbg bits 8/16/32/64) and in-kernel timestamps cost scalar branches in every
// stage: compiled in only with -DRLDM_ABLATE (make ABLATE=1); bits 1/2/4 (outside the loop) are always available.
#ifdef RLDM_ABLATE
#define RLDM_DBG(p, bit) (((p).dbg & (bit)) != 0)
#else
#define RLDM_DBG(p, bit) false
#endif

// Same with a wave-uniform (SGPR) base and a per-lane byte offset: no 64-bit VALU address per piece.  m0 is not saved:
// nothing hipcc emits for these kernels on gfx950 reads it (DS instructions have not needed M0 since gfx9).
__device__ __forceinline__ void lds_dma16s(const void* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr)
                 : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// workgroup barrier that orders LDS traffic but leaves VMEM (DMA) operations in flight
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

template <int NW, int BM, int BN, int WM, int WN, int KG, int CK, int TAPS, int TG, int ACH>
__global__ void __launch_bounds__(64 * NW, 1) conv_igemm_kernel(const ConvParams p) {
    constexpr int NT = 64 * NW;               // threads
    constexpr int MI = BM / (32 * WM);        // 32-pixel MFMA tiles per wave
    constexpr int NI = BN / (32 * WN);        // 32-channel MFMA tiles per wave
    constexpr int RS = CK * 2 + 16;           // LDS row stride in bytes (halo rows and weight rows)
    constexpr int KSF = CK / 16;              // MFMA k-steps per tap of a chunk ...
    constexpr int KS = KSF / KG;              // ... and the share of one wave (KG wave groups split the k-steps)
    constexpr int KW = (TAPS == 9) ? 3 : 1;
    constexpr int C8 = CK / 8;                // 16-byte pieces per LDS row
    constexpr int WTILE = BN * RS;            // bytes of one tap of weights (one residual-phase stage)
    constexpr int WCH = WTILE / 16;           // ... in 16-byte pieces
    constexpr int STILE = TG * WTILE;         // bytes of one main-phase stage (TG taps) = one ring slot
    constexpr int SCH = STILE / 16;
    constexpr int SPC = TAPS / TG;            // stages per main-phase chunk
    constexpr int DPT = (SCH + NT - 1) / NT;  // DMA instructions per wave per main-phase stage
    constexpr int DPTR = (WCH + NT - 1) / NT; // ... per residual-phase stage
    constexpr int PPT = (DPT + TG - 1) / TG;  // DMA instructions issued in the shadow of one tap
    constexpr int NBUF = (TG == 9) ? 2 : 3;   // weight ring depth
    constexpr int DD = NBUF - 1;              // the DMA of stage s + DD is issued during stage s
    static_assert(TAPS % TG == 0 && (TG == 1 || TG == 9), "taps per stage");
    static_assert(PPT >= DPTR, "residual-stage DMA must fit the first tap's shadow");
    constexpr int ERS = BN * 2 + 16;          // epilogue staging row stride (bytes)
    constexpr int NC8 = BN / 8;               // 16-byte pieces per output pixel row
    static_assert(WM * WN * KG == NW && KSF % KG == 0 && KS >= 1, "wave grid: WM x WN spatial, KG k-groups");
    static_assert(MI >= 1 && NI >= 1, "tile shape");
    static_assert(WCH >= 64 && WTILE % 16 == 0, "weight stage too small for a full-wave DMA");
    static_assert(NT % NC8 == 0 && NT % C8 == 0, "thread count must be a multiple of the pieces per row");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / (WM * WN);           // k-group: this wave's k-steps are [kg*KS, (kg+1)*KS) of every tap
    const int wmn = wave - kg * (WM * WN);
    const int wm = wmn % WM, wn = wmn / WM;
    const int kh = lane >> 5, l31 = lane & 31;
    // the sampler's step index: advanced by the first launch of a step (conv_in: nothing in it reads the index; every later launch of
    // the step sees the new value across the kernel boundary)
    if (p.step_inc && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) *p.step_inc += 1;
    int ts_n = 0;
    // in-kernel timeline (ABLATE builds): s_memtime stamps go to the last 512 bytes of the LDS allocation (no VMEM
    // traffic that would perturb the counted waits) and are copied out at the very end by `flush_stamps`
    auto stamp = [&]() {
#ifdef RLDM_ABLATE
        if (p.ts && blockIdx.x < 4 && tid == 0 && ts_n < 64)
            reinterpret_cast<unsigned long long*>(smem + p.lds_total - 512)[ts_n++] = __builtin_amdgcn_s_memtime();
#else
        (void)ts_n;
#endif
    };
    auto flush_stamps = [&]() {
#ifdef RLDM_ABLATE
        if (p.ts && blockIdx.x < 4 && tid == 0)
            for (int i = 0; i < ts_n; ++i) p.ts[blockIdx.x * 64 + i] = reinterpret_cast<unsigned long long*>(smem + p.lds_total - 512)[i];
#endif
    };
    stamp();

    // ---- which tile / K slice -----------------------------------------------------------------------------------
    const int tiles_h = p.tiles_h;
    const int tiles_img = p.tiles_img;
    int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int ksl = lid % p.ksplit;
    lid /= p.ksplit;
    const int nt = lid % p.ntile_n;
    int mt = lid / p.ntile_n;
    const int tile_id = lid;                  // (m-tile, n-tile) index: ticket / slab slot
    const int b = mt / tiles_img;
    mt -= b * tiles_img;
    const int tw = mt / tiles_h, th = mt - tw * tiles_h;
    const int w0 = tw * p.TW, h0 = th * p.TH;
    const int npx = p.TW * p.TH;

    // ---- bias (+ time-embedding projection of this sample): fetched first, parked in LDS, becomes the initial value
    //      of the accumulators right before the main loop (K-slices > 0 start at zero) -------------------------------
    float bias_v = 0.f;
    if (tid < BN) {
        const int ch = nt * BN + tid;
        if (ksl == 0) {
            bias_v = p.bias[ch];
            if (p.temb && ch < p.N) {
                const int step = p.step_ptr ? *p.step_ptr : 0;
                bias_v += p.temb[(size_t)(step * p.temb_rows_per_step + (p.temb_per_sample ? b : 0)) * p.temb_ld + ch];
            }
        }
    }

    const int Cin = p.C0 + p.C1;              // main phase: TAPS taps per chunk, GroupNorm (+SiLU) prologue
    const int NCC = Cin / CK;
    const int NCB = (p.R0 + p.R1) / CK;       // residual phase: 1 (centre) tap per chunk, raw input
    const int NCT = NCC + NCB;
    // this block's chunk range [cbeg, cend)
    const int cper = (NCT + p.ksplit - 1) / p.ksplit;
    const int cbeg = ksl * cper;
    const int cend = (cbeg + cper < NCT) ? cbeg + cper : NCT;
    auto stages_before = [&](int c) { return c <= NCC ? c * SPC : NCC * SPC + (c - NCC); };
    const int NMS = NCC * SPC;                // main-phase stages of the whole conv (absolute stage index < NMS: TG taps)
    const int sbeg = stages_before(cbeg);
    const int NS = (p.dbg & 2) ? 0 : stages_before(cend) - sbeg;

    const int THv = (p.TH - 1) * p.stride + KW;
    const int TWv = (p.TW - 1) * p.stride + KW;
    const int colb = p.colb;                  // halo column pitch (bytes): >= THv*RS, chosen so fragment reads are conflict-free
    const int abytes = TWv * colb;
    const int Wv = p.Win * p.up, Hv = p.Hin * p.up;
    const int upshift = p.up - 1;             // up in {1,2}

    unsigned char* sW = smem;                                  // NBUF * STILE
    unsigned char* sA = smem + NBUF * STILE;                   // 2 * abytes
    float* sGa = reinterpret_cast<float*>(sA + 2 * abytes);    // Cin
    float* sGs = sGa + Cin;                                    // Cin
    float* sBias = sGs + Cin;                                  // BN
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // ---- weight stream: per n-tile [NMS main stages of STILE bytes][NCB residual stages of WTILE bytes] ---------------
    const unsigned char* wtile0 = reinterpret_cast<const unsigned char*>(p.wpk) + (size_t)nt * (NCC * TAPS + NCB) * WTILE;
    auto wbyte = [&](int sa) __attribute__((always_inline)) {     // byte offset of absolute stage sa
        return sa < NMS ? (size_t)sa * STILE : (size_t)NMS * STILE + (size_t)(sa - NMS) * WTILE;
    };
    const unsigned lane16 = (unsigned)lane * 16u;
    auto issue_pieces = [&](const unsigned char* src, unsigned dst, int lo, int hi, int last) __attribute__((always_inline)) {
        // pieces [lo, hi) of a stage (compile-time bounds); every wave issues full-wave DMAs, passes beyond the stage's
        // last 64-piece block (`last`) overlap it instead of running short.  All address math is scalar.
#pragma unroll
        for (int i = 0; i < DPT; ++i) {
            if (i < lo || i >= hi) continue;
            int piece = i * NT + wave * 64;
            piece = piece > last ? last : piece;
            lds_dma16s(src + (size_t)piece * 16, lane16, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + (unsigned)piece * 16)));
        }
    };
#pragma unroll
    for (int j = 0; j < DD; ++j)
        if (j < NS) {
            const bool m = (sbeg + j) < NMS;
            issue_pieces(wtile0 + wbyte(sbeg + j), lds0 + (unsigned)(j * STILE), 0, DPT, (m ? SCH : WCH) - 64);
        }

    // ---- halo staging: thread-constant source pixel / LDS offset of each of its ACH 16-byte pieces ----------------
    const int atotal = TWv * THv * C8;
    int apix[ACH];          // source pixel index ((b*Win + sw)*Hin + sh), or -1 for zero padding / out of range
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        const int q = tid + i * NT;
        const int slot = q / C8, c8 = q - slot * C8;
        const int vwl = (slot * p.magic_thv) >> 20, vhl = slot - vwl * THv;    // exact for slot * THv < 2^20
        const int vh = h0 * p.stride - p.pad_lo + vhl;
        int vw = w0 * p.stride - p.pad_lo + vwl;
        vw = vw < 0 ? vw + Wv : (vw >= Wv ? vw - Wv : vw);          // azimuth: wrap-around
        const bool ok = q < atotal && vh >= 0 && vh < Hv;            // beams: zero padding
        apix[i] = ok ? ((b * p.Win + (vw >> upshift)) * p.Hin + (vh >> upshift)) : -1;
    }
    const int my_c8 = (tid % C8) * 8;          // NT % C8 == 0: a thread always handles the same 8-channel column

    uint4 areg[ACH];
    float4 ga0, ga1, gs0, gs1;                 // GroupNorm affine of the chunk held in areg (this thread's 8 channels)
    bool anorm = false;
    const bool gn = p.st0 != nullptr && !(p.dbg & 4);
    // (fields copied to locals: selecting between kernel-argument fields by address sent the whole struct to scratch)
    const bf16_t* const gx0 = p.x0;
    const bf16_t* const gx1 = p.x1;
    const bf16_t* const gr0 = p.r0;
    const bf16_t* const gr1 = p.r1;
    const int nC0 = p.C0, nC1 = p.C1, nR0 = p.R0, nR1 = p.R1;
    auto load_a = [&](int cc) __attribute__((always_inline)) {
        const bool main_phase = cc < NCC;
        const int c = (main_phase ? cc : cc - NCC) * CK + my_c8;
        const int split = main_phase ? nC0 : nR0;
        const bool first = c < split;
        const bf16_t* t0 = main_phase ? gx0 : gr0;
        const bf16_t* t1 = main_phase ? gx1 : gr1;
        const bf16_t* base = first ? t0 + c : t1 + (c - split);
        const int ld = first ? split : (main_phase ? nC1 : nR1);
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int pix = apix[i] < 0 ? 0 : apix[i];
            areg[i] = *reinterpret_cast<const uint4*>(base + (size_t)pix * ld);
        }
    };
    auto load_affine = [&](int cc) __attribute__((always_inline)) {           // after the GroupNorm finalize; cheap LDS reads
        anorm = gn && cc < NCC;
        if (anorm) {
            const int c = cc * CK + my_c8;
            ga0 = *reinterpret_cast<const float4*>(sGa + c);
            ga1 = *reinterpret_cast<const float4*>(sGa + c + 4);
            gs0 = *reinterpret_cast<const float4*>(sGs + c);
            gs1 = *reinterpret_cast<const float4*>(sGs + c + 4);
        }
    };
    auto store_piece = [&](int cc, int i) __attribute__((always_inline)) {    // i must fold to a constant (register array)
        unsigned char* dstbuf = sA + (cc & 1) * abytes;
        uint4 v = areg[i];
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
        // LDS offset recomputed (a handful of VALU ops per chunk) instead of held in registers across the main loop
        const int q = tid + i * NT;
        const int slot = q / C8, c8 = q - slot * C8;
        const int vwl = (slot * p.magic_thv) >> 20, vhl = slot - vwl * THv;
        if (q < atotal) *reinterpret_cast<uint4*>(dstbuf + vwl * colb + vhl * RS + c8 * 16) = v;
    };
    auto store_a = [&](int cc) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) store_piece(cc, i);
    };
    stamp();
    // ---- GroupNorm finalize: per-channel partial (sum, sumsq) of the producers -> per-channel affine a*x + s.  The
    // statistics, gamma and beta of this thread's channels are requested first, then the first halo chunk: one memory
    // round trip; every channel's thread folds its own group (no serial phase) --------------------------------------------
    double gS[2] = {0.0, 0.0}, gSS[2] = {0.0, 0.0};           // Cin <= 2 * NT
    float g_gamma[2] = {0.f, 0.f}, g_beta[2] = {0.f, 0.f};
    if (gn) {
        const float2* const gs0p = p.st0;
        const float2* const gs1p = p.st1;
        const int nP0 = p.P0, nP1 = p.P1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = tid + j * NT;
            if (t < Cin) {
                const bool first = t < nC0;
                const int c = first ? t : t - nC0;
                const int C = first ? nC0 : nC1;
                const int P = first ? nP0 : nP1;
                const float2* src = (first ? gs0p : gs1p) + (size_t)b * P * C + c;
                g_gamma[j] = p.gn_gamma[t];
                g_beta[j] = p.gn_beta[t];
                double S = 0.0, SS = 0.0;
                int q = 0;
                for (; q + 16 <= P; q += 16) {
                    float2 v[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = src[(size_t)(q + i) * C];
#pragma unroll
                    for (int i = 0; i < 16; ++i) { S += (double)v[i].x; SS += (double)v[i].y; }
                }
                for (; q + 4 <= P; q += 4) {
                    float2 v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = src[(size_t)(q + i) * C];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { S += (double)v[i].x; SS += (double)v[i].y; }
                }
                for (; q < P; ++q) {
                    const float2 v = src[(size_t)q * C];
                    S += (double)v.x;
                    SS += (double)v.y;
                }
                gS[j] = S;
                gSS[j] = SS;
            }
        }
    }
    if (NS > 0) load_a(cbeg);                 // in flight while the GroupNorm statistics are folded
    if (gn) {
        double* sD = reinterpret_cast<double*>(sA);           // scratch: [2][Cin] doubles (halo not yet written)
        const int cpg = Cin / p.gn_groups;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = tid + j * NT;
            if (t < Cin) { sD[t] = gS[j]; sD[Cin + t] = gSS[j]; }
        }
        __syncthreads();
        float ga[2], gs[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = tid + j * NT;
            if (t < Cin) {
                const int g0 = ((t * p.magic_cpg) >> 20) * cpg;
                double S = 0.0, SS = 0.0;
                for (int i = 0; i < cpg; ++i) {
                    S += sD[g0 + i];
                    SS += sD[Cin + g0 + i];
                }
                const double inv_n = (double)p.gn_inv_n;
                const double mean = S * inv_n;
                double var = SS * inv_n - mean * mean;
                var = var < 0.0 ? 0.0 : var;
                ga[j] = g_gamma[j] * __builtin_amdgcn_rsqf((float)var + p.gn_eps);      // hardware rsqrt, 1 ulp
                gs[j] = g_beta[j] - (float)mean * ga[j];
            }
        }
        __syncthreads();                       // sD fully consumed before sGa/sGs (which may overlap it) and the halo are written
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = tid + j * NT;
            if (t < Cin) { sGa[t] = ga[j]; sGs[t] = gs[j]; }
        }
        __syncthreads();
    }
    if (tid < BN) sBias[tid] = bias_v;         // (after the GroupNorm scratch is dead) visible after the main loop's barriers
    stamp();
    int loaded = cbeg, stored = cbeg;          // halo chunks whose loads were issued / whose LDS image is complete
    if (NS > 0) {
        load_affine(cbeg);
        store_a(cbeg);
        if ((SPC == 1 || cbeg >= NCC) && cbeg + 1 < cend) { load_a(cbeg + 1); loaded = cbeg + 1; }   // 1-stage chunks: two ahead
    }

    // ---- per-lane LDS byte offsets of the MFMA operands ----------------------------------------------------------
    int xoff[MI], woff[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        int pidx = wm * (MI * 32) + mi * 32 + l31;
        if (pidx >= npx) pidx = 0;            // masked at the store
        const int pw = pidx >> p.th_shift, ph = pidx - (pw << p.th_shift);
        xoff[mi] = (pw * p.stride) * colb + (ph * p.stride) * RS + kh * 16 + kg * (KS * 32);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) woff[ni] = (wn * (NI * 32) + ni * 32 + l31) * RS + kh * 16 + kg * (KS * 32);

    constexpr int AS = (NI * MI >= 4) ? 1 : 4 / (NI * MI);     // accumulator sets
    f32x16 acc[AS][NI][MI];
    lds_barrier();                             // sBias (and, without GroupNorm, the first halo chunk) written
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            float4 bv = *reinterpret_cast<const float4*>(sBias + wn * (NI * 32) + ni * 32 + 8 * r4 + 4 * kh);
            if (kg != 0) bv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                acc[0][ni][mi][r4 * 4 + 0] = bv.x; acc[0][ni][mi][r4 * 4 + 1] = bv.y;
                acc[0][ni][mi][r4 * 4 + 2] = bv.z; acc[0][ni][mi][r4 * 4 + 3] = bv.w;
#pragma unroll
                for (int a = 1; a < AS; ++a) {
                    acc[a][ni][mi][r4 * 4 + 0] = 0.f; acc[a][ni][mi][r4 * 4 + 1] = 0.f;
                    acc[a][ni][mi][r4 * 4 + 2] = 0.f; acc[a][ni][mi][r4 * 4 + 3] = 0.f;
                }
            }
        }
    stamp();

    // ---- main loop ---------------------------------------------------------------------------------------------------
    // A stage = TG taps of one CK-channel chunk (main phase) or the centre tap of one residual chunk, one raw barrier per
    // stage.  A wave can start an MFMA only every 32 cycles while the pipe needs 16, so everything else is placed in the
    // shadows between the wave's own MFMAs (tools/ubench/stage_model.hip: 842 -> 667 ns per stage); the tap loops are
    // branch-free (a weight DMA past the last stage re-copies a valid stage into a free slot).
    //   TG == 1 (256-pixel tiles, LDS-limited to one tap of weights per stage): right after the MFMAs of k-step ks the
    //     wave re-loads the SAME pixel-fragment registers with the next stage's k-step ks (the halo is stable across a
    //     chunk's taps; a prefetch from a halo image that is not complete yet is repeated after the barrier) and issues
    //     its share of the LDS-DMA of the stage DD ahead; only the weight fragments are read after the barrier.
    //   TG == 9 (small pixel tiles: the whole 3x3 of a 32-channel chunk is resident, one barrier per chunk): a ring of
    //     NSET fragment register sets, tap tt + NSET - 1 is read while tap tt computes.  Small wave tiles rotate over AS
    //     accumulator sets so that consecutive MFMAs never depend on each other (summed before the epilogue).
    {
        constexpr int NSET = (TG == 1) ? 1 : 9;     // TG == 9: the fragments of all nine taps are read right after the barrier
        int cc = cbeg, sg = 0;                 // chunk, stage within the chunk
        int toff = (cbeg < NCC || TAPS == 1) ? 0 : colb + RS;     // tap offset of the stage's first tap (residual: centre)
        int tj = 0;                            // TG == 1: column of the 3x3 the stage is in
        int wslot = 0, dslot = DD % NBUF;      // ring slot of stage s / of the stage whose DMA stage s issues
        int dsa = sbeg + DD;                   // absolute index of the stage whose DMA stage s issues
        bool x_ready = false;                  // TG == 1: the stage's pixel fragments were prefetched (and valid)
        bool hl_prev = loaded > stored;        // halo loads were issued after the last weight DMA
        bool pm_prev = true;                   // the previous stage was a main-phase stage (it issued DPT DMAs, else PPT)
        bf16x8 wf[NSET][KS][NI], xf[NSET][KS][MI];
        int s = 0;
#define RLDM_READ_TAP(SET, WPTR, XPTR)                                                                               \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                          \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                        \
                wf[SET][ks][ni] = *reinterpret_cast<const bf16x8*>((WPTR) + woff[ni] + ks * 32);                     \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                        \
                xf[SET][ks][mi] = *reinterpret_cast<const bf16x8*>((XPTR) + xoff[mi] + ks * 32);                     \
        }
#define RLDM_MMA(SET, KSTEP, AIDX)                                                                                   \
        if (!nomma) {                                                                                                \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                        \
                _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                    \
                    acc[(AIDX) % AS][ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                              \
                        wf[SET][KSTEP][ni], xf[SET][KSTEP][mi], acc[(AIDX) % AS][ni][mi], 0, 0, 0);                  \
        } else {                                                                                                     \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(wf[SET][KSTEP][ni]));            \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(xf[SET][KSTEP][mi]));            \
        }
        // tap TT of a TG == 9 stage: the MFMAs of fragment set TT and, in their shadow, the DMA share of the tap
#define RLDM_TAP9(TT)                                                                                                \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                          \
            RLDM_MMA(TT, ks, (TT) * KS + ks)                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }
        // One stage; `cmain_c` selects the main-phase / residual-phase code at compile time: two loops, no join of the two
        // accumulator-register layouts inside one loop body.
        auto stage = [&](auto cmain_c) __attribute__((always_inline)) {
            // W(s) has landed once only the operations issued after its DMAs are outstanding (VMEM retires in order)
            if (NBUF == 2) {
                // the previous stage requested W(s) first thing after its barrier; only its halo loads are younger
                if (hl_prev) wait_vmcnt<ACH>(); else wait_vmcnt<0>();
            } else if (s + 1 >= NS) {
                wait_vmcnt<0>();
            } else if (pm_prev) {
                if (hl_prev) wait_vmcnt<DPT + ACH>(); else wait_vmcnt<DPT>();
            } else {
                if (hl_prev) wait_vmcnt<PPT + ACH>(); else wait_vmcnt<PPT>();
            }
            if (RLDM_DBG(p, 64)) stamp();
            lds_barrier();
            if (RLDM_DBG(p, 64)) stamp();
            const int vis = stored;            // halo images complete and visible to every wave
            constexpr bool cmain = decltype(cmain_c)::value;     // main-phase stage (TG taps) or residual-phase stage (1 tap)
            const unsigned char* wbase = sW + wslot * STILE;
            const unsigned char* abase = sA + (cc & 1) * abytes + toff;
            // DMA target: stage s + DD, or (past the end) this stage again: a valid source into a free slot
            const int dst_sa = (s + DD < NS) ? dsa : sbeg + s;
            const bool dmain = dst_sa < NMS;
            const int dlast = (dmain ? SCH : WCH) - 64;
            const unsigned char* dsrc = wtile0 + wbyte(dst_sa);
            const unsigned ddst = lds0 + (unsigned)(dslot * STILE);
            const bool nodma = RLDM_DBG(p, 16);        // tuning ablations: no weight streaming after the prologue,
            const bool nomma = RLDM_DBG(p, 8);         // no MFMAs
            // TG == 9: the whole next stage is requested right at the barrier (its slot was vacated by the stage before
            // this one), so the DMA has the entire stage to land -- and every later VMEM operation of the stage (the halo
            // loads) is younger, which is what the counted wait at the top of the next stage relies on
            if (TG == 9 && !nodma) issue_pieces(dsrc, ddst, 0, cmain ? DPT : PPT, dlast);
            if (TG == 1 || !cmain) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) wf[0][ks][ni] = *reinterpret_cast<const bf16x8*>(wbase + woff[ni] + ks * 32);
                if (TG != 1 || !x_ready) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) xf[0][ks][mi] = *reinterpret_cast<const bf16x8*>(abase + xoff[mi] + ks * 32);
                }
            } else {
                // the small wave tiles do 1-2 MFMAs per tap -- far less than an LDS round trip -- so all nine taps'
                // fragments are requested at once and the MFMAs consume them as they return
                RLDM_READ_TAP(0, wbase, abase)
                RLDM_READ_TAP(1, wbase + WTILE, abase + RS)
                RLDM_READ_TAP(2, wbase + 2 * WTILE, abase + 2 * RS)
                RLDM_READ_TAP(3, wbase + 3 * WTILE, abase + colb)
                RLDM_READ_TAP(4, wbase + 4 * WTILE, abase + colb + RS)
                RLDM_READ_TAP(5, wbase + 5 * WTILE, abase + colb + 2 * RS)
                RLDM_READ_TAP(6, wbase + 6 * WTILE, abase + 2 * colb)
                RLDM_READ_TAP(7, wbase + 7 * WTILE, abase + 2 * colb + RS)
                RLDM_READ_TAP(8, wbase + 8 * WTILE, abase + 2 * colb + 2 * RS)
            }
            hl_prev = false;
            // halo pipeline: write the chunk whose loads are in flight (GroupNorm + SiLU: VALU work), load the one after.
            // With two k-groups on a TG == 9 stage, group 1 does it before its taps and group 0 after, so that one
            // group's VALU work runs under the other group's MFMAs.
            const bool halo_first = !(KG == 2 && TG == 9 && kg == 0) || RLDM_DBG(p, 128);
#define RLDM_HALO_STEP                                                                                               \
            {                                                                                                        \
                if (loaded > stored && stored + 1 <= cc + 1) { load_affine(stored + 1); store_a(stored + 1); ++stored; } \
                if (loaded == stored && loaded + 1 < cend && loaded <= cc + 1) { load_a(loaded + 1); ++loaded; hl_prev = true; } \
            }
            const int cc_now = cc;
            (void)cc_now;
            if (RLDM_DBG(p, 64)) stamp();
            if (halo_first && !RLDM_DBG(p, 32)) RLDM_HALO_STEP
            if (RLDM_DBG(p, 64)) stamp();
            // coordinates of the next stage
            int ncc = cc, nsg = sg + 1, ntoff = toff, ntj = tj;
            if (nsg == (cmain ? SPC : 1)) {
                nsg = 0; ++ncc; ntj = 0;
                ntoff = (ncc < NCC || TAPS == 1) ? 0 : colb + RS;
            } else {                           // TG == 1: next tap of the 3x3
                ntoff += RS;
                if (++ntj == 3) { ntj = 0; ntoff += colb - 3 * RS; }
            }
            const unsigned char* nbase = sA + (ncc & 1) * abytes + ntoff;
            __builtin_amdgcn_sched_barrier(0);
            if (TG == 1 || !cmain) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    RLDM_MMA(0, ks, ks)
                    __builtin_amdgcn_sched_barrier(0);
                    // the stage's DMA share, spread over its k-steps
#pragma unroll
                    for (int i = ks; i < PPT; i += KS) if (TG == 1 && !nodma) issue_pieces(dsrc, ddst, i, i + 1, dlast);
                    if (TG == 1) {             // in-place prefetch of the next stage's pixel fragments
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) xf[0][ks][mi] = *reinterpret_cast<const bf16x8*>(nbase + xoff[mi] + ks * 32);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                RLDM_TAP9(0) RLDM_TAP9(1) RLDM_TAP9(2) RLDM_TAP9(3) RLDM_TAP9(4) RLDM_TAP9(5) RLDM_TAP9(6) RLDM_TAP9(7) RLDM_TAP9(8)
            }
            if (RLDM_DBG(p, 64)) stamp();
            if (!halo_first && !RLDM_DBG(p, 32)) RLDM_HALO_STEP
            if (RLDM_DBG(p, 64)) stamp();
#undef RLDM_HALO_STEP
            pm_prev = cmain;
            x_ready = (s + 1 < NS) && (ncc <= vis);
            cc = ncc; sg = nsg; toff = ntoff; tj = ntj;
            wslot = (wslot + 1 == NBUF) ? 0 : wslot + 1;
            dslot = (dslot + 1 == NBUF) ? 0 : dslot + 1;
            ++dsa;
            ++s;
            stamp();
        };
        const int nsm = (NMS - sbeg) < 0 ? 0 : ((NMS - sbeg) < NS ? (NMS - sbeg) : NS);    // main-phase stages of this slice
#pragma unroll 1
        while (s < nsm) stage(std::true_type{});
#pragma unroll 1
        while (s < NS) stage(std::false_type{});
#undef RLDM_TAP9
#undef RLDM_MMA
#undef RLDM_READ_TAP
        wait_vmcnt<0>();                       // the unconditional tail DMAs
        // fold the accumulator sets
#pragma unroll
        for (int a = 1; a < AS; ++a)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][ni][mi][r] += acc[a][ni][mi][r];
    }
    lds_barrier();                             // all waves are done with the ring / halo: LDS is reused below
    stamp();
    if constexpr (KG == 2) {
        // k-group 1 hands its partial accumulators to group 0 through LDS (fragment order: conflict-free float4s)
        constexpr int NTG = 64 * WM * WN;
        float* sX = reinterpret_cast<float*>(smem);
        const int tg = tid - kg * NTG;
        if (kg == 1) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
                        *reinterpret_cast<float4*>(sX + ((((ni * MI + mi) * 4 + r4) * NTG) + tg) * 4) =
                            make_float4(acc[0][ni][mi][r4 * 4], acc[0][ni][mi][r4 * 4 + 1], acc[0][ni][mi][r4 * 4 + 2],
                                        acc[0][ni][mi][r4 * 4 + 3]);
        }
        lds_barrier();
        if (kg == 0) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const float4 v = *reinterpret_cast<const float4*>(sX + ((((ni * MI + mi) * 4 + r4) * NTG) + tg) * 4);
                        acc[0][ni][mi][r4 * 4] += v.x; acc[0][ni][mi][r4 * 4 + 1] += v.y;
                        acc[0][ni][mi][r4 * 4 + 2] += v.z; acc[0][ni][mi][r4 * 4 + 3] += v.w;
                    }
        }
        lds_barrier();                         // sX is dead before the staging below reuses the same bytes
    }

    // ---- split-K: park the slice, last arriver combines (guide 6 G16 recipe: release -> ticket -> acquire) ----------
    if (p.ksplit > 1) {
        int* flag = reinterpret_cast<int*>(smem);
        float* slab = p.slab + ((size_t)tile_id * p.ksplit) * (size_t)(BM * BN);
        float* mine = slab + (size_t)ksl * (BM * BN);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 v = make_float4(acc[0][ni][mi][r4 * 4], acc[0][ni][mi][r4 * 4 + 1], acc[0][ni][mi][r4 * 4 + 2],
                                                 acc[0][ni][mi][r4 * 4 + 3]);
                    *reinterpret_cast<float4*>(mine + ((((ni * MI + mi) * 4 + r4) * NT) + tid) * 4) = v;
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int t = __hip_atomic_fetch_add(p.ticket + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (t == p.ksplit - 1);
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(p.ticket + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
            }
            *flag = last;
        }
        __syncthreads();
        if (!*flag) return;
        __syncthreads();                       // flag is read by everyone before the staging below overwrites it
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int k = 0; k < p.ksplit; ++k) {      // slice order: deterministic
                        const float4 v = *reinterpret_cast<const float4*>(
                            slab + (size_t)k * (BM * BN) + ((((ni * MI + mi) * 4 + r4) * NT) + tid) * 4);
                        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
                    }
                    acc[0][ni][mi][r4 * 4] = sum.x; acc[0][ni][mi][r4 * 4 + 1] = sum.y;
                    acc[0][ni][mi][r4 * 4 + 2] = sum.z; acc[0][ni][mi][r4 * 4 + 3] = sum.w;
                }
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------------
    if (p.y_nchw) {                            // fp32 NCHW (network outputs: few channels), straight from the registers
        if (kg != 0) return;
        // fused scheduler step (sched_step_kernel's arithmetic on the value just computed)
        const bool sched = p.sch.coef_table != nullptr;
        float c0 = 1.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f;
        const float* nz = nullptr;
        if (sched) {
            const int step = *p.sch.step_ptr;
            const float* c = p.sch.coef_table + 5 * step;
            c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3]; c4 = c[4];
            if (p.sch.noise && c4 != 0.f) nz = p.sch.noise + (size_t)step * p.sch.noise_step_stride;
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int pidx = wm * (MI * 32) + mi * 32 + l31;
            if (pidx >= npx) continue;
            const int pw = pidx >> p.th_shift, ph = pidx - (pw << p.th_shift);
            const int ow = w0 + pw, oh = h0 + ph;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = nt * BN + wn * (NI * 32) + ni * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
                    if (ch < p.N && !(p.dbg & 1)) {
                        const size_t i = (((size_t)b * p.N + ch) * p.Wout + ow) * p.Hout + oh;
                        const float e = acc[0][ni][mi][r];
                        p.y_nchw[i] = e;
                        if (sched) {
                            const float x = p.sch.x[i];
                            float prev = sched_prev(p.sch.mode, c0, c1, c2, c3, x, e);
                            if (nz) prev += c4 * nz[i];
                            p.sch.x_prev[i] = prev;
                            if (p.sch.pack) p.sch.pack[(((size_t)b * p.Wout + ow) * p.Hout + oh) * p.sch.pack_ld + ch] = f32_to_bf16(prev);
                        }
                    }
                }
        }
        return;
    }
    // (1) accumulators -> bf16 -> LDS [pixel][channel]
    unsigned char* sE = smem;
    if (kg == 0) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int pidx = wm * (MI * 32) + mi * 32 + l31;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int chl = wn * (NI * 32) + ni * 32 + 8 * r4 + 4 * kh;
                uint2 o;
                o.x = pack_bf16x2(acc[0][ni][mi][r4 * 4 + 0], acc[0][ni][mi][r4 * 4 + 1]);
                o.y = pack_bf16x2(acc[0][ni][mi][r4 * 4 + 2], acc[0][ni][mi][r4 * 4 + 3]);
                *reinterpret_cast<uint2*>(sE + pidx * ERS + chl * 2) = o;
            }
    }
    }
    lds_barrier();
    stamp();
    // (2) 16-byte pieces, channel-fastest: fully coalesced stores; per-channel statistics of exactly the stored values
    const int c8 = tid % NC8;
    const int chg = nt * BN + c8 * 8;
    const bool cvalid = chg < p.N;
    const bool vec_ok = (p.N & 7) == 0;        // rows 16-byte aligned and whole pieces only
    float s8[8], q8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s8[e] = q8[e] = 0.f;
    for (int pidx = tid / NC8; pidx < npx; pidx += NT / NC8) {
        const uint4 v = *reinterpret_cast<const uint4*>(sE + pidx * ERS + c8 * 16);
        const int pw = pidx >> p.th_shift, ph = pidx - (pw << p.th_shift);
        const size_t pix = ((size_t)b * p.Wout + (w0 + pw)) * p.Hout + (h0 + ph);
        if (cvalid && !(p.dbg & 1)) {
            if (vec_ok) {
                *reinterpret_cast<uint4*>(p.y + pix * p.y_ld + chg) = v;
            } else {                           // odd channel counts (test shapes): element stores
                const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (chg + e < p.N) p.y[pix * p.y_ld + chg + e] = (bf16_t)((e & 1) ? (wv[e >> 1] >> 16) : (wv[e >> 1] & 0xffffu));
            }
        }
        const float f[8] = {bf16lo(v.x), bf16hi(v.x), bf16lo(v.y), bf16hi(v.y),
                            bf16lo(v.z), bf16hi(v.z), bf16lo(v.w), bf16hi(v.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s8[e] += f[e];
            q8[e] += f[e] * f[e];
        }
    }
    stamp();
    if (p.y_stats) {
        // cross-thread reduction through LDS, fixed order: sS[kind][g = tid / NC8][channel], 32-byte writes, then one
        // thread per (kind, channel) adds the NT / NC8 partials
        constexpr int G = NT / NC8;
        float* sS = reinterpret_cast<float*>(sE + BM * ERS);
        const int g = tid / NC8;
        *reinterpret_cast<float4*>(sS + g * BN + c8 * 8) = make_float4(s8[0], s8[1], s8[2], s8[3]);
        *reinterpret_cast<float4*>(sS + g * BN + c8 * 8 + 4) = make_float4(s8[4], s8[5], s8[6], s8[7]);
        *reinterpret_cast<float4*>(sS + (G + g) * BN + c8 * 8) = make_float4(q8[0], q8[1], q8[2], q8[3]);
        *reinterpret_cast<float4*>(sS + (G + g) * BN + c8 * 8 + 4) = make_float4(q8[4], q8[5], q8[6], q8[7]);
        lds_barrier();                         // (not __syncthreads: that would also drain the output stores)
        for (int t = tid; t < 2 * BN; t += NT) {
            const int kind = t / BN, c = t - kind * BN;
            float S = 0.f;
#pragma unroll 8
            for (int j = 0; j < G; ++j) S += sS[(kind * G + j) * BN + c];
            if (nt * BN + c < p.N)
                reinterpret_cast<float*>(p.y_stats + ((size_t)b * tiles_img + mt) * p.N + nt * BN + c)[kind] = S;
        }
    }
    stamp();
    flush_stamps();
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
//          NW   BM   BN  WM WN KG  CK TAPS TG ACH
// Large pixel tiles (256 px: L0, VAE) are LDS-limited to one tap per stage; the small ones keep the whole 3x3 of a
// 32-channel chunk resident (TG = 9: one barrier per chunk) and run two k-groups of waves on the same tile (KG = 2):
// two waves per SIMD, the GroupNorm+SiLU staging work split over twice the threads and overlapped with the other
// group's MFMAs.  ACH = halo 16-byte pieces a thread stages per chunk.
#define RLDM_CONV_INSTANCES(X)                                                                                  \
    X(8, 256, 128, 4, 2, 1, 64, 9, 1, 6) X(8, 256, 64, 4, 2, 1, 64, 9, 1, 6) X(8, 256, 32, 8, 1, 1, 64, 9, 1, 6) \
    X(4, 128, 128, 2, 2, 1, 64, 9, 1, 7) X(8, 128, 64, 2, 2, 2, 32, 9, 9, 2) X(8, 64, 64, 2, 2, 2, 32, 9, 9, 2)  \
    X(4, 128, 32, 4, 1, 1, 64, 9, 1, 7)                                                                         \
    X(8, 256, 128, 4, 2, 1, 64, 1, 1, 4) X(8, 256, 64, 4, 2, 1, 64, 1, 1, 4)                                     \
    X(4, 128, 128, 2, 2, 1, 64, 1, 1, 4) X(4, 128, 64, 2, 2, 1, 64, 1, 1, 4) X(4, 64, 64, 2, 2, 1, 64, 1, 1, 2)  \
    X(8, 256, 128, 4, 2, 1, 16, 9, 1, 2) X(8, 256, 64, 4, 2, 1, 16, 9, 1, 2)                                     \
    X(4, 128, 128, 2, 2, 1, 16, 9, 1, 2) X(4, 128, 64, 2, 2, 1, 16, 9, 1, 2) X(4, 64, 64, 2, 2, 1, 16, 9, 1, 2)  \
    X(4, 128, 32, 4, 1, 1, 16, 9, 1, 2) X(4, 64, 64, 2, 2, 1, 16, 1, 1, 1)

struct ConvInst {
    int NW, BM, BN, KG, CK, taps, TG, ACH;
};
static const ConvInst* find_inst(const ConvTile& t) {
    static const ConvInst table[] = {
#define X(nw_, bm_, bn_, wm_, wn_, kg_, ck_, taps_, tg_, ach_) {nw_, bm_, bn_, kg_, ck_, taps_, tg_, ach_},
        RLDM_CONV_INSTANCES(X)
#undef X
    };
    for (const auto& i : table)
        if (i.BM == t.BM && i.BN == t.BN && i.CK == t.CK && i.taps == t.taps) return &i;
    return nullptr;
}

bool conv_tile_supported(const ConvTile& t) { return find_inst(t) != nullptr; }

// Halo column pitch in bytes.  A 32-pixel MFMA tile covers 32/TH columns x TH rows of the output tile; ds_read_b128 is
// served in fixed 16-lane groups (MI355X_MICROARCH LDS table) that straddle those columns, and a group is conflict-
// free when its 16 rows land on 16 distinct 16-byte slots of the 256-byte bank row.  With an odd number of slots per
// row (CK/8 + 1) that holds exactly when the column pitch in slots is congruent to TH modulo 16.
int conv_halo_col_bytes(const ConvTile& t, int TH, int stride) {
    const int KW = t.taps == 9 ? 3 : 1;
    const int THv = (TH - 1) * stride + KW;
    const int rs = conv_row_bytes(t.CK) / 16;
    int slots = THv * rs;
    while (slots % 16 != TH % 16) ++slots;
    return slots * 16;
}

int conv_max_halo_slots(const ConvTile& t) {
    const ConvInst* i = find_inst(t);
    return i ? i->ACH * 64 * i->NW / (t.CK / 8) : 0;
}
int conv_tile_threads(const ConvTile& t) {
    const ConvInst* i = find_inst(t);
    return i ? 64 * i->NW : 0;
}

size_t conv_lds_bytes(const ConvTile& t, const ConvParams& p) {
    const ConvInst* inst = find_inst(t);
    const int NW = inst ? inst->NW : 4;
    const int RS = conv_row_bytes(t.CK);
    const int KW = t.taps == 9 ? 3 : 1;
    const int TWv = (p.TW - 1) * p.stride + KW;
    const size_t a = (size_t)TWv * p.colb;
    const size_t g = (size_t)(p.C0 + p.C1) * 8 + (size_t)t.BN * 4;       // sGa, sGs, sBias
    const int TG = inst ? inst->TG : 1;
    const size_t w = (size_t)(TG == 9 ? 2 : 3) * TG * t.BN * RS;
    size_t main_bytes = w + 2 * a + g;
    // GroupNorm finalize scratch lives in the (not yet written) halo buffers: 2*Cin + 2*groups doubles
    const size_t gscratch = p.st0 ? w + (size_t)2 * (p.C0 + p.C1) * 8 : 0;
    size_t epi = (size_t)t.BM * (t.BN * 2 + 16) + (size_t)2 * (64 * NW / (t.BN / 8)) * t.BN * 4;
    if (inst && inst->KG == 2) epi = std::max(epi, (size_t)t.BM * t.BN * 4);      // k-group accumulator hand-over
    return std::max(std::max(main_bytes, gscratch), epi) + 512;    // + timeline scratch (ABLATE builds)
}

template <int NW, int BM, int BN, int WM, int WN, int KG, int CK, int TAPS, int TG, int ACH>
static int launch_inst(const ConvParams& p, int grid, size_t lds, hipStream_t stream) {
    auto kern = conv_igemm_kernel<NW, BM, BN, WM, WN, KG, CK, TAPS, TG, ACH>;
    static DynLdsLimit lds_limit;                // per device, thread safe
    RLDM_HIP_CHECK(lds_limit.ensure(reinterpret_cast<const void*>(kern), lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, stream, p);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_conv(const ConvTile& t, const ConvParams& p, hipStream_t stream) {
    RLDM_REQUIRE(p.Wout % p.TW == 0 && p.Hout % p.TH == 0, "conv: output size is not a multiple of the pixel tile");
    RLDM_REQUIRE(p.TW * p.TH <= t.BM, "conv: pixel tile larger than BM");
    RLDM_REQUIRE((p.C0 + p.C1) % t.CK == 0, "conv: input channels not a multiple of CK");
    RLDM_REQUIRE(p.C1 == 0 || p.C0 % t.CK == 0, "conv: concat boundary must be a multiple of the channel chunk");
    RLDM_REQUIRE((p.R0 + p.R1) % t.CK == 0 && (p.R1 == 0 || p.R0 % t.CK == 0), "conv: residual channels vs chunk size");
    RLDM_REQUIRE((p.R0 + p.R1) == 0 || (p.stride == 1 && p.up == 1), "conv: residual phase needs stride 1, no upsample");
    RLDM_REQUIRE(p.up == 1 || p.up == 2, "conv: upsample factor must be 1 or 2");
    RLDM_REQUIRE(p.Win * p.up >= 2 || t.taps == 1, "conv: azimuth extent too small for wrap-around");
    RLDM_REQUIRE(p.ksplit >= 1 && (p.ksplit == 1 || (p.slab && p.ticket)), "conv: split-K needs slab and ticket buffers");
    RLDM_REQUIRE(p.ksplit == 1 || (find_inst(t) && find_inst(t)->KG == 1), "conv: split-K is not combined with k-group instances");
    const ConvInst* inst = find_inst(t);
    RLDM_REQUIRE(inst != nullptr, "conv: no kernel instance for tile BM=" + std::to_string(t.BM) + " BN=" +
                                      std::to_string(t.BN) + " CK=" + std::to_string(t.CK) + " taps=" + std::to_string(t.taps));
    RLDM_REQUIRE(!p.st0 || (p.C0 + p.C1) <= 128 * inst->NW, "conv: GroupNorm prologue needs Cin <= 2 * threads");
    RLDM_REQUIRE(!p.st0 || p.gn_groups <= 64, "conv: GroupNorm prologue supports at most 64 groups");
    const int KW = t.taps == 9 ? 3 : 1;
    const int nslots = ((p.TW - 1) * p.stride + KW) * ((p.TH - 1) * p.stride + KW);
    RLDM_REQUIRE(nslots <= conv_max_halo_slots(t), "conv: halo larger than the instance's register staging capacity");
    RLDM_REQUIRE(p.colb % 16 == 0 && p.colb >= ((p.TH - 1) * p.stride + KW) * conv_row_bytes(t.CK), "conv: bad halo column pitch");
    const int grid = p.B * (p.Wout / p.TW) * (p.Hout / p.TH) * p.ntile_n * p.ksplit;
    const size_t lds = conv_lds_bytes(t, p);
    RLDM_REQUIRE(lds <= 160 * 1024, "conv: LDS footprint exceeds 160 KiB");
    ConvParams pp = p;
    pp.lds_total = (int)lds;
#define X(nw_, bm_, bn_, wm_, wn_, kg_, ck_, taps_, tg_, ach_)                            \
    if (t.BM == bm_ && t.BN == bn_ && t.CK == ck_ && t.taps == taps_)                     \
        return launch_inst<nw_, bm_, bn_, wm_, wn_, kg_, ck_, taps_, tg_, ach_>(pp, grid, lds, stream);
    RLDM_CONV_INSTANCES(X)
#undef X
    return 1;
}

}  // namespace rldm
