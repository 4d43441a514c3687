// Circular 3x3 / stride 1 convolution of a 64-channel tensor at full resolution with the WEIGHTS RESIDENT IN REGISTERS and a persistent
// loop over pixel tiles (round 4; the VAE decoder's 1024 x 64 level: vae/sgm/modules/diffusionmodules/model.py:93-125 ResnetBlock,
// :620-640 norm_out / conv_out).
//
// The same fused unit as conv_stream.hip -- GroupNorm + SiLU of the input on its way into LDS, conv 3x3 (wrap W / zero H), bias, identity
// residual, per-channel statistics of the rounded output for the next GroupNorm -- for the one shape where that kernel's division of labour
// does not fit: with 64 input channels a tile has ONE chunk, so its statistics fold, weight ring (74 KB per tile from the L2), halo round
// trip, GroupNorm + SiLU, 72 MFMAs per wave and epilogue run back to back, 22-40 k cycles per 128-pixel tile for 2.3 k cycles of MFMAs
// (171-297 us per conv against an HBM floor of 34 us).  Here
//   * a wave keeps the 36 weight fragments of its 32-channel tile (K = 9 taps x 64 channels = 36 k-steps of 16: 144 registers) for the
//     whole launch: no weight traffic and no LDS operand for the A side of the MFMAs, the K loop is 72 ds_read_b128 with immediate
//     offsets + 72 MFMAs and no address arithmetic;
//   * a workgroup (4 waves = 2 pixel halves x 2 channel tiles, two workgroups per CU) walks a run of consecutive 16 x 8 tiles of ONE image:
//     the statistics fold, bias and weights are per launch, the halo of tile t + 1 is requested before the K loop of tile t and normalised
//     into the second LDS buffer behind it, the epilogue's staging re-uses the buffer the K loop has just consumed (two barriers per tile);
//   * the identity residual enters through the accumulators' initial value (bias + x), loaded in the MFMA's own layout while the
//     accumulators are dead;
//   * the output statistics are accumulated over the run in registers: one partial per workgroup (32 per image at batch 16) instead of
//     one per tile (512) -- no gn_fold launch behind the conv.
#include "kernels.h"
#include "common.h"

namespace rldm {

namespace {
constexpr int kRwTW = 16, kRwTH = 8, kRwCK = 64;
constexpr int kRwRS = kRwCK * 2 + 16;                   // halo row stride (bytes)
constexpr int kRwColb = 1664;                           // conv_halo_col_bytes({CK 64, taps 9}, TH 8, stride 1): checked by the launcher
constexpr int kRwTWv = kRwTW + 2, kRwTHv = kRwTH + 2;
constexpr int kRwABytes = kRwTWv * kRwColb;             // one halo buffer
constexpr int kRwTeam = 256;                            // threads of a team (4 waves, one per SIMD)
constexpr int kRwPieces = kRwTWv * kRwTHv * (kRwCK / 8);          // 1440 16-byte pieces per halo
constexpr int kRwACH = (kRwPieces + kRwTeam - 1) / kRwTeam;       // 6 per thread
constexpr int kRwWRS = 32 * 2 + 16;                     // a wave's epilogue staging row (bytes): [64 pixels][32 channels] bf16
constexpr int kRwRawWave = kRwACH * 64 * 16;            // a wave's raw halo pieces [piece][lane]: the landing zone of its LDS-DMA prefetch, and -- once
                                                        // they are normalised -- the staging of its 64 x 32 output sub-tile (wave private: no barrier)
static_assert(64 * kRwWRS <= kRwRawWave, "the epilogue staging fits the consumed landing zone");
constexpr int kRwTeamBytes = kRwABytes + 4 * kRwRawWave;
}  // namespace

// A team = 4 waves (2 pixel halves x 2 channel tiles, one wave per SIMD): a wave owns 64 pixels x 32 channels of its team's tile.
// TEAMS = 2: one 8-wave workgroup per CU whose teams swap roles at workgroup barriers (strict ping-pong); TEAMS = 1: a team is a workgroup of
// its own, two per CU, free running
// WN = 32-channel tiles of the layer: 2 (64 output channels, bf16 [B][W][H][64] + statistics; a wave owns 64 pixels x 32 channels) | 1 (up to 4
// output channels as fp32 NCHW -- the decoder's conv_out; a wave owns 32 pixels x one padded tile)
template <int TEAMS, int WN>
__global__ void __launch_bounds__(kRwTeam * TEAMS, TEAMS == 1 ? 2 : 1) conv_regw_kernel(const ConvParams p, const int wg_per_image, const int tiles_per_team) {
    constexpr int MI = WN;
    constexpr int C8 = kRwCK / 8, RS = kRwRS, COLB = kRwColb, ACH = kRwACH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = TEAMS == 2 ? wave >> 2 : 0, tw4 = wave & 3;        // (waves w and w + 4 share a SIMD: one of each team)
    const int ttid = tid & (kRwTeam - 1);
    const int wn = tw4 % WN, wm = tw4 / WN;
    const int kh = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x / wg_per_image, part = blockIdx.x - b * wg_per_image;
    const int t_first = (part * TEAMS + team) * tiles_per_team;
    const unsigned long long ts_core0 = __builtin_amdgcn_s_memtime(), ts_real0 = __builtin_amdgcn_s_memrealtime();   // (tuning: the launch's clock)
    const int abl = p.exp >> 16;                        // (tuning: RLDM_RW_ABL -- 1 no K loop, 2 no staging arithmetic, 4 no output stores, 8 no halo loads, 16 start offset)

    unsigned char* const sA = smem + team * kRwTeamBytes;               // this team's halo (normalised), then its waves' landing zones
    unsigned char* const sRaw = sA + kRwABytes + tw4 * kRwRawWave;      // this wave's
    float* const sGa = reinterpret_cast<float*>(smem + TEAMS * kRwTeamBytes);   // 64
    float* const sGs = sGa + kRwCK;                                     // 64
    float* const sBias = sGs + kRwCK;                                   // 64
    float* const sS = sBias + 64;                                       // [8 waves][2][32]
    // the last three taps' 12 weight fragments of each channel tile live in LDS (24 KB), not in registers: with all 36 in registers the compiler
    // spilled some and reloaded them at the head of every K loop -- behind an s_waitcnt vmcnt(0) that also drained the halo prefetch; the
    // registers this frees hold the per-piece constants of the staging
    unsigned char* const sWt = reinterpret_cast<unsigned char*>(sS + 8 * 2 * 32);     // [2 channel tiles][12 k-steps][64 lanes][16 bytes]

    const int tiles_h = p.tiles_h;                                      // a power of two
    const int th_bits = 31 - __builtin_clz(tiles_h);
    const bool gn = p.st0 != nullptr;
    const bool has_res = WN == 2 && p.r0 != nullptr;

    // ---- halo pieces of this thread, tile independent: LDS offset within the halo, source offset relative to the tile's first pixel, and
    // the flags that make a piece a zero or wrap it around the image (one wave issues a VALU instruction every ~8 cycles, and in the V role it
    // is alone on its SIMD: the first version recomputed these per use and spent 400 of its 1100 instructions per tile on them)
    const int my_c8 = ttid % C8;
    int hoff[ACH];                                                      // LDS byte offset of the piece (column pitch COLB, row pitch RS); -1: no such piece
    int soff[ACH];                                                      // ((column - 1) * Hin + row - 1) * 128: source byte offset from the tile's pixel (0, 0)
    int pflag[ACH];                                                     // 1 top row, 2 bottom row, 4 first column, 8 last column, 16 no such piece
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        const int q = ttid + i * kRwTeam;
        const int slot = q / C8;
        const int vwl = (slot * 6554) >> 16, vhl = slot - vwl * kRwTHv;             // slot / 10 for slot < 16384
        const bool none = q >= kRwPieces;
        hoff[i] = none ? -1 : vwl * COLB + vhl * RS + my_c8 * 16;
        soff[i] = ((vwl - 1) * p.Hin + (vhl - 1)) * (kRwCK * 2);
        pflag[i] = none ? 16 : ((vhl == 0 ? 1 : 0) | (vhl == kRwTHv - 1 ? 2 : 0) | (vwl == 0 ? 4 : 0) | (vwl == kRwTWv - 1 ? 8 : 0));
    }
    const unsigned char* const xbase = reinterpret_cast<const unsigned char*>(p.x0) + my_c8 * 16;
    const int last_tw = (p.Win / kRwTW) - 1;
    const int wrap_bytes = p.Win * p.Hin * (kRwCK * 2);
    // per tile (scalar): which flags make a piece a zero / wrap it, and the byte offset of the tile's pixel (0, 0)
    struct TileS { int pad_mask, wrap_lo, wrap_hi; unsigned base; };
    auto tile_scalars = [&](int mt) __attribute__((always_inline)) {
        const int tw = mt >> th_bits, th = mt & (tiles_h - 1);
        TileS t;
        t.pad_mask = 16 | (th == 0 ? 1 : 0) | (th == tiles_h - 1 ? 2 : 0);
        t.wrap_lo = tw == 0 ? wrap_bytes : 0;                           // the column left of azimuth 0 is azimuth W - 1
        t.wrap_hi = tw == last_tw ? wrap_bytes : 0;
        t.base = (unsigned)(((b * p.Win + tw * kRwTW) * p.Hin + th * kRwTH) * (kRwCK * 2));
        return t;
    };
    auto piece_src = [&](const TileS& t, int i) __attribute__((always_inline)) {      // byte offset from xbase (a valid address for zero pieces too)
        int o = soff[i] + ((pflag[i] & 4) ? t.wrap_lo : 0) - ((pflag[i] & 8) ? t.wrap_hi : 0);
        o = (pflag[i] & t.pad_mask) ? 0 : o;
        return t.base + (unsigned)o;
    };
    // GroupNorm + SiLU of one 16-byte piece, stored at its halo position (no branch: a zero piece is normalised garbage replaced at the end)
    float4 ga0, ga1, gs0, gs1;
    auto put_piece = [&](int i, uint4 v, bool pad) __attribute__((always_inline)) {
        if (gn && !(abl & 2)) {
            float f0 = bf16lo(v.x) * ga0.x + gs0.x, f1 = bf16hi(v.x) * ga0.y + gs0.y;
            float f2 = bf16lo(v.y) * ga0.z + gs0.z, f3 = bf16hi(v.y) * ga0.w + gs0.w;
            float f4 = bf16lo(v.z) * ga1.x + gs1.x, f5 = bf16hi(v.z) * ga1.y + gs1.y;
            float f6 = bf16lo(v.w) * ga1.z + gs1.z, f7 = bf16hi(v.w) * ga1.w + gs1.w;
            if (p.silu) silu_x8(f0, f1, f2, f3, f4, f5, f6, f7);
            v.x = pack_bf16x2(f0, f1); v.y = pack_bf16x2(f2, f3);
            v.z = pack_bf16x2(f4, f5); v.w = pack_bf16x2(f6, f7);
        }
        v.x = pad ? 0u : v.x; v.y = pad ? 0u : v.y; v.z = pad ? 0u : v.z; v.w = pad ? 0u : v.w;
        if (hoff[i] >= 0) *reinterpret_cast<uint4*>(sA + hoff[i]) = v;
    };
    auto load_affine = [&]() __attribute__((always_inline)) {
        if (gn) {
            ga0 = *reinterpret_cast<const float4*>(sGa + my_c8 * 8);
            ga1 = *reinterpret_cast<const float4*>(sGa + my_c8 * 8 + 4);
            gs0 = *reinterpret_cast<const float4*>(sGs + my_c8 * 8);
            gs1 = *reinterpret_cast<const float4*>(sGs + my_c8 * 8 + 4);
        }
    };
    // Prefetch of a halo WITHOUT registers (LDS-DMA: global_load_lds_dwordx4 writes M0 + lane * 16): six pieces in flight across the K role
    // (24 registers) made the compiler spill weight fragments into the K loop.  The compiler does not see these loads (inline asm): the
    // consumer waits with its own s_waitcnt vmcnt(0).
    const unsigned raw_lds = (unsigned)(uintptr_t)sRaw;                 // (wave-uniform; low 32 bits of a flat LDS address = the LDS offset)
    auto dma_halo = [&](int mt) __attribute__((always_inline)) {
        const TileS t = tile_scalars(mt);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the zone's previous contents (raw pieces, then the output sub-tile) are consumed
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const unsigned char* src = xbase + piece_src(t, i);
            const unsigned dst = __builtin_amdgcn_readfirstlane(raw_lds + i * (64 * 16));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };
    auto stage_from_raw = [&](int mt) __attribute__((always_inline)) {      // tile mt's pieces: raw (LDS) -> normalised halo
        const TileS t = tile_scalars(mt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the DMA of this tile (requested a whole K phase ago) has landed
        load_affine();
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const uint4 v = *reinterpret_cast<const uint4*>(sRaw + (i * 64 + lane) * 16);
            put_piece(i, v, (pflag[i] & t.pad_mask) != 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the raw pieces are consumed: the zone becomes the epilogue's staging
    };

    // ---- the team's first halo (through registers), then the wave's 36 weight fragments ---------------------------------------------------
    const TileS tile0 = tile_scalars(t_first);
    uint4 areg[ACH];
#pragma unroll
    for (int i = 0; i < ACH; ++i) areg[i] = *reinterpret_cast<const uint4*>(xbase + piece_src(tile0, i));
    const int nsteps = 36 + (p.R0 / kRwCK) * 4;                         // (the packed stream carries the identity's k-steps behind the taps: skipped)
    const unsigned char* const wsrc = reinterpret_cast<const unsigned char*>(p.wpk) + (size_t)wn * nsteps * 1024 + lane * 16;

    // ---- GroupNorm: channel `tid` folds its partials; its group's (mean, rstd) through LDS; affine (a, s) per channel ----------------
    if (gn) {
        double S = 0.0, SS = 0.0;
        float gamma = 0.f, beta = 0.f;
        if (tid < kRwCK) {
            const float2* src = p.st0 + (size_t)b * p.P0 * kRwCK + tid;
            gamma = p.gn_gamma[tid];
            beta = p.gn_beta[tid];
            int q = 0;
            for (; q + 8 <= p.P0; q += 8) {
                float2 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(q + j) * kRwCK];
#pragma unroll
                for (int j = 0; j < 8; ++j) { S += (double)v[j].x; SS += (double)v[j].y; }
            }
            for (; q < p.P0; ++q) {
                const float2 v = src[(size_t)q * kRwCK];
                S += (double)v.x;
                SS += (double)v.y;
            }
        }
        double* sD = reinterpret_cast<double*>(smem);                   // scratch [2][64] (the halos are not written yet)
        if (tid < kRwCK) { sD[tid] = S; sD[kRwCK + tid] = SS; }
        __syncthreads();
        if (tid < kRwCK) {
            const int cpg = kRwCK / p.gn_groups;
            const int g0 = (tid / cpg) * cpg;
            double Sg = 0.0, SSg = 0.0;
            for (int i = 0; i < cpg; ++i) { Sg += sD[g0 + i]; SSg += sD[kRwCK + g0 + i]; }
            const double inv_n = (double)p.gn_inv_n;
            const double mean = Sg * inv_n;
            double var = SSg * inv_n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const float a = gamma * __builtin_amdgcn_rsqf((float)var + p.gn_eps);
            sGa[tid] = a;
            sGs[tid] = beta - (float)mean * a;
        }
    }
    if (tid < 32 * WN) sBias[tid] = p.bias[tid];
    __syncthreads();                                                    // affine + bias visible, scratch consumed
    load_affine();
#pragma unroll
    for (int i = 0; i < ACH; ++i) put_piece(i, areg[i], (pflag[i] & tile0.pad_mask) != 0);
    if (tiles_per_team > 1 && !(abl & 8)) dma_halo(t_first + 1);

    constexpr int NWR = TEAMS == 2 ? 24 : 28;                           // fragments in registers; the last taps' from LDS (24 / 16 KB)
    bf16x8 wr[NWR];
#pragma unroll
    for (int i = 0; i < NWR; ++i) wr[i] = *reinterpret_cast<const bf16x8*>(wsrc + i * 1024);
    if (wave < WN) {                                                    // (waves 0 and 1 are channel tiles 0 and 1 of team 0)
#pragma unroll
        for (int i = NWR; i < 36; ++i)
            *reinterpret_cast<bf16x8*>(sWt + ((wn * (36 - NWR) + (i - NWR)) * 64 + lane) * 16) = *reinterpret_cast<const bf16x8*>(wsrc + i * 1024);
    }

    const unsigned char* const rbase = reinterpret_cast<const unsigned char*>(p.r0) + (wn * 32 + 4 * kh) * 2;
    // epilogue roles inside the wave's own 64 x 32 sub-tile: 16-byte output pieces (pixel lane / 4 + 16 k, channels 8 (lane % 4) ..) and
    // statistics (channel pair lane % 16, pixel group lane / 16 of 16)
    const int ep = lane >> 2, ec = lane & 3;
    const int cp = lane & 15, pg = lane >> 4;
    const unsigned lane_yoff = (unsigned)(((wm * 8 + (ep >> 3)) * p.Hout + (ep & 7)) * p.y_ld + wn * 32 + ec * 8) * 2u;
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
    auto phase_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    // (every weight fragment is USED here: the compiler's s_waitcnt for the prologue's loads then sits in front of the loop -- left to the first
    //  MFMAs it would be a counted wait inside the loop that, the counter being in order, also waits for whatever was requested last: the halo
    //  prefetch)
#pragma unroll
    for (int i = 0; i < NWR; ++i) asm volatile("" :: "v"(wr[i]));
    __syncthreads();                                                    // both teams' first halos written

    // ---- the two teams alternate: one runs its K loop (matrix pipe) while the other normalises its next halo, rounds / stores its previous
    // tile and requests the halo after next (VALU, LDS, memory); they swap at every barrier.  Team 1 starts one phase late.
    if (TEAMS == 2 && team == 1) phase_barrier();
    if (TEAMS == 1 && (abl & 16) && 2 * blockIdx.x >= gridDim.x) {      // (tuning: the second workgroup of a CU starts half a tile late)
        for (int k = 0; k < 2; ++k) __builtin_amdgcn_s_sleep(32);
    }
    for (int i = 0; i < tiles_per_team; ++i) {
        const int mt = t_first + i;
        // ================= K role: accumulators = bias, 9 taps x 4 k-steps, + x =================
        f32x16 acc[MI];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const float4 bv = *reinterpret_cast<const float4*>(sBias + wn * 32 + 8 * r4 + 4 * kh);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                acc[mi][r4 * 4 + 0] = bv.x; acc[mi][r4 * 4 + 1] = bv.y;
                acc[mi][r4 * 4 + 2] = bv.z; acc[mi][r4 * 4 + 3] = bv.w;
            }
        }
        // the identity residual in the MFMA's own layout (8 bytes = 4 channels of a lane's pixel): requested here, added behind the K loop --
        // the only compiler-visible loads of the loop, so its s_waitcnt for them sits at the END of a K phase, a whole phase after this team's
        // last halo request (an earlier wait would also wait for that: the counter is in order)
        uint2 rreg[MI][4];
        if (has_res) {
            const int tw = mt >> th_bits, th = mt & (tiles_h - 1);
            const int pix0 = (b * p.Wout + tw * kRwTW) * p.Hout + th * kRwTH;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int pidx = wm * (32 * MI) + mi * 32 + l31;
                const unsigned off = (unsigned)((pix0 + (pidx >> 3) * p.Hout + (pidx & 7)) * (64 * 2));
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) rreg[mi][r4] = *reinterpret_cast<const uint2*>(rbase + off + r4 * 16);
            }
        }
        {
            // every address an immediate offset from the lane's pixel; the pixel fragment of step s + 2 is requested right behind the MFMA
            // that frees its register (two fragment sets)
            const unsigned char* xp[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int pidx = wm * (32 * MI) + mi * 32 + l31;
                xp[mi] = sA + (pidx >> 3) * COLB + (pidx & 7) * RS + kh * 16;
            }
            auto xaddr = [&](int mi, int step) __attribute__((always_inline)) {
                const int tap = step / 4, ks = step % 4;
                return reinterpret_cast<const bf16x8*>(xp[mi] + (tap / 3) * COLB + (tap % 3) * RS + ks * 32);
            };
            bf16x8 xr[2][MI];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) xr[j][mi] = *xaddr(mi, j);
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 wt[4];                                               // the LDS-resident fragments, each read 4 k-steps ahead of its MFMAs
            auto wt_read = [&](int step) __attribute__((always_inline)) {
                return *reinterpret_cast<const bf16x8*>(sWt + ((wn * (36 - NWR) + (step - NWR)) * 64 + lane) * 16);
            };
            if (!(abl & 1))
#pragma unroll
            for (int step = 0; step < 36; ++step) {
                bf16x8 wf;
                if (step < NWR) wf = wr[step];
                else wf = wt[step % 4];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xr[step % 2][mi], acc[mi], 0, 0, 0);
                    if (step + 2 < 36) xr[step % 2][mi] = *xaddr(mi, step + 2);
                }
                if (step + 4 >= NWR && step + 4 < 36) wt[step % 4] = wt_read(step + 4);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (step + 2 < 36) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                if (step + 4 >= NWR && step + 4 < 36) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (has_res) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    acc[mi][r4 * 4 + 0] += bf16lo(rreg[mi][r4].x); acc[mi][r4 * 4 + 1] += bf16hi(rreg[mi][r4].x);
                    acc[mi][r4 * 4 + 2] += bf16lo(rreg[mi][r4].y); acc[mi][r4 * 4 + 3] += bf16hi(rreg[mi][r4].y);
                }
        }
        phase_barrier();                                                // roles swap: the halo is consumed by every wave of the team
        // ================= V role: tile i + 1's halo, tile i's epilogue, tile i + 2's request =================
        if (i + 1 < tiles_per_team) stage_from_raw(mt + 1);             // over the halo the K loop has consumed
        if constexpr (WN == 1) {
            // fp32 NCHW straight from the registers: lanes 0..31 hold channels 0..3 of their pixel
            if (kh == 0 && !(abl & 4)) {
                const int tw = mt >> th_bits, th = mt & (tiles_h - 1);
                const int pidx = wm * 32 + l31;
                const int w = tw * kRwTW + (pidx >> 3), h = th * kRwTH + (pidx & 7);
#pragma unroll
                for (int ch = 0; ch < 4; ++ch)
                    if (ch < p.N) p.y_nchw[((size_t)(b * p.N + ch) * p.Wout + w) * p.Hout + h] = acc[0][ch];
            }
        } else {
        // part 1: the wave's rounded 64 x 32 sub-tile -> its own landing zone (consumed just above; wave private: no barrier)
        unsigned char* const sE = sRaw;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                uint2 o;
                o.x = pack_bf16x2(acc[mi][r4 * 4 + 0], acc[mi][r4 * 4 + 1]);
                o.y = pack_bf16x2(acc[mi][r4 * 4 + 2], acc[mi][r4 * 4 + 3]);
                *reinterpret_cast<uint2*>(sE + (mi * 32 + l31) * kRwWRS + (8 * r4 + 4 * kh) * 2) = o;
            }
        // part 2: 16-byte stores (4 lanes = the 64 bytes of a pixel's 32 channels), statistics of the rounded values
        {
            // (32-bit byte offsets from the scalar base: per-store 64-bit pointers are loop invariants the compiler would hoist into -- and
            //  spill from -- the registers the weights need)
            const int tw = mt >> th_bits, th = mt & (tiles_h - 1);
            const unsigned tile_off = (unsigned)(((b * p.Wout + tw * kRwTW) * p.Hout + th * kRwTH) * p.y_ld) * 2u;
            const unsigned kstep = (unsigned)(2 * p.Hout * p.y_ld) * 2u;                     // 16 pixels = 2 columns of the tile
            unsigned char* const ybase = reinterpret_cast<unsigned char*>(p.y);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint4 v = *reinterpret_cast<const uint4*>(sE + (ep + 16 * k) * kRwWRS + ec * 16);
                if (!(abl & 4)) *reinterpret_cast<uint4*>(ybase + (tile_off + lane_yoff + k * kstep)) = v;
            }
            if (p.y_stats) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t w2 = *reinterpret_cast<const uint32_t*>(sE + (pg * 16 + j) * kRwWRS + cp * 4);
                    const float a0 = bf16lo(w2), a1 = bf16hi(w2);
                    s0 += a0; s1 += a1;
                    q0 += a0 * a0; q1 += a1 * a1;
                }
            }
        }
        }
        if (i + 2 < tiles_per_team && !(abl & 8)) dma_halo(mt + 2);
        phase_barrier();
    }
    if (TEAMS == 2 && team == 0) phase_barrier();
    if (p.ts && blockIdx.x == 0 && tid == 0) {          // rldm_debug_timestamps: workgroup 0's lifetime on the core clock and on the 100 MHz counter
        p.ts[0] = ts_core0; p.ts[1] = __builtin_amdgcn_s_memtime();
        p.ts[2] = ts_real0; p.ts[3] = __builtin_amdgcn_s_memrealtime();
    }
    // ---- the workgroup's statistics: one partial (both teams' runs) ---------------------------------------------------------------------
    if (WN == 2 && p.y_stats) {
        s0 += __shfl_xor(s0, 16); s1 += __shfl_xor(s1, 16); q0 += __shfl_xor(q0, 16); q1 += __shfl_xor(q1, 16);
        s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32); q0 += __shfl_xor(q0, 32); q1 += __shfl_xor(q1, 32);
        if (lane < 16) {
            *reinterpret_cast<float2*>(sS + (wave * 2 + 0) * 32 + cp * 2) = make_float2(s0, s1);
            *reinterpret_cast<float2*>(sS + (wave * 2 + 1) * 32 + cp * 2) = make_float2(q0, q1);
        }
        __syncthreads();
        if (tid < 128) {
            const int kind = tid / 64, c = tid - kind * 64;
            float S = 0.f;
#pragma unroll
            for (int w = 0; w < 4 * TEAMS; ++w)
                if ((w & 3) % WN == c / 32) S += sS[(w * 2 + kind) * 32 + (c & 31)];
            reinterpret_cast<float*>(p.y_stats + ((size_t)b * wg_per_image + part) * p.N + c)[kind] = S;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv_c16_kernel: the network's INPUT layer -- 3x3 / stride 1 over 16 (padded) input channels, 128 output channels per workgroup
// (UNet conv_in: 5 -> 128 over cat[x_t, positional encoding], ldm/inference.py:120-138 / diffusers UNet2DModel.conv_in; the decoder's
// conv_in: 4 -> 256).  K = 9 taps x 16 channels = 9 k-steps, no channel chunks, no weight ring, no barrier per tap: a
// wave keeps the 9 tap fragments of its 32-channel tile in registers, and the layer is a load, 36 MFMAs per wave and a store.  It ran on the generic kernel
// (one barrier per tap, 256-pixel tiles, 256 workgroups) at 20 us for 0.75 GFLOP; it is also the launch that advances the sampler's step
// index (ConvParams::step_inc: nothing in it reads the index).
namespace {
constexpr int kC16Colb = 10 * 32 + 16;                  // halo column pitch (bytes): 10 rows of 32 bytes + a slot (bank spread)
constexpr int kC16ERS = 128 * 2 + 16;                   // epilogue staging row: [pixel][128 channels] bf16
}  // namespace

__global__ void __launch_bounds__(256, 2) conv_c16_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);            // the wave's 32-channel tile of the group's 128
    const int kh = lane >> 5, l31 = lane & 31;
    int ng, mt, b;
    {
        const int gx = p.ntile_n, gy = p.tiles_img;                      // (128-channel groups, pixel tiles)
        const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const int rid = xcd_remap(lin, gx * gy * p.B);
        const int q = rid / gx;
        ng = rid - q * gx;
        b = q / gy;
        mt = q - b * gy;
    }
    if (p.step_inc && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) *p.step_inc += 1;
    const int tiles_h = p.tiles_h;
    const int tw = mt >> (31 - __builtin_clz(tiles_h)), th = mt & (tiles_h - 1);
    const int w0 = tw * 16, h0 = th * 8;
    unsigned char* const sA = smem;                                     // halo: 18 columns
    unsigned char* const sE = smem + 18 * kC16Colb;                     // epilogue staging
    float* const sS = reinterpret_cast<float*>(sE + 128 * kC16ERS);     // [4 waves][2][128]

    // ---- halo: 18 x 10 pixels x two 16-byte pieces, requested first ------------------------------------------------------------------
    uint4 hv[2];
    int hdst[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + i * 256;
        const int slot = q >> 1, c8 = q & 1;
        const int vwl = (slot * 6554) >> 16, vhl = slot - vwl * 10;
        const int vh = h0 - 1 + vhl;
        int vw = w0 - 1 + vwl;
        vw = vw < 0 ? vw + p.Win : (vw >= p.Win ? vw - p.Win : vw);
        const bool ok = q < 360 && vh >= 0 && vh < p.Hin;
        hdst[i] = q < 360 ? vwl * kC16Colb + vhl * 32 + c8 * 16 : -1;
        hv[i] = make_uint4(0u, 0u, 0u, 0u);
        if (ok) hv[i] = *reinterpret_cast<const uint4*>(p.x0 + ((size_t)(b * p.Win + vw) * p.Hin + vh) * 16 + c8 * 8);
    }
    // ---- the wave's weights: the 9 tap fragments of ITS 32-channel tile (wave = channel tile: a fragment is fetched by one wave of the
    // workgroup, 9 KB per wave; the first version gave every wave all 36 fragments of the 128 channels and a quarter of the pixels -- 4 x the
    // weight traffic for a quarter of the LDS reads, 10.1 against 9.x us) --------------------------------------------------------------------
    const int nt = wm;
    const unsigned char* const wsrc = reinterpret_cast<const unsigned char*>(p.wpk) + (size_t)(ng * 4 + nt) * (9 * 1024) + lane * 16;
    bf16x8 wr[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wr[i] = *reinterpret_cast<const bf16x8*>(wsrc + i * 1024);
    f32x16 acc[4];                                                      // four 32-pixel fragments of the tile
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + ng * 128 + nt * 32 + 8 * r4 + 4 * kh);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            acc[mi][r4 * 4 + 0] = bv.x; acc[mi][r4 * 4 + 1] = bv.y; acc[mi][r4 * 4 + 2] = bv.z; acc[mi][r4 * 4 + 3] = bv.w;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
        if (hdst[i] >= 0) *reinterpret_cast<uint4*>(sA + hdst[i]) = hv[i];
    __syncthreads();
    // ---- 9 taps x 4 pixel fragments ----------------------------------------------------------------------------------------------------
    const unsigned char* xp[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int pidx = mi * 32 + l31;
        xp[mi] = sA + (pidx >> 3) * kC16Colb + (pidx & 7) * 32 + kh * 16;
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xp[mi] + (tap / 3) * kC16Colb + (tap % 3) * 32);
            acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[tap], xf, acc[mi], 0, 0, 0);
        }
    }
    // ---- epilogue: bf16 -> LDS [pixel][channel] -> 16-byte stores; statistics of the rounded tile --------------------------------------
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            uint2 o;
            o.x = pack_bf16x2(acc[mi][r4 * 4 + 0], acc[mi][r4 * 4 + 1]);
            o.y = pack_bf16x2(acc[mi][r4 * 4 + 2], acc[mi][r4 * 4 + 3]);
            *reinterpret_cast<uint2*>(sE + (mi * 32 + l31) * kC16ERS + (nt * 32 + 8 * r4 + 4 * kh) * 2) = o;
        }
    __syncthreads();
    {
        const int g = tid >> 4, c8 = tid & 15;                          // 16 pixels per pass, 8 passes
        bf16_t* yp = p.y + (((size_t)b * p.Wout + w0 + (g >> 3)) * p.Hout + h0 + (g & 7)) * p.y_ld + ng * 128 + c8 * 8;
        const size_t ystep = (size_t)2 * p.Hout * p.y_ld;               // 16 pixels = 2 columns of the tile
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            *reinterpret_cast<uint4*>(yp) = *reinterpret_cast<const uint4*>(sE + (g + 16 * i) * kC16ERS + c8 * 16);
            yp += ystep;
        }
    }
    if (p.y_stats) {
        const int cp = tid & 63, pg = tid >> 6;                         // channel pair, pixel group of 32
        float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll 8
        for (int j = 0; j < 32; ++j) {
            const uint32_t w2 = *reinterpret_cast<const uint32_t*>(sE + (pg * 32 + j) * kC16ERS + cp * 4);
            const float a0 = bf16lo(w2), a1 = bf16hi(w2);
            s0 += a0; s1 += a1;
            q0 += a0 * a0; q1 += a1 * a1;
        }
        *reinterpret_cast<float2*>(sS + (pg * 2 + 0) * 128 + cp * 2) = make_float2(s0, s1);
        *reinterpret_cast<float2*>(sS + (pg * 2 + 1) * 128 + cp * 2) = make_float2(q0, q1);
        __syncthreads();
        {
            const int kind = tid >> 7, c = tid & 127;
            float S = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) S += sS[(w * 2 + kind) * 128 + c];
            reinterpret_cast<float*>(p.y_stats + ((size_t)b * p.tiles_img + mt) * p.N + ng * 128 + c)[kind] = S;
        }
    }
}

size_t conv_c16_lds_bytes() { return 18 * kC16Colb + 128 * kC16ERS + 4 * 2 * 128 * sizeof(float); }

bool conv_c16_supported(const ConvParams& p) {
    if (p.C0 != 16 || p.C1 != 0 || p.R0 != 0 || p.R1 != 0 || p.N % 128 != 0 || p.up != 1 || p.stride != 1 || p.pad_lo != 1) return false;
    if (p.st0 || p.temb || p.y_nchw || p.ksplit > 1 || p.TW != 16 || p.TH != 8) return false;
    if (p.Win != p.Wout || p.Hin != p.Hout || p.Wout % 16 != 0 || p.Hout % 8 != 0 || (p.tiles_h & (p.tiles_h - 1)) != 0) return false;
    return p.B <= 65535 && p.tiles_img <= 65535;
}

int launch_conv_c16(const ConvParams& p, hipStream_t stream) {
    RLDM_REQUIRE(conv_c16_supported(p), "conv_c16: unsupported shape");
    const size_t lds = conv_c16_lds_bytes();
    auto kern = conv_c16_kernel;
    static DynLdsLimit lds_limit;
    RLDM_HIP_CHECK(lds_limit.ensure(reinterpret_cast<const void*>(kern), lds));
    hipLaunchKernelGGL(kern, dim3(p.N / 128, p.tiles_img, p.B), dim3(256), lds, stream, p);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv_o4_kernel: the UNet's OUTPUT layer -- GroupNorm + SiLU -> 3x3 over 128 channels -> <= 4 channels as fp32 NCHW, with the sampler's
// scheduler step in the epilogue (diffusers UNet2DModel.conv_norm_out / conv_act / conv_out; DDIMScheduler.step / DDPMScheduler.step as in
// conv_igemm.hip's epilogue).  On the generic kernel (256-pixel tiles x one 32-channel tile, 4 k-groups) it was 26-28 us of statistics
// round trip, 48 normalised elements per thread and chunk with nothing to hide under, and a four-pass epilogue.  Here a workgroup is 4 waves
// on a 16 x 8 tile and a wave is a K-GROUP: it owns 32 of the 128 input channels for all 128 pixels -- it normalises ITS channels of the halo
// into its own LDS region (no barrier between staging and K loop), keeps its 18 weight fragments in registers, runs 72 MFMAs, and hands
// 4 floats per pixel to the reduction; 128 threads then own one pixel each for the scheduler arithmetic.
namespace {
constexpr int kO4RS = 32 * 2 + 16;                      // a wave's halo row: 32 channels + a slot
constexpr int kO4Colb = 10 * kO4RS + 16;                // ... column pitch
constexpr int kO4Wave = 18 * kO4Colb;                   // ... region
}  // namespace

__global__ void __launch_bounds__(256, 2) conv_o4_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int kg = __builtin_amdgcn_readfirstlane(tid >> 6);            // the wave's 32 input channels: [32 kg, 32 kg + 32)
    const int kh = lane >> 5, l31 = lane & 31;
    int mt, b;
    {
        const int gy = p.tiles_img;
        const int rid = xcd_remap(blockIdx.x + gy * blockIdx.y, gy * p.B);
        b = rid / gy;
        mt = rid - b * gy;
    }
    const int tiles_h = p.tiles_h;
    const int tw = mt >> (31 - __builtin_clz(tiles_h)), th = mt & (tiles_h - 1);
    const int w0 = tw * 16, h0 = th * 8;
    unsigned char* const sW = smem + kg * kO4Wave;                      // this wave's halo
    float* const sGa = reinterpret_cast<float*>(smem + 4 * kO4Wave);    // [128]
    float* const sGs = sGa + 128;
    double* const sD = reinterpret_cast<double*>(sGs + 128);            // [2][128] statistics scratch
    float4* const sP = reinterpret_cast<float4*>(sD + 256);             // [4 k-groups][128 pixels] partial outputs (4 channels)

    // ---- requests first: the statistics partials of channel `tid`, the wave's halo pieces (12 per lane), its 18 weight fragments ---------
    const bool gn = p.st0 != nullptr;
    double S = 0.0, SS = 0.0;
    float gamma = 0.f, beta = 0.f;
    if (gn && tid < 128) {
        const float2* src = p.st0 + (size_t)b * p.P0 * 128 + tid;
        gamma = p.gn_gamma[tid];
        beta = p.gn_beta[tid];
        int q = 0;
        for (; q + 16 <= p.P0; q += 16) {
            float2 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = src[(size_t)(q + j) * 128];
#pragma unroll
            for (int j = 0; j < 16; ++j) { S += (double)v[j].x; SS += (double)v[j].y; }
        }
        for (; q < p.P0; ++q) {
            const float2 v = src[(size_t)q * 128];
            S += (double)v.x;
            SS += (double)v.y;
        }
    }
    constexpr int NP = 12;                                              // 180 pixels x 4 pieces of 8 channels = 720 = 11.25 per lane
    uint4 hv[NP];
    const int c8 = lane & 3;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int slot = (lane >> 2) + i * 16;                          // halo pixel
        const int vwl = (slot * 6554) >> 16, vhl = slot - vwl * 10;
        const int vh = h0 - 1 + vhl;
        int vw = w0 - 1 + vwl;
        vw = vw < 0 ? vw + p.Win : (vw >= p.Win ? vw - p.Win : vw);
        const bool ok = slot < 180 && vh >= 0 && vh < p.Hin;
        hv[i] = make_uint4(0u, 0u, 0u, 0u);
        if (ok) hv[i] = *reinterpret_cast<const uint4*>(p.x0 + ((size_t)(b * p.Win + vw) * p.Hin + vh) * 128 + kg * 32 + c8 * 8);
    }
    // (stream-packed with 2 k-groups: [tile 0][k-group][2 chunks x 9 taps x 2 k-steps] -- channels 32 kg .. are chunk kg / 2, k-group kg % 2)
    const unsigned char* const wsrc = reinterpret_cast<const unsigned char*>(p.wpk) + ((size_t)(kg & 1) * 36 + (kg >> 1) * 18) * 1024 + lane * 16;
    bf16x8 wr[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) wr[i] = *reinterpret_cast<const bf16x8*>(wsrc + i * 1024);

    // ---- GroupNorm fold -> affine per channel ----------------------------------------------------------------------------------------
    if (gn) {
        if (tid < 128) { sD[tid] = S; sD[128 + tid] = SS; }
        __syncthreads();
        if (tid < 128) {
            const int cpg = 128 / p.gn_groups;
            const int g0 = (tid / cpg) * cpg;
            double Sg = 0.0, SSg = 0.0;
            for (int i = 0; i < cpg; ++i) { Sg += sD[g0 + i]; SSg += sD[128 + g0 + i]; }
            const double inv_n = (double)p.gn_inv_n;
            const double mean = Sg * inv_n;
            double var = SSg * inv_n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const float a = gamma * __builtin_amdgcn_rsqf((float)var + p.gn_eps);
            sGa[tid] = a;
            sGs[tid] = beta - (float)mean * a;
        }
        __syncthreads();
    }
    // ---- the wave normalises its 32 channels of the halo into its own region (wave private: no workgroup barrier) ----------------------
    {
        float4 ga0, ga1, gs0, gs1;
        if (gn) {
            ga0 = *reinterpret_cast<const float4*>(sGa + kg * 32 + c8 * 8);
            ga1 = *reinterpret_cast<const float4*>(sGa + kg * 32 + c8 * 8 + 4);
            gs0 = *reinterpret_cast<const float4*>(sGs + kg * 32 + c8 * 8);
            gs1 = *reinterpret_cast<const float4*>(sGs + kg * 32 + c8 * 8 + 4);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int slot = (lane >> 2) + i * 16;
            const int vwl = (slot * 6554) >> 16, vhl = slot - vwl * 10;
            const int vh = h0 - 1 + vhl;
            const bool pad = vh < 0 || vh >= p.Hin;
            uint4 v = hv[i];
            if (gn) {
                float f0 = bf16lo(v.x) * ga0.x + gs0.x, f1 = bf16hi(v.x) * ga0.y + gs0.y;
                float f2 = bf16lo(v.y) * ga0.z + gs0.z, f3 = bf16hi(v.y) * ga0.w + gs0.w;
                float f4 = bf16lo(v.z) * ga1.x + gs1.x, f5 = bf16hi(v.z) * ga1.y + gs1.y;
                float f6 = bf16lo(v.w) * ga1.z + gs1.z, f7 = bf16hi(v.w) * ga1.w + gs1.w;
                if (p.silu) silu_x8(f0, f1, f2, f3, f4, f5, f6, f7);
                v.x = pack_bf16x2(f0, f1); v.y = pack_bf16x2(f2, f3);
                v.z = pack_bf16x2(f4, f5); v.w = pack_bf16x2(f6, f7);
            }
            v.x = pad ? 0u : v.x; v.y = pad ? 0u : v.y; v.z = pad ? 0u : v.z; v.w = pad ? 0u : v.w;
            if (slot < 180) *reinterpret_cast<uint4*>(sW + vwl * kO4Colb + vhl * kO4RS + c8 * 16) = v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // (the wave's own stores: visible to the wave's own reads)
    // ---- K loop: 9 taps x 2 k-steps x 4 pixel fragments, the wave's slice of K ---------------------------------------------------------
    f32x16 acc[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
    {
        const unsigned char* xp[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int pidx = mi * 32 + l31;
            xp[mi] = sW + (pidx >> 3) * kO4Colb + (pidx & 7) * kO4RS + kh * 16;
        }
#pragma unroll
        for (int step = 0; step < 18; ++step) {
            const int tap = step >> 1, ks = step & 1;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xp[mi] + (tap / 3) * kO4Colb + (tap % 3) * kO4RS + ks * 32);
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[step], xf, acc[mi], 0, 0, 0);
            }
        }
    }
    // ---- the k-groups' partial outputs (channels 0..3: the first register quad of lanes 0..31) -> LDS -> one thread per pixel ------------
    if (kh == 0) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) sP[kg * 128 + mi * 32 + l31] = make_float4(acc[mi][0], acc[mi][1], acc[mi][2], acc[mi][3]);
    }
    __syncthreads();
    if (tid < 128) {
        float e[4] = {p.bias[0], p.bias[1], p.bias[2], p.bias[3]};      // (bias padded to 32)
#pragma unroll
        for (int g = 0; g < 4; ++g) {                                   // fixed order: deterministic
            const float4 v = sP[g * 128 + tid];
            e[0] += v.x; e[1] += v.y; e[2] += v.z; e[3] += v.w;
        }
        // fused scheduler step (sched_step_kernel's arithmetic on the value just computed; conv_igemm.hip's epilogue)
        const bool sched = p.sch.coef_table != nullptr;
        float c0 = 1.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f;
        const float* nz = nullptr;
        if (sched) {
            const int step = *p.sch.step_ptr;
            const float* c = p.sch.coef_table + 5 * step;
            c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3]; c4 = c[4];
            if (p.sch.noise && c4 != 0.f) nz = p.sch.noise + (size_t)step * p.sch.noise_step_stride;
        }
        const int ow = w0 + (tid >> 3), oh = h0 + (tid & 7);
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            if (ch < p.N) {
                const size_t i = (((size_t)b * p.N + ch) * p.Wout + ow) * p.Hout + oh;
                p.y_nchw[i] = e[ch];
                if (sched) {
                    const float x = p.sch.x[i];
                    float prev = sched_prev(p.sch.mode, c0, c1, c2, c3, x, e[ch]);
                    if (nz) prev += c4 * nz[i];
                    p.sch.x_prev[i] = prev;
                    if (p.sch.pack) p.sch.pack[(((size_t)b * p.Wout + ow) * p.Hout + oh) * p.sch.pack_ld + ch] = f32_to_bf16(prev);
                }
            }
        }
    }
}

size_t conv_o4_lds_bytes() { return 4 * kO4Wave + 2 * 128 * sizeof(float) + 2 * 128 * sizeof(double) + 4 * 128 * sizeof(float4); }

bool conv_o4_supported(const ConvParams& p) {
    if (p.C0 != 128 || p.C1 != 0 || p.R0 != 0 || p.R1 != 0 || p.N < 1 || p.N > 4 || p.up != 1 || p.stride != 1 || p.pad_lo != 1) return false;
    if (!p.y_nchw || p.temb || p.ksplit > 1 || p.TW != 16 || p.TH != 8 || p.y_stats) return false;
    if (p.st0 && (p.gn_groups <= 0 || 128 % p.gn_groups != 0)) return false;
    if (p.Win != p.Wout || p.Hin != p.Hout || p.Wout % 16 != 0 || p.Hout % 8 != 0 || (p.tiles_h & (p.tiles_h - 1)) != 0) return false;
    return p.B <= 65535 && p.tiles_img <= 65535;
}

int launch_conv_o4(const ConvParams& p, hipStream_t stream) {
    RLDM_REQUIRE(conv_o4_supported(p), "conv_o4: unsupported shape");
    const size_t lds = conv_o4_lds_bytes();
    auto kern = conv_o4_kernel;
    static DynLdsLimit lds_limit;
    RLDM_HIP_CHECK(lds_limit.ensure(reinterpret_cast<const void*>(kern), lds));
    hipLaunchKernelGGL(kern, dim3(p.tiles_img, p.B), dim3(256), lds, stream, p);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv_ds2_kernel: 3x3 / stride 2 / pad 1 over 256 channels onto a FEW pixels (the UNet's 64x4 -> 32x2 Downsample2D, ldm/utils.py:76-116: a
// convolution of the raw input, no norm).  1 024 output pixels x 256 channels with K = 2 304: on the generic kernel 64-128 workgroups walked K
// alone (21 us in the step graph).  conv_o4's wave grid: a workgroup is a 32-pixel output row x 32 output channels and its 8 waves are K-GROUPS --
// wave kg owns input channels [32 kg, 32 kg + 32): it copies its channels of the 65 x 3 input footprint into its own LDS region, keeps its 18
// weight fragments in registers, runs 18 MFMAs; the eight partial tiles meet in LDS.  (conv_small.hip cannot take a stride: it holds ALL input
// channels of its tile in LDS, and a stride-2 footprint of 64 pixels x 256 channels is 166 KB.)
namespace {
constexpr int kDsRS = 32 * 2 + 16;                      // a wave's halo row: 32 channels + a slot
constexpr int kDsColb = 272;                            // ... column pitch: 3 rows + padding; lanes read columns 2 l apart: 136 dwords = 8 (mod 64)
constexpr int kDsWave = 65 * kDsColb;                   // ... region (17 680 bytes)
}  // namespace

__global__ void __launch_bounds__(512, 1) conv_ds2_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int kg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = lane >> 5, l31 = lane & 31;
    int nt, mt, b;
    {
        const int gx = p.ntile_n, gy = p.tiles_img;
        const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const int rid = xcd_remap(lin, gx * gy * p.B);
        const int q = rid / gx;
        nt = rid - q * gx;
        b = q / gy;
        mt = q - b * gy;
    }
    const int tw = mt / p.Hout, oh = mt - tw * p.Hout;                  // the tile: output pixels (32 tw .. + 32, oh)
    const int w0 = tw * 32;
    unsigned char* const sW = smem + kg * kDsWave;

    // ---- the wave's channels of the footprint: 65 columns x 3 rows x 4 pieces = 780 pieces, 13 per lane (zero rows above / below the image) ----
    constexpr int NP = 13;
    uint4 hv[NP];
    const int c8 = lane & 3;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int slot = (lane >> 2) + i * 16;                          // footprint pixel: column slot / 3, row slot % 3
        const int vwl = (slot * 21846) >> 16, vhl = slot - vwl * 3;     // slot / 3 for slot < 32768
        const int vh = 2 * oh - 1 + vhl;
        int vw = 2 * w0 - 1 + vwl;
        vw = vw < 0 ? vw + p.Win : (vw >= p.Win ? vw - p.Win : vw);
        const bool ok = slot < 195 && vh >= 0 && vh < p.Hin;
        hv[i] = make_uint4(0u, 0u, 0u, 0u);
        if (ok) hv[i] = *reinterpret_cast<const uint4*>(p.x0 + ((size_t)(b * p.Win + vw) * p.Hin + vh) * 256 + kg * 32 + c8 * 8);
    }
    // (stream-packed with 2 k-groups: [32-channel tile][k-group][4 chunks x 9 taps x 2 k-steps] -- input channels 32 kg .. are chunk kg / 2, k-group kg % 2)
    const unsigned char* const wsrc = reinterpret_cast<const unsigned char*>(p.wpk) +
                                      ((size_t)(nt * 2 + (kg & 1)) * 72 + (kg >> 1) * 18) * 1024 + lane * 16;
    bf16x8 wr[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) wr[i] = *reinterpret_cast<const bf16x8*>(wsrc + i * 1024);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int slot = (lane >> 2) + i * 16;
        const int vwl = (slot * 21846) >> 16, vhl = slot - vwl * 3;
        if (slot < 195) *reinterpret_cast<uint4*>(sW + vwl * kDsColb + vhl * kDsRS + c8 * 16) = hv[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // (the wave's own stores: visible to the wave's own reads)
    // ---- 9 taps x 2 k-steps: output pixel l reads footprint column 2 l + tap column ----------------------------------------------------
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const unsigned char* const xp = sW + (2 * l31) * kDsColb + kh * 16;
#pragma unroll
    for (int step = 0; step < 18; ++step) {
        const int tap = step >> 1, ks = step & 1;
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xp + (tap / 3) * kDsColb + (tap % 3) * kDsRS + ks * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[step], xf, acc, 0, 0, 0);
    }
    // ---- the k-groups' partial tiles [pixel][32 channels] fp32 into the head of each wave's own region, then 256 threads sum them --------
    __syncthreads();                                                    // (every wave has consumed its footprint)
    constexpr int FRS = 32 * 4 + 16;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4)
        *reinterpret_cast<float4*>(sW + l31 * FRS + (8 * r4 + 4 * kh) * 4) = make_float4(acc[r4 * 4 + 0], acc[r4 * 4 + 1], acc[r4 * 4 + 2], acc[r4 * 4 + 3]);
    __syncthreads();
    unsigned char* const sT = smem + 8 * kDsWave;                       // the rounded tile [32 pixels][32 channels] bf16, row 80 bytes
    if (tid < 256) {
        const int px = tid >> 3, c4 = tid & 7;
        float4 sum = *reinterpret_cast<const float4*>(p.bias + nt * 32 + c4 * 4);
#pragma unroll
        for (int g = 0; g < 8; ++g) {                                   // fixed order: deterministic
            const float4 v = *reinterpret_cast<const float4*>(smem + g * kDsWave + px * FRS + c4 * 16);
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        uint2 o;
        o.x = pack_bf16x2(sum.x, sum.y);
        o.y = pack_bf16x2(sum.z, sum.w);
        *reinterpret_cast<uint2*>(p.y + (((size_t)b * p.Wout + w0 + px) * p.Hout + oh) * p.y_ld + nt * 32 + c4 * 4) = o;
        *reinterpret_cast<uint2*>(sT + px * 80 + c4 * 8) = o;
    }
    if (p.y_stats) {
        __syncthreads();
        if (tid < 64) {                                                 // (channel pair tid % 16, 8 pixels each; folded across the 4 groups by shuffles)
            const int cp = tid & 15, pg = tid >> 4;
            float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t w2 = *reinterpret_cast<const uint32_t*>(sT + (pg * 8 + j) * 80 + cp * 4);
                const float a0 = bf16lo(w2), a1 = bf16hi(w2);
                s0 += a0; s1 += a1;
                q0 += a0 * a0; q1 += a1 * a1;
            }
            s0 += __shfl_xor(s0, 16); s1 += __shfl_xor(s1, 16); q0 += __shfl_xor(q0, 16); q1 += __shfl_xor(q1, 16);
            s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32); q0 += __shfl_xor(q0, 32); q1 += __shfl_xor(q1, 32);
            if (tid < 16) {
                float2* dst = p.y_stats + ((size_t)b * p.tiles_img + mt) * p.N + nt * 32 + cp * 2;
                dst[0] = make_float2(s0, q0);
                dst[1] = make_float2(s1, q1);
            }
        }
    }
}

size_t conv_ds2_lds_bytes() { return 8 * kDsWave + 32 * 80; }

bool conv_ds2_supported(const ConvParams& p) {
    if (p.C0 != 256 || p.C1 != 0 || p.R0 != 0 || p.R1 != 0 || p.N % 32 != 0 || p.up != 1 || p.stride != 2 || p.pad_lo != 1) return false;
    if (p.st0 || p.temb || p.y_nchw || p.ksplit > 1) return false;
    if (p.Win != 2 * p.Wout || p.Hin != 2 * p.Hout || p.Wout % 32 != 0) return false;
    return p.B <= 65535 && p.tiles_img <= 65535 && p.tiles_img == (p.Wout / 32) * p.Hout;
}

int launch_conv_ds2(const ConvParams& p, hipStream_t stream) {
    RLDM_REQUIRE(conv_ds2_supported(p), "conv_ds2: unsupported shape");
    const size_t lds = conv_ds2_lds_bytes();
    auto kern = conv_ds2_kernel;
    static DynLdsLimit lds_limit;
    RLDM_HIP_CHECK(lds_limit.ensure(reinterpret_cast<const void*>(kern), lds));
    hipLaunchKernelGGL(kern, dim3(p.N / 32, p.tiles_img, p.B), dim3(512), lds, stream, p);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

static int regw_teams() { static const int t = getenv("RLDM_RW_TEAMS") ? atoi(getenv("RLDM_RW_TEAMS")) : 1; return t == 2 ? 2 : 1; }
size_t conv_regw_lds_bytes() { return regw_teams() * kRwTeamBytes + (3 * 64 + 8 * 2 * 32) * sizeof(float) + 2 * (regw_teams() == 2 ? 12 : 8) * 1024; }

// workgroups per image for a launch of one per CU (a power of two that divides the tiles of an image), 0: the shape is not served
int conv_regw_partials(const ConvParams& p) { return conv_regw_wg_per_image(p) / regw_teams(); }
int conv_regw_wg_per_image(const ConvParams& p) {
    if (p.Wout % kRwTW != 0 || p.Hout % kRwTH != 0 || p.B <= 0) return 0;
    const int tiles = (p.Wout / kRwTW) * (p.Hout / kRwTH);
    if (tiles & (tiles - 1)) return 0;
    const int cap = (p.exp & 0xffff) > 0 ? (p.exp & 0xffff) : 512;     // (p.exp: the tests' small grids -- rldm_debug_set_flags2(1 << 25))
    int wpi = 1;
    while (wpi * 2 <= tiles && (long long)wpi * 2 * p.B <= cap) wpi *= 2;     // (team runs per image)
    return wpi;
}

bool conv_regw_supported(const ConvParams& p) {
    if (p.C0 != kRwCK || p.C1 != 0 || p.up != 1 || p.stride != 1 || p.pad_lo != 1) return false;
    if (p.y_nchw ? (p.N < 1 || p.N > 4 || p.R0 != 0 || p.y_stats != nullptr) : p.N != 64) return false;
    if (p.Win != p.Wout || p.Hin != p.Hout || p.TW != kRwTW || p.TH != kRwTH) return false;
    if (!((p.R0 == 0 && p.R1 == 0) || (p.R0 == 64 && p.R1 == 0))) return false;
    if (p.st0 && (p.gn_groups <= 0 || kRwCK % p.gn_groups != 0)) return false;
    if (p.temb || p.ksplit > 1 || p.sch.coef_table) return false;
    if ((long long)p.B * p.Win * p.Hin * kRwCK * 2 >= (1ll << 32)) return false;      // (32-bit byte offsets)
    if ((p.tiles_h & (p.tiles_h - 1)) != 0) return false;
    const int wpi = conv_regw_wg_per_image(p);
    if (!wpi || wpi < regw_teams()) return false;
    const int tiles = (p.Wout / kRwTW) * (p.Hout / kRwTH);
    return tiles / wpi >= 4;                    // (runs of at least four tiles per team; shorter: the per-tile kernel's grid is as good)
}

// one launcher -- and one DynLdsLimit -- per kernel instance: the four instances share the pointer type void(*)(ConvParams, int, int),
// so a limit kept inside a generic lambda over the pointer would be ONE limit for all of them, and the instance launched second
// would never get its hipFuncSetAttribute(MaxDynamicSharedMemorySize)
template <int TEAMS, int WN>
static int launch_regw_inst(const ConvParams& p, int wpi, int tiles, size_t lds, hipStream_t stream) {
    auto kern = conv_regw_kernel<TEAMS, WN>;
    static DynLdsLimit lds_limit;
    RLDM_HIP_CHECK(lds_limit.ensure(reinterpret_cast<const void*>(kern), lds));
    hipLaunchKernelGGL(kern, dim3(wpi / TEAMS * p.B), dim3(TEAMS * kRwTeam), lds, stream, p, wpi / TEAMS, tiles / wpi);
    return 0;
}

int launch_conv_regw(const ConvParams& p, hipStream_t stream) {
    RLDM_REQUIRE(conv_regw_supported(p), "conv_regw: unsupported shape");
    RLDM_REQUIRE(p.colb == kRwColb, "conv_regw: halo column pitch");
    const int wpi = conv_regw_wg_per_image(p);
    const int tiles = (p.Wout / kRwTW) * (p.Hout / kRwTH);
    const size_t lds = conv_regw_lds_bytes();
    // (wpi = team runs per image: the statistics partials are one per WORKGROUP)
    int rc;
    if (regw_teams() == 2) rc = p.y_nchw ? launch_regw_inst<2, 1>(p, wpi, tiles, lds, stream) : launch_regw_inst<2, 2>(p, wpi, tiles, lds, stream);
    else rc = p.y_nchw ? launch_regw_inst<1, 1>(p, wpi, tiles, lds, stream) : launch_regw_inst<1, 2>(p, wpi, tiles, lds, stream);
    if (rc) return rc;
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace rldm
