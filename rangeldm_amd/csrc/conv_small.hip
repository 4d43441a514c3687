// Activation-stationary circular 3x3 convolution for the low-resolution UNet levels (gfx950, bf16 MFMA, fp32 accumulate).
//
// Same arithmetic as conv_igemm.hip -- conv 3x3 (wrap W / zero H) over an already normalised + activated input, bias + time
// embedding, shortcut-conv / residual as extra K, per-channel statistics of the output for the next GroupNorm
// (ldm/utils.py:40-58, vae/sgm/modules/diffusionmodules/model.py:342-362) -- but organised for the levels where an image is only
// 64..256 pixels (64x4, 32x2 latents).  There a wave issues ~1 instruction per 5 cycles whatever its kind, and the generic
// kernel's per-chunk work (barrier, weight DMA, halo pipeline: ~300 instructions per 9 MFMAs of a wave) is the whole cost;
// its GroupNorm prologue is also redone by every channel tile (4-8x).  So here:
//   * GroupNorm + SiLU of a single-tensor input are applied while the tile is on its way into LDS (the statistics fold of
//     the image overlaps the first loads; ~50 elements per thread, against a 7 us launch of its own); a concatenated
//     input (up-block conv1) is normalised once by a separate launch (norm.hip: gn_apply_kernel) into one bf16 tensor;
//   * the whole input tile of the workgroup (all C_in channels of its 64 pixels + halo, 60-150 KB) is copied to LDS ONCE.
//     A wave owns halo columns; everything per-lane about a piece is loop-invariant, so a piece costs a load, an add and a
//     store;
//   * the weights are packed on the host in MFMA fragment order, one contiguous stream per (32-channel tile, k-group), and
//     go straight from L2 into registers: one global_load (scalar base + lane offset + immediate) per k-step, a ring of
//     G = 3..12 fragments in flight, exact s_waitcnt counts (the ring is issued in program order, pinned by sched_barrier);
//   * the pixel operand is one ds_read_b128 per 32 pixels at a per-tap base register + immediate;
//   * a k-step therefore costs 2 MFMAs + 3 memory instructions; no barrier, no LDS ring, no DMA issue in the K loop;
//   * 8 waves = NWN 32-channel tiles x KG k-groups.  Within every tap, k-group kg takes the 16-channel groups kg, kg + KG,
//     ...; the k-groups' fp32 partial tiles meet in LDS, where all 512 threads sum them, round, store 16 bytes each and
//     keep (sum, sumsq) of the rounded values for the next GroupNorm (conv_igemm.hip's contract).
#include "conv_small_body.h"

namespace rldm {

template <int NWN, int CPT, int TAPS, int MI, bool H16 = false>
__global__ void __launch_bounds__(512, 1) conv_small_kernel(const ConvParams p) {
    bf16x8 wpf[kTrunkPrefetch];                             // (persistent trunk only: unused here)
    const TrunkSeam none = {};
    conv_small_body<NWN, CPT, TAPS, MI, false, kTrunkPrefetch, H16>(p, blockIdx.x, blockIdx.y, blockIdx.z, wpf, none);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
int conv_small_col_bytes(int Cin, int TH, int taps) {
    const int rs = Cin / 8 + 1;                 // 16-byte slots per row (odd)
    int slots = (TH + (taps == 9 ? 2 : 0)) * rs;
    while (slots % 16 != TH % 16) ++slots;      // same conflict-freedom rule as conv_halo_col_bytes
    return slots * 16;
}

int conv_small_kgroups(int BN) { return BN <= 32 ? 8 : 8 / (BN / 32); }     // (16-channel tiles: 8 k-groups of 32-channel steps)

size_t conv_small_lds_bytes(const ConvParams& p, int taps, int BN) {
    const int BM = 64, KG = conv_small_kgroups(BN);             // (epilogue: 64-pixel half-tiles)
    const int Cin = p.C0 + p.C1, R = p.R0 + p.R1;
    const size_t a = (size_t)(p.TW + (taps == 9 ? 2 : 0)) * p.colb;
    const size_t r = (size_t)p.TW * p.TH * (R * 2 + 16);
    const size_t main_bytes = a + r + BN * 4 + 3072;            // + read-ahead slack past the residual image
    const size_t gn_bytes = a + 64 + (size_t)2 * Cin * 8 + (size_t)2 * Cin * 4;    // GroupNorm scratch behind the image
    const size_t epi = (size_t)KG * BM * (BN * 4 + 16) + (size_t)BM * (BN * 2 + 16) + (size_t)(8 * 2 + 2 + 6) * BN * 4;
    return std::max(std::max(main_bytes, gn_bytes), epi);
}

// k-steps per tap of one k-group, or 0 if the kernel is not instantiated for this (taps, BN, Cin)
static int small_cpt(int Cin, int taps, int BN) {
    if (BN == 16) return (Cin == 256 || (Cin == 512 && taps == 9)) ? Cin / 256 : 0;      // (k-steps of 32 channels x 8 k-groups)
    const int KG = conv_small_kgroups(BN);
    if (Cin % (16 * KG) != 0 || Cin > 512) return 0;
    const int cpt = Cin / (16 * KG);
    if (taps == 9) return (BN <= 64 && (cpt == 1 || cpt == 2 || cpt == 3 || cpt == 4 || cpt == 6 || cpt == 8)) ? cpt : 0;
    return (cpt == 2 || cpt == 4 || cpt == 8) ? cpt : 0;
}

bool conv_small_supported(const ConvParams& p, int taps, int BN) {
    const int Cin = p.C0 + p.C1, R = p.R0 + p.R1;
    if (BN != 16 && BN != 32 && BN != 64 && BN != 128) return false;
    // (round 5) 16-channel tiles: image-owning 64-pixel tiles only, pre-activated or plain input (no statistics fold in the staging)
    if (BN == 16 && (p.TW * p.TH != 64 || p.tiles_img != 1 || p.up != 1 || p.st0 != nullptr || p.C1 != 0)) return false;
    if ((taps != 9 && taps != 1) || p.stride != 1 || (p.up != 1 && !(p.up == 2 && taps == 9 && p.R0 + p.R1 == 0 && !p.res)) ||
        p.pad_lo != (taps == 9 ? 1 : 0) || p.y_nchw || p.ksplit > 1)
        return false;
    // one input tensor, or (round 4) a concatenated 3x3 input whose GroupNorm is folded into the staging
    if (p.C1 != 0 && !(taps == 9 && p.st0 != nullptr && p.C0 % 8 == 0 && p.C1 % 8 == 0)) return false;
    if (p.st0 != nullptr && (p.gn_groups > 64 || Cin % p.gn_groups != 0)) return false;   // GroupNorm (+ SiLU) folded into the staging
    const int KG = conv_small_kgroups(BN), cpt = small_cpt(Cin, taps, BN);
    const int ksc = BN == 16 ? 32 : 16;
    if (cpt == 0 || p.N % BN != 0 || R % (ksc * KG) != 0 || p.R0 % 8 != 0 || R > 512) return false;
    const int G = BN == 16 ? (taps == 1 ? 1 : 9) * cpt : (taps == 1 ? 1 : (cpt <= 4 ? 3 : 1)) * cpt;
    if (R / (ksc * KG) > std::min(G, 8)) return false;
    const int BMpx = p.TW * p.TH;
    if ((BMpx != 32 && BMpx != 64 && BMpx != 128) || (p.TH < 2 && BMpx != 32) || p.TW + 2 > 40 || p.Win * p.up < 2) return false;
    // 32-pixel tiles (32x1 images): the 256- / 512-channel 3x3 convs and the 256-channel pointwise conv, 32-channel tiles
    if (BMpx == 32 && (BN != 32 || p.up != 1 || !((taps == 9 && (cpt == 2 || cpt == 4)) || (taps == 1 && cpt == 2)))) return false;
    if (BMpx == 128 && (taps != 9 || BN != 64 || p.TW + 2 > 24 || (cpt != 2 && cpt != 4 && cpt != 6))) return false;
    if ((p.tiles_h & (p.tiles_h - 1)) != 0 || p.B > 65535 || p.tiles_img > 65535) return false;
    // normalised copies for the consumers: the tile owns the image, statistics on, whole groups inside a 32-channel tile
    if (p.nviews < 0 || p.nviews > 3 || (p.nviews > 0 && (p.tiles_img != 1 || BMpx > 64))) return false;
    for (int v = 0; v < p.nviews; ++v)
        if (p.nv[v].cpg_shift < 0 || p.nv[v].cpg_shift > 5) return false;
    return conv_small_lds_bytes(p, taps, BN) <= 160 * 1024;
}

template <int NWN, int CPT, int TAPS, int MI, bool H16 = false>
static int launch_small_inst(const ConvParams& p, size_t lds, hipStream_t stream) {
    auto kern = conv_small_kernel<NWN, CPT, TAPS, MI, H16>;
    static DynLdsLimit lds_limit;                // per device, thread safe
    RLDM_HIP_CHECK(lds_limit.ensure(reinterpret_cast<const void*>(kern), lds));
    hipLaunchKernelGGL(kern, dim3(p.ntile_n, p.tiles_img, p.B), dim3(512), lds, stream, p);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_conv_small(const ConvParams& p, int taps, int BN, hipStream_t stream) {
    RLDM_REQUIRE(conv_small_supported(p, taps, BN), "conv_small: unsupported shape");
    const size_t lds = conv_small_lds_bytes(p, taps, BN);
    const int cpt = small_cpt(p.C0 + p.C1, taps, BN);
    const int mi = p.TW * p.TH / 32;
    if (BN == 16) {                             // (round 5) 16-channel image-owning tiles
        if (cpt == 1 && taps == 9) return launch_small_inst<1, 1, 9, 2, true>(p, lds, stream);
        if (cpt == 2 && taps == 9) return launch_small_inst<1, 2, 9, 2, true>(p, lds, stream);
        if (cpt == 1 && taps == 1) return launch_small_inst<1, 1, 1, 2, true>(p, lds, stream);
        RLDM_REQUIRE(false, "conv_small: no 16-channel instance");
    }
#define RLDM_SMALL4(NWN_, CPT_, TAPS_, MI_) \
    if (BN == 32 * NWN_ && cpt == CPT_ && taps == TAPS_ && mi == MI_) return launch_small_inst<NWN_, CPT_, TAPS_, MI_>(p, lds, stream);
#define RLDM_SMALL(NWN_, CPT_, TAPS_) RLDM_SMALL4(NWN_, CPT_, TAPS_, 2)
    RLDM_SMALL(1, 1, 9) RLDM_SMALL(1, 2, 9) RLDM_SMALL(1, 3, 9) RLDM_SMALL(1, 4, 9)
    RLDM_SMALL(2, 1, 9) RLDM_SMALL(2, 2, 9) RLDM_SMALL(2, 3, 9) RLDM_SMALL(2, 4, 9) RLDM_SMALL(2, 6, 9) RLDM_SMALL(2, 8, 9)
    RLDM_SMALL(1, 2, 1) RLDM_SMALL(1, 4, 1)
    RLDM_SMALL(2, 2, 1) RLDM_SMALL(2, 4, 1) RLDM_SMALL(2, 8, 1)
    RLDM_SMALL(4, 4, 1) RLDM_SMALL(4, 8, 1)
    RLDM_SMALL4(2, 2, 9, 4) RLDM_SMALL4(2, 4, 9, 4) RLDM_SMALL4(2, 6, 9, 4)
    RLDM_SMALL4(1, 2, 9, 1) RLDM_SMALL4(1, 4, 9, 1) RLDM_SMALL4(1, 2, 1, 1)
#undef RLDM_SMALL
#undef RLDM_SMALL4
    RLDM_REQUIRE(false, "conv_small: no instance");
    return 1;
}

}  // namespace rldm
