// Weight-streaming circular 3x3 convolution for the full-resolution levels (UNet 256x16, VAE decoder) on gfx950:
// bf16 MFMA 32x32x16, fp32 accumulate, 256 pixels x 128 channels per workgroup.
//
// Same fused unit as conv_igemm.hip -- GroupNorm + SiLU of cat[x0, x1] on the way into LDS, conv 3x3 (wrap W / zero H,
// optional nearest-x2 folded into the indexing), bias + time embedding, shortcut-conv / residual as extra K over the raw
// block input, per-channel statistics of the output for the next GroupNorm (ldm/utils.py:40-58,107-116;
// vae/sgm/modules/diffusionmodules/model.py:93-125,342-362) -- with a different division of labour between LDS and L2:
//   * only the ACTIVATIONS go through LDS: per 64-channel chunk the halo of the 32x8 pixel tile (34x10 positions) is
//     double-buffered, chunk c+1 is fetched and normalised while chunk c computes, and the workgroup meets at ONE barrier
//     per chunk (conv_igemm.hip: one per tap, because its weight ring lives in LDS too);
//   * the WEIGHTS are packed on the host in MFMA A-fragment order, one contiguous stream of 1 KiB k-steps per 32-channel
//     tile, and every wave loads its own fragments straight from L2 into registers: a ring of 12 fragments (one row of
//     taps) in flight, refilled in program order so the compiler's s_waitcnt counts are exact; the stream never stops at
//     a barrier;
//   * 8 waves = 2 pixel halves x 4 channel tiles; a wave owns 128 pixels x 32 channels (4 MFMAs per fragment), so each
//     fragment is requested by two waves (the second hit is an L1 hit) and the LDS feeds 4 pixel fragments per k-step;
//   * waves 0-3 normalise their share of the next chunk after the first row of taps, waves 4-7 after the second, so the
//     VALU work (GroupNorm affine + SiLU) of one half runs under the other half's MFMAs: the two waves of a SIMD are never
//     both in it.
// Second instance for the 128x8 level (too few 256-pixel tiles to fill the chip): 128 pixels x 64 channels per workgroup,
// 8 waves = 2 channel tiles x 4 k-groups (k-group kg owns the kg-th 16-channel group of every tap of a chunk: one k-step
// per tap, ring = the 9 steps of a chunk); the k-groups' fp32 partial tiles meet in LDS for the epilogue (conv_small.hip's).
#include "conv_stream_body.h"
#include "conv_stream_spec_body.h"

namespace rldm {

// WM pixel parts (128 pixels each) x WN 32-channel tiles x KG k-groups = NW waves; the 4-wave instances are built for two workgroups
// per CU (__launch_bounds__' second argument is waves per SIMD: 2 x 256 threads = 2)
template <int WM, int WN, int NW = 8, int MI = 4, int S = 1, bool SUB = false, bool T4 = false, bool FH = false>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) conv_stream_kernel(const ConvParams p) {
    // Workgroups are dispatched x-fastest and land on XCD (linear id % 8).  Re-number them so that every XCD owns a contiguous
    // run of (image, pixel tile, channel tile) ids: the tiles of an image then share ONE L2, and the halo rows two neighbouring
    // tiles both read (34 x 10 positions for 32 x 8 pixels: 1.33x the tile) are fetched from HBM / Infinity Cache once.
    int nt, mt, b;
    {
        const int gx = p.ntile_n, gy = p.tiles_img;    // (== gridDim.x / .y: from the arguments, not a dependent read of the dispatch packet)
        const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const int rid = (p.dbg & (1 << 21)) ? lin : xcd_remap(lin, gx * gy * p.B);
        const int q = rid / gx;
        nt = rid - q * gx;
        b = q / gy;
        mt = q - b * gy;
    }
    const TrunkSeam none = {};
    conv_stream_body<WM, WN, false, NW, MI, S, SUB, T4, FH>(p, nt, mt, b, none);
}

// the 256 x 128 tile with specialised waves (conv_stream_spec_body.h): 4 matrix waves + 4 staging waves
__global__ void __launch_bounds__(512, 1) conv_stream_spec_kernel(const ConvParams p) {
    int nt, mt, b;
    {
        const int gx = p.ntile_n, gy = p.tiles_img;
        const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const int rid = (p.dbg & (1 << 21)) ? lin : xcd_remap(lin, gx * gy * p.B);
        const int q = rid / gx;
        nt = rid - q * gx;
        b = q / gy;
        mt = q - b * gy;
    }
    const TrunkSeam none = {};
    conv_stream_spec_body<false>(p, nt, mt, b, none);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
// the instance is chosen by the pixel tile: 32 x 8 -> 256 px x 128 ch (one k-group), 16 x 8 -> 128 px x 64 ch (4 k-groups)
// the instance is chosen by the pixel tile: 256 pixels (32 x 8) -> <2, 4> (128 channels, no k-groups); 128 pixels (16 x 8, or 32 x 4 for
// images of 4 beams) -> <1, 2> (64 channels x 4 k-groups)
// (round 3) <2, 2>: 256 pixels x 64 channels x 2 k-groups for layers of 64 (192, ...) output channels -- the VAE decoder's
// full-resolution level, which ran on the generic kernel at 278 us per conv
// (round 4) st_inst 1 / 2: 4-wave workgroups on 16 x 8 tiles -- 128 channels (one k-group) / 64 channels x 2 k-groups; two per CU
// (round 4) st_inst 6: st_inst 1's tile for nearest x2 + 3x3 convs in their sub-pixel form (conv_stream_body.h, SUB)
// (round 4) st_inst 4: 8 x 8 tiles, 64 pixels x 128 channels x 2 k-groups on 8 waves (two 32-pixel fragments per wave) for the 128x8 level
int conv_stream_bn(const ConvParams& p) {
    if (p.st_inst == 3 || p.st_inst == 4 || p.st_inst == 5) return 128;       // (3: the 256 x 128 tile with specialised waves; 5: 4 at stride 2)
    if (p.st_inst) return p.st_inst == 1 || p.st_inst == 6 || p.st_inst == 7 ? 128 : 64;
    return p.TW * p.TH == 256 && p.N % 128 == 0 ? 128 : 64;
}
int conv_stream_kgroups(const ConvParams& p) {
    if (p.st_inst == 3) return 1;
    if (p.st_inst == 4 || p.st_inst == 5) return 2;
    if (p.st_inst) return p.st_inst == 1 || p.st_inst == 6 || p.st_inst == 7 ? 1 : 2;
    return p.TW * p.TH == 256 ? (p.N % 128 == 0 ? 1 : 2) : 4;
}
int conv_stream_threads(const ConvParams& p) { return p.st_inst == 1 || p.st_inst == 2 || p.st_inst == 6 || p.st_inst == 7 ? 256 : 512; }

size_t conv_stream_lds_bytes(const ConvParams& p) {
    const int BN = conv_stream_bn(p), KG = conv_stream_kgroups(p), BM = p.TW * p.TH;
    const size_t a = (size_t)((p.TW - 1) * p.stride + 3) * p.colb;
    const size_t main_bytes = 2 * a + (size_t)(p.C0 + p.C1) * 8 + BN * 4;
    const size_t gscratch = p.st0 ? (size_t)2 * (p.C0 + p.C1) * 8 : 0;
    const size_t epi = KG == 1 ? (size_t)BM * (BN * 2 + 16) + (size_t)8 * 2 * BN * 4
                               : (size_t)KG * 64 * (BN * 4 + 16) + (size_t)64 * (BN * 2 + 16) + (size_t)8 * 2 * BN * 4;
    return std::max(std::max(main_bytes, gscratch), epi);
}

bool conv_stream_supported(const ConvParams& p, int taps) {
    const int Cin = p.C0 + p.C1, R = p.R0 + p.R1;
    if (taps != 9 || (p.stride != 1 && !(p.stride == 2 && p.st_inst == 5)) || p.pad_lo != 1 || (p.up != 1 && p.up != 2) || p.y_nchw || p.ksplit > 1) return false;
    if (p.st_inst == 5 && (p.stride != 2 || p.up != 1 || R != 0)) return false;
    if (Cin % 64 != 0 || (p.C1 != 0 && p.C0 % 64 != 0) || R % 64 != 0 || (p.R1 != 0 && p.R0 % 64 != 0)) return false;
    if (R != 0 && p.up != 1) return false;
    if (!((p.TW == 32 && p.TH == 8) || (p.TW == 16 && p.TH == 8) || (p.TW == 32 && p.TH == 4) || (p.st_inst == 7 && p.TW == 8 && p.TH == 16) || ((p.st_inst == 4 || p.st_inst == 5) && p.TW == 8 && p.TH == 8) || (p.st_inst == 5 && p.TW == 16 && p.TH == 4)) ||
        p.Win * p.up < 2) return false;
    if (p.N % conv_stream_bn(p) != 0 || Cin > 512) return false;
    if (p.st_inst == 3 && (p.TW != 32 || p.TH != 8 || p.N % 128 != 0)) return false;
    if (p.st_inst == 4 && (p.TW != 8 || p.TH != 8 || p.N % 128 != 0)) return false;
    // (5: 8 x 8 tiles, or 16 x 4 for outputs of 4 beams -- a 33 x 9 halo, the same five pieces per thread)
    if (p.st_inst == 5 && (!((p.TW == 8 && p.TH == 8) || (p.TW == 16 && p.TH == 4)) || p.N % 128 != 0)) return false;
    if ((p.st_inst == 1 || p.st_inst == 2) && (p.TW != 16 || p.TH != 8)) return false;
    if (p.st_inst == 6 && !((p.TW == 16 && p.TH == 8) || (p.TW == 32 && p.TH == 4))) return false;
    // (6, round 4: nearest x2 + 3x3 in its sub-pixel form -- tiles over the INPUT, four parities per tile, Wout = 2 Win)
    if (p.st_inst == 6 && (p.up != 1 || R != 0 || p.N % 128 != 0 || p.Wout != 2 * p.Win || p.Hout != 2 * p.Hin)) return false;
    // (7, round 4: st_inst 1 on 8 x 16 tiles as tall as the image -- the halo rows above / below are never staged)
    if (p.st_inst == 7 && (!((p.TW == 8 && p.TH == 16) || (p.TW == 16 && p.TH == 8)) || p.Hout != p.TH || p.up != 1 || p.N % 128 != 0)) return false;
    if (p.st_inst < 0 || p.st_inst > 7) return false;
    if (p.st0 && (p.gn_groups > 64 || Cin % p.gn_groups != 0)) return false;
    if ((p.tiles_h & (p.tiles_h - 1)) != 0 || p.B > 65535 || p.tiles_img > 65535) return false;
    return conv_stream_lds_bytes(p) <= (size_t)(p.st_inst == 1 || p.st_inst == 2 || p.st_inst == 6 || p.st_inst == 7 ? 80 : 160) * 1024;      // (4-wave instances: two workgroups share the CU's LDS)
}

template <int WM, int WN, int NW = 8, int MI = 4, int S = 1, bool SUB = false, bool T4 = false, bool FH = false>
static int launch_stream_inst(const ConvParams& p, size_t lds, hipStream_t stream) {
    auto kern = conv_stream_kernel<WM, WN, NW, MI, S, SUB, T4, FH>;
    static DynLdsLimit lds_limit;                // per device, thread safe
    RLDM_HIP_CHECK(lds_limit.ensure(reinterpret_cast<const void*>(kern), lds));
    hipLaunchKernelGGL(kern, dim3(p.N / (32 * WN) * (SUB ? 4 : 1), p.tiles_img, p.B), dim3(64 * NW), lds, stream, p);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_conv_stream(const ConvParams& p, hipStream_t stream) {
    RLDM_REQUIRE(conv_stream_supported(p, 9), "conv_stream: unsupported shape");
    const size_t lds = conv_stream_lds_bytes(p);
    if (p.st_inst == 3) {
        auto kern = conv_stream_spec_kernel;
        static DynLdsLimit lds_limit;
        RLDM_HIP_CHECK(lds_limit.ensure(reinterpret_cast<const void*>(kern), lds));
        hipLaunchKernelGGL(kern, dim3(p.N / 128, p.tiles_img, p.B), dim3(512), lds, stream, p);
        RLDM_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (p.st_inst == 4) return launch_stream_inst<1, 4, 8, 2>(p, lds, stream);
    if (p.st_inst == 5) return launch_stream_inst<1, 4, 8, 2, 2>(p, lds, stream);
    if (p.st_inst == 6) return p.TH == 4 ? launch_stream_inst<1, 4, 4, 4, 1, true, true>(p, lds, stream) : launch_stream_inst<1, 4, 4, 4, 1, true>(p, lds, stream);
    if (p.st_inst == 7) return launch_stream_inst<1, 4, 4, 4, 1, false, false, true>(p, lds, stream);
    if (p.st_inst == 1) return launch_stream_inst<1, 4, 4>(p, lds, stream);
    if (p.st_inst == 2) return launch_stream_inst<1, 2, 4>(p, lds, stream);
    if (p.TW * p.TH == 256) return p.N % 128 == 0 ? launch_stream_inst<2, 4>(p, lds, stream) : launch_stream_inst<2, 2>(p, lds, stream);
    return launch_stream_inst<1, 2>(p, lds, stream);
}

}  // namespace rldm
