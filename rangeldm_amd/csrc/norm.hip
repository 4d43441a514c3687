// Per-channel statistics for gfx950: deterministic partial (sum, sum of squares) per (image, pixel-chunk, channel) of a
// channels-last bf16 tensor, in the exact [B][P][C] float2 layout the conv epilogue emits (conv_igemm.hip), for
// tensors that did not come out of a conv launch (external inputs of the kernel-level entry points).  The consuming
// conv's prologue folds them into the GroupNorm affine.
// Reference op: torch.nn.GroupNorm(32, C, eps) in ResnetBlock2D / Attention [3P diffusers] and
// `Normalize` (vae/sgm/modules/diffusionmodules/model.py:59-62).
//
// HBM-bound streaming read: every lane loads 16 B (8 channels) of one pixel; a wave covers floor(64/(C/8)) pixels per
// pass, fully coalesced.  The reduction order is fixed (no atomics) so results are bit-reproducible run to run.
#include "kernels.h"

namespace rldm {

__global__ void __launch_bounds__(256) gn_stats_kernel(const GnStatsParams p) {
    __shared__ float sSum[256 * 8];
    __shared__ float sSq[256 * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = p.C;
    const int nch8 = C >> 3;                 // 16-byte pieces per pixel (<= 64)
    const int ppw = 64 / nch8;               // pixels per wave pass
    const int c8 = lane % nch8, sub = lane / nch8;
    const bool active = sub < ppw;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int per = (p.npix + p.P - 1) / p.P;
    const int px0 = chunk * per;
    const int px1 = min(p.npix, px0 + per);

    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    if (active) {
        const bf16_t* base = p.x + c8 * 8;
        for (int px = px0 + wave * ppw + sub; px < px1; px += 4 * ppw) {
            const uint4 v = *reinterpret_cast<const uint4*>(base + ((size_t)b * p.npix + px) * C);
            const float f[8] = {bf16lo(v.x), bf16hi(v.x), bf16lo(v.y), bf16hi(v.y),
                                bf16lo(v.z), bf16hi(v.z), bf16lo(v.w), bf16hi(v.w)};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s[e] += f[e];
                q[e] += f[e] * f[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sSum[tid * 8 + e] = s[e];
        sSq[tid * 8 + e] = q[e];
    }
    __syncthreads();
    // per-channel totals, fixed order over (wave, sub-pixel lane)
    for (int c = tid; c < C; c += 256) {
        const int cc8 = c >> 3, e = c & 7;
        float a = 0.f, d = 0.f;
        for (int w = 0; w < 4; ++w)
            for (int j = 0; j < ppw; ++j) {
                const int t = w * 64 + j * nch8 + cc8;
                a += sSum[t * 8 + e];
                d += sSq[t * 8 + e];
            }
        p.part[((size_t)b * p.P + chunk) * C + c] = make_float2(a, d);
    }
}

int launch_gn_stats(const GnStatsParams& p, hipStream_t stream) {
    RLDM_REQUIRE(p.C % 8 == 0 && p.C <= 512, "gn_stats: channels must be a multiple of 8, <= 512");
    hipLaunchKernelGGL(gn_stats_kernel, dim3(p.P, p.B), dim3(256), 0, stream, p);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace rldm
