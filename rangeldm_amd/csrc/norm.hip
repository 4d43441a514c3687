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

// GroupNorm (+SiLU) of cat[x0, x1] from the producers' per-channel partials -> one activated bf16 tensor.  A block
// re-derives the per-channel affine of its image (a few KB of partials) and then streams `ppb` pixels: 16 bytes per
// lane in, 16 bytes out.  The finalize arithmetic is the conv prologue's (conv_igemm.hip), so both routes agree bit for bit.
__global__ void __launch_bounds__(256) gn_apply_kernel(const GnApplyParams p, int ppb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    const int tid = threadIdx.x;
    const int Cin = p.C0 + p.C1;
    const int b = blockIdx.y;
    double* sD = reinterpret_cast<double*>(gsm);                 // [2*Cin + 2*groups]
    float* sGa = reinterpret_cast<float*>(sD + 2 * Cin + 2 * p.groups);
    float* sGs = sGa + Cin;
    const int cpg = Cin / p.groups;
    for (int t = tid; t < Cin; t += 256) {
        const bool first = t < p.C0;
        const int c = first ? t : t - p.C0;
        const int C = first ? p.C0 : p.C1;
        const int P = first ? p.P0 : p.P1;
        const float2* src = (first ? p.st0 : p.st1) + (size_t)b * P * C + c;
        double S = 0.0, SS = 0.0;
        for (int q = 0; q < P; ++q) {
            const float2 v = src[(size_t)q * C];
            S += (double)v.x;
            SS += (double)v.y;
        }
        sD[t] = S;
        sD[Cin + t] = SS;
    }
    __syncthreads();
    if (tid < p.groups) {
        double S = 0.0, SS = 0.0;
        for (int i = 0; i < cpg; ++i) {
            S += sD[tid * cpg + i];
            SS += sD[Cin + tid * cpg + i];
        }
        const double inv_n = 1.0 / ((double)p.npix * (double)cpg);
        const double mean = S * inv_n;
        double var = SS * inv_n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        sD[2 * Cin + tid] = mean;
        sD[2 * Cin + p.groups + tid] = (double)__builtin_amdgcn_rsqf((float)var + p.eps);
    }
    __syncthreads();
    for (int t = tid; t < Cin; t += 256) {
        const int g = t / cpg;
        const float ga = p.gamma[t] * (float)sD[2 * Cin + p.groups + g];
        sGa[t] = ga;
        sGs[t] = p.beta[t] - (float)sD[2 * Cin + g] * ga;
    }
    __syncthreads();
    const int C8 = Cin >> 3;
    const int px0 = blockIdx.x * ppb;
    const int total = min(ppb, p.npix - px0) * C8;
    for (int q = tid; q < total; q += 256) {
        const int px = px0 + q / C8, c = (q % C8) * 8;
        const bool first = c < p.C0;
        const bf16_t* src = first ? p.x0 + ((size_t)b * p.npix + px) * p.C0 + c
                                  : p.x1 + ((size_t)b * p.npix + px) * p.C1 + (c - p.C0);
        const uint4 v = *reinterpret_cast<const uint4*>(src);
        const float4 a0 = *reinterpret_cast<const float4*>(sGa + c), a1 = *reinterpret_cast<const float4*>(sGa + c + 4);
        const float4 s0 = *reinterpret_cast<const float4*>(sGs + c), s1 = *reinterpret_cast<const float4*>(sGs + c + 4);
        float f0 = bf16lo(v.x) * a0.x + s0.x, f1 = bf16hi(v.x) * a0.y + s0.y;
        float f2 = bf16lo(v.y) * a0.z + s0.z, f3 = bf16hi(v.y) * a0.w + s0.w;
        float f4 = bf16lo(v.z) * a1.x + s1.x, f5 = bf16hi(v.z) * a1.y + s1.y;
        float f6 = bf16lo(v.w) * a1.z + s1.z, f7 = bf16hi(v.w) * a1.w + s1.w;
        if (p.silu) {
            f0 = silu_f(f0); f1 = silu_f(f1); f2 = silu_f(f2); f3 = silu_f(f3);
            f4 = silu_f(f4); f5 = silu_f(f5); f6 = silu_f(f6); f7 = silu_f(f7);
        }
        uint4 o;
        o.x = pack_bf16x2(f0, f1); o.y = pack_bf16x2(f2, f3);
        o.z = pack_bf16x2(f4, f5); o.w = pack_bf16x2(f6, f7);
        *reinterpret_cast<uint4*>(p.y + ((size_t)b * p.npix + px) * Cin + c) = o;
    }
}

int launch_gn_apply(const GnApplyParams& p, hipStream_t stream) {
    const int Cin = p.C0 + p.C1;
    RLDM_REQUIRE(Cin % 8 == 0 && p.C0 % 8 == 0 && Cin % p.groups == 0 && p.groups <= 256, "gn_apply: unsupported channels");
    int ppb = std::max(1, 512 / (Cin / 8));                     // ~2 pieces per thread
    while (ppb > 1 && (long long)p.B * ((p.npix + ppb - 1) / ppb) < 512) ppb >>= 1;
    const size_t lds = ((size_t)2 * Cin + 2 * p.groups) * 8 + (size_t)Cin * 8;
    hipLaunchKernelGGL(gn_apply_kernel, dim3((p.npix + ppb - 1) / ppb, p.B), dim3(256), lds, stream, p, ppb);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}


}  // namespace rldm
