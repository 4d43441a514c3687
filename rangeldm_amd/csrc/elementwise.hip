// Small HBM-bound kernels of the RangeLDM path (gfx950): boundary layout conversions, conv_in input packing
// (pos-encoding channel / per-step condition concat without a `torch.cat` copy), timestep embedding + all per-resnet
// projections in one launch, DDIM / DDPM scheduler steps, add_noise, DiagonalGaussian sampling.
#include "kernels.h"

namespace rldm {

// ---- conv_in input: [B][W][H][Cpad] bf16 <- x (cx ch, fp32 NCHW, * scale) | pos-encoding | cond (cc ch) | 0 --------
// replaces torch.cat([latents, pos_encoding], 1) / torch.cat([latents, image], 1): ldm/pipelines.py:238,358,498
__global__ void __launch_bounds__(256) pack_input_kernel(const PackInputParams p) {
    const long long npix = (long long)p.B * p.W * p.H;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const int hw = p.W * p.H;
    const int b = (int)(i / hw);
    const int r = (int)(i - (long long)b * hw);
    const int w = r / p.H;
    bf16_t* dst = p.out + i * p.Cpad;
    int c = 0;
    for (int k = 0; k < p.cx; ++k, ++c) dst[c] = f32_to_bf16(p.x[((size_t)b * p.cx + k) * hw + r] * p.scale);
    if (p.pos_encoding) dst[c++] = f32_to_bf16(w == 0 ? 1.0f : 0.0f);
    for (int k = 0; k < p.cc; ++k, ++c) dst[c] = f32_to_bf16(p.cond[((size_t)b * p.cc + k) * hw + r]);
    for (; c < p.Cpad; ++c) dst[c] = 0;
}

int launch_pack_input(const PackInputParams& p, hipStream_t stream) {
    RLDM_REQUIRE(p.cx + (p.pos_encoding ? 1 : 0) + p.cc <= p.Cpad, "pack_input: channels exceed padded width");
    const long long npix = (long long)p.B * p.W * p.H;
    hipLaunchKernelGGL(pack_input_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, stream, p);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* src, bf16_t* dst, int B, int C, int hw, int Cpad) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // over B*hw*Cpad, c fastest
    const long long total = (long long)B * hw * Cpad;
    if (i >= total) return;
    const int c = (int)(i % Cpad);
    const long long pix = i / Cpad;
    const int b = (int)(pix / hw), r = (int)(pix - (long long)b * hw);
    dst[i] = c < C ? f32_to_bf16(src[((size_t)b * C + c) * hw + r]) : (bf16_t)0;
}
int launch_nchw_f32_to_nhwc_bf16(const float* src, bf16_t* dst, int B, int C, int W, int H, int Cpad, hipStream_t s) {
    const long long total = (long long)B * W * H * Cpad;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, B, C,
                       W * H, Cpad);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const bf16_t* src, float* dst, int B, int C, int hw, int ld) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // over B*C*hw, r fastest
    const long long total = (long long)B * C * hw;
    if (i >= total) return;
    const int r = (int)(i % hw);
    const long long bc = i / hw;
    const int c = (int)(bc % C), b = (int)(bc / C);
    dst[i] = bf16_to_f32(src[((size_t)b * hw + r) * ld + c]);
}
int launch_nhwc_bf16_to_nchw_f32(const bf16_t* src, float* dst, int B, int C, int W, int H, int ld, hipStream_t s) {
    const long long total = (long long)B * C * W * H;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, B, C,
                       W * H, ld);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- time embedding + every resnet's time_emb_proj, one block per timestep row -------------------------------------
// diffusers Timesteps(flip_sin_to_cos=True, freq_shift=0) + TimestepEmbedding + ResnetBlock2D.time_emb_proj(SiLU(emb))
// [3P; SURVEY.md A.2]; analogue vae/sgm/modules/diffusionmodules/model.py:28-46,349.  fp32 throughout.
__global__ void __launch_bounds__(256) temb_kernel(const TembParams p) {
    extern __shared__ float sm[];
    float* e = sm;                  // dim0
    float* h1 = sm + p.dim0;        // D
    float* h2 = h1 + p.D;           // D   (SiLU(emb))
    const int row = blockIdx.x, tid = threadIdx.x;
    const float t = p.t[row];
    const int half = p.dim0 / 2;
    for (int i = tid; i < half; i += 256) {
        const float f = expf(-9.210340371976184f * (float)i / (float)half);   // ln(10000)
        const float a = t * f;
        e[i] = cosf(a);
        e[half + i] = sinf(a);
    }
    __syncthreads();
    for (int o = tid; o < p.D; o += 256) {
        float acc = p.b1[o];
        const float* w = p.w1 + (size_t)o * p.dim0;
        for (int k = 0; k < p.dim0; ++k) acc += w[k] * e[k];
        h1[o] = silu_f(acc);
    }
    __syncthreads();
    for (int o = tid; o < p.D; o += 256) {
        float acc = p.b2[o];
        const float* w = p.w2 + (size_t)o * p.D;
        for (int k = 0; k < p.D; ++k) acc += w[k] * h1[k];
        h2[o] = silu_f(acc);
    }
    __syncthreads();
    // projections: one wave per output, lanes stride the D inputs (coalesced weight rows)
    const int lane = tid & 63, wave = tid >> 6;
    for (int o = wave; o < p.total; o += 4) {
        const float* w = p.wp + (size_t)o * p.D;
        float acc = 0.f;
        for (int k = lane; k < p.D; k += 64) acc += w[k] * h2[k];
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
        if (lane == 0) p.out[(size_t)row * p.total + o] = acc + p.bp[o];
    }
}
int launch_temb(const TembParams& p, hipStream_t stream) {
    const size_t lds = (size_t)(p.dim0 + 2 * p.D) * sizeof(float);
    hipLaunchKernelGGL(temb_kernel, dim3(p.rows), dim3(256), lds, stream, p);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- scheduler steps (diffusers DDIMScheduler.step / DDPMScheduler.step [3P]) ------------------
__global__ void __launch_bounds__(256) sched_step_kernel(const SchedParams p) {
    float c0, c1, c2, c3, c4;
    const float* nz = p.noise;
    if (p.coef_table) {
        const int step = *p.step_ptr;
        const float* c = p.coef_table + 5 * step;
        c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3]; c4 = c[4];
        if (nz) nz += (size_t)step * p.noise_step_stride;
    } else {
        c0 = p.coef[0]; c1 = p.coef[1]; c2 = p.coef[2]; c3 = p.coef[3]; c4 = p.coef[4];
    }
    const bool use_noise = nz != nullptr && c4 != 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.n; i += (long long)gridDim.x * 256) {
        const float x = p.x[i], e = p.eps[i];
        const float x0 = (x - c1 * e) / c0;
        float prev = (p.mode == 0) ? c2 * x0 + c3 * e : c2 * x0 + c3 * x;
        if (use_noise) prev += c4 * nz[i];
        p.x_prev[i] = prev;
    }
}
int launch_sched_step(const SchedParams& p, hipStream_t stream) {
    const unsigned grid = (unsigned)((p.n + 255) / 256 > 2048 ? 2048 : (p.n + 255) / 256);
    hipLaunchKernelGGL(sched_step_kernel, dim3(grid), dim3(256), 0, stream, p);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

struct AddNoiseCoef { float sa[64]; float sb[64]; };
__global__ void __launch_bounds__(256) add_noise_kernel(const float* x0, const float* noise, AddNoiseCoef c, long long per,
                                                        long long n, float* out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / per);
        out[i] = c.sa[b] * x0[i] + c.sb[b] * noise[i];
    }
}
int launch_add_noise(const float* x0, const float* noise, const float* sa, const float* sb, int B, long long per,
                     float* out, hipStream_t stream) {
    RLDM_REQUIRE(B <= 64, "add_noise: batch > 64");
    AddNoiseCoef c;
    for (int i = 0; i < B; ++i) { c.sa[i] = sa[i]; c.sb[i] = sb[i]; }
    const long long n = per * B;
    const unsigned grid = (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(add_noise_kernel, dim3(grid), dim3(256), 0, stream, x0, noise, c, per, n, out);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

// DiagonalGaussianDistribution.sample (vae/sgm/modules/distributions/distributions.py:24-41), times `scale`
__global__ void __launch_bounds__(256) diag_gaussian_kernel(const float* mom, const float* noise, float scale, int z,
                                                            int spatial, long long n, float* out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long per = (long long)z * spatial;
        const long long b = i / per, r = i - b * per;
        const float mean = mom[b * 2 * per + r];
        float logvar = mom[b * 2 * per + per + r];
        logvar = fminf(fmaxf(logvar, -30.0f), 20.0f);
        out[i] = (mean + expf(0.5f * logvar) * noise[i]) * scale;
    }
}
int launch_diag_gaussian(const float* moments, const float* noise, float scale, int B, int z, int spatial, float* out,
                         hipStream_t stream) {
    const long long n = (long long)B * z * spatial;
    const unsigned grid = (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(diag_gaussian_kernel, dim3(grid), dim3(256), 0, stream, moments, noise, scale, z, spatial, n, out);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void step_counter_kernel(int* p, int set_to, int inc) {
    if (threadIdx.x == 0) *p = (inc ? *p + inc : set_to);
}
// in-graph timeline (tools/graph_trace.py): one thread writes the constant-rate (100 MHz) real-time counter
__global__ void stamp_kernel(unsigned long long* slot) {
    if (threadIdx.x == 0) *slot = wall_clock64();
}
int launch_stamp(unsigned long long* slot, hipStream_t stream) {
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, stream, slot);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_step_counter(int* step_ptr, int set_to, int increment, hipStream_t stream) {
    hipLaunchKernelGGL(step_counter_kernel, dim3(1), dim3(64), 0, stream, step_ptr, set_to, increment);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void __launch_bounds__(256) scale_kernel(const float* src, float* dst, float scale, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        dst[i] = src[i] * scale;
}
int launch_scale_f32(const float* src, float* dst, float scale, long long n, hipStream_t stream) {
    const unsigned grid = (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(scale_kernel, dim3(grid), dim3(256), 0, stream, src, dst, scale, n);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace rldm
