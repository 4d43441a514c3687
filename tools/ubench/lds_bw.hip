// micro-benchmark: ds_read_b128 throughput per CU, alone and interleaved with MFMAs (tuning aid, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>   // 0: reads only, 1: mfma only, 2: both interleaved (16 reads + 16 mfma per iteration)
__global__ void __launch_bounds__(512, 1) k(float* out, int iters, int stride_bytes, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 40000; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)i * 1e-9f;
    __syncthreads();
    const int l31 = lane & 31, kh = lane >> 5;
    const unsigned char* base = smem + l31 * stride_bytes + kh * 16 + (tid >> 6) * 64;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 f[16];
    for (int j = 0; j < 16; ++j) f[j] = *reinterpret_cast<const bf16x8*>(base + j * 32);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE != 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) f[j] = *reinterpret_cast<const bf16x8*>(base + ((it & 3) * 4608) + j * 32 + (j >> 2) * 4608 * 0);
        }
        if (MODE != 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[j], f[(j + 1) & 15], acc[j & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("" ::"v"(f[j]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int j = 0; j < 16; ++j) s += (float)f[j][0];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int nthreads : {256, 512}) for (int stride : {144, 128, 272}) {
        for (int mode = 0; mode < 3; ++mode) {
            auto kern = mode == 0 ? k<0> : (mode == 1 ? k<1> : k<2>);
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(kern, dim3(256), dim3(nthreads), 160 * 1024, 0, out, 10, stride, cyc);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(nthreads), 160 * 1024, 0, out, iters, stride, cyc);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double waves = nthreads / 64.0;
            printf("threads %d stride %d mode %d: %.1f us, %.1f cycles/iter (memtime), per-CU ds_read_b128: %.2f cyc each, mfma: %.1f cyc per mfma per SIMD\n",
                   nthreads, stride, mode, ms * 1e3, (double)c / iters, (double)c / iters / (16 * waves), (double)c / iters / (16 * waves / 4));
        }
    }
    return 0;
}
