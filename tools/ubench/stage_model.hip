// micro-benchmark: model of one conv main-loop stage (tuning aid, not product code)
//   per stage and wave: barrier, 8 "weight" ds_read_b128, 16 MFMA 32x32x16 with 3 LDS-DMA pieces and 8 "pixel"
//   ds_read_b128 (next stage) interleaved in the MFMA shadows.  Variants toggle the interleave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

template <int MODE>   // 0: reads up front, DMA up front (v3 style); 1: DMA + next-stage pixel reads interleaved with MFMAs
__global__ void __launch_bounds__(512, 1) k(float* out, const unsigned char* wsrc, int iters, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 40000; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)i * 1e-9f;
    __syncthreads();
    const int l31 = lane & 31, kh = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned char* wb = smem + l31 * 144 + kh * 16 + (wave >> 2) * 4608 * 2;
    const unsigned char* xb = smem + 60000 + l31 * 144 + kh * 16 + (wave & 3) * 4608 * 2;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 wf[4][2], xa[4][2], xn[4][2];
    for (int ks = 0; ks < 4; ++ks) for (int j = 0; j < 2; ++j) xa[ks][j] = *reinterpret_cast<const bf16x8*>(xb + ks * 32 + j * 4608);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned char* src = wsrc + (size_t)(it % 18) * 18432;
        const unsigned dst = lds0 + (unsigned)((it % 3) * 18432);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[ks][j] = *reinterpret_cast<const bf16x8*>(wb + ((it % 3) * 18432) + ks * 32 + j * 4608);
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                int piece = i * 512 + wave * 64; piece = piece > 1088 ? 1088 : piece;
                lds_dma16(src + (size_t)(piece + lane) * 16, dst + piece * 16);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) xa[ks][j] = *reinterpret_cast<const bf16x8*>(xb + (it & 7) * 144 + ks * 32 + j * 4608);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][a >> 1], xa[ks][a & 1], acc[a], 0, 0, 0);
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][a >> 1], xa[ks][a & 1], acc[a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ks < 3) {
                    int piece = ks * 512 + wave * 64; piece = piece > 1088 ? 1088 : piece;
                    lds_dma16(src + (size_t)(piece + lane) * 16, dst + piece * 16);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) xn[ks][j] = *reinterpret_cast<const bf16x8*>(xb + ((it + 1) & 7) * 144 + ks * 32 + j * 4608);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) xa[ks][j] = xn[ks][j];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc; unsigned char* w;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8); hipMalloc(&w, 18 * 18432 + 4096); hipMemset(w, 0, 18 * 18432 + 4096);
    const int iters = 1800;
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 2; ++mode) {
        auto kern = mode == 0 ? k<0> : k<1>;
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 160 * 1024, 0, out, w, 10, cyc);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 160 * 1024, 0, out, w, iters, cyc);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("mode %d: %.1f us total, %.0f ns/stage, %.0f ticks/stage -> %.0f TF/s-equivalent\n", mode, ms * 1e3, ms * 1e6 / iters,
               (double)c / iters, 256.0 * 8 * 16 * 32768 / (ms * 1e6 / iters) / 1e3);
    }
    return 0;
}
