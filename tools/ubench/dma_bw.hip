// micro-benchmark: LDS-DMA (global_load_lds_dwordx4) streaming rate per CU from an L2-resident buffer that every
// workgroup reads (the conv kernels' weight stream).  Tuning aid, not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void lds_dma16s(const void* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
}
__global__ void __launch_bounds__(512, 1) k(float* out, const unsigned char* src, int stage_bytes, int nstage_src, int iters, int sync_every) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int pieces = stage_bytes / 1024;            // wave-instructions per stage
    for (int it = 0; it < iters; ++it) {
        const unsigned char* s = src + (size_t)(it % nstage_src) * stage_bytes;
        const unsigned dst = lds0 + (unsigned)((it & 1) * stage_bytes);
        for (int pc = wave; pc < pieces; pc += nw)
            lds_dma16s(s + (size_t)pc * 1024, lane * 16, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + pc * 1024)));
        if ((it + 1) % sync_every == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[blockIdx.x * blockDim.x + tid] = reinterpret_cast<float*>(smem)[tid];
}
int main() {
    float* out; unsigned char* w;
    const int nstage = 16;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&w, (size_t)nstage * 65536); hipMemset(w, 0, (size_t)nstage * 65536);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int blocks : {256, 64}) for (int threads : {256, 512}) for (int sb : {18432, 46080}) for (int sync_every : {1, 4}) {
        const int iters = 400;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 140 * 1024, 0, out, w, sb, nstage, 10, sync_every);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 140 * 1024, 0, out, w, sb, nstage, iters, sync_every);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("blocks %3d threads %3d stage %5d B sync/%d: %.0f ns/stage, %.1f GB/s per CU, %.2f TB/s total\n", blocks, threads, sb, sync_every,
               ms * 1e6 / iters, sb / (ms * 1e6 / iters), (double)sb * blocks / (ms * 1e6 / iters) / 1e3);
    }
    return 0;
}
