// Issue rate of the transcendental instructions on gfx950: cycles per wave64 instruction for v_exp_f32, v_exp_f16,
// v_rcp_f32, v_exp_legacy_f32 and a plain v_fma_f32, one wave per SIMD (256 threads) and two (512).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/exp_rate.hip -o gpurun_out/exp_rate ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) { REP16(asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));) }
            if (KIND == 1) { REP16(asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));) }
            if (KIND == 2) { REP16(asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));) }
            if (KIND == 3) { REP16(asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(a[i]));) }
            if (KIND == 4) { REP16(asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));) }
            if (KIND == 5) { REP16(asm volatile("v_rcp_f16 %0, %0" : "+v"(a[i]));) }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* cyc) {
    for (int threads : {256, 512}) {
        const int iters = 200;
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double n = (double)iters * 8 * 16;          // instructions per wave
        printf("%-18s %d thr: %.2f ticks / instr / wave,  %.2f ns / instr / wave (wall)\n", name, threads, c / n, ms * 1e6 / n);
    }
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 8);
    run<4>("v_fma_f32", out, cyc);
    run<0>("v_exp_f32", out, cyc);
    run<1>("v_exp_f16", out, cyc);
    run<3>("v_exp_legacy_f32", out, cyc);
    run<2>("v_rcp_f32", out, cyc);
    run<5>("v_rcp_f16", out, cyc);
    return 0;
}
