// What a kernel boundary costs inside a captured graph on MI355X: a chain of N dependent launches of one kernel, replayed
// from a hipGraph, timed with events; per-launch time for kernels that do (almost) nothing but differ in grid size, LDS
// footprint and in how much dirty data they leave in the XCDs' L2s (plain or nontemporal stores).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_floor.hip -o gpurun_out/launch_floor ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// every thread stores `per_thread` 16-byte pieces (coalesced across the workgroup), reading them first if `rd`
template <int NT_STORE>
__global__ void k(uint4* buf, int per_thread, int rd, int spin) {
    extern __shared__ unsigned char smem[];
    if (spin) {                                           // a little LDS traffic so the allocation is real
        reinterpret_cast<int*>(smem)[threadIdx.x] = threadIdx.x;
        __syncthreads();
    }
    uint4* p = buf + (size_t)blockIdx.x * blockDim.x * per_thread + threadIdx.x;
    for (int i = 0; i < per_thread; ++i) {
        uint4 v = make_uint4(i, i, i, i);
        if (rd) v = p[(size_t)i * blockDim.x];
        v.x += 1;
        if (NT_STORE) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            u32x4 w = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(p + (size_t)i * blockDim.x));
        }
        else p[(size_t)i * blockDim.x] = v;
    }
}

// spins until `cycles` ticks of s_memtime have passed (and, per thread, keeps the VALU / MFMA busy if asked): per-launch wall
// time - launch floor = cycles / f gives the clock the shader counter runs at under that load
__global__ void spin_k(unsigned long long cycles, int load, float* sink) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float a = threadIdx.x;
    f32x16 acc = {0};
    s16x8 x = {1, 2, 3, 4, 5, 6, 7, 8};
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) {
        if (load == 1)
            for (int i = 0; i < 64; ++i) a = a * 1.0001f + 0.5f;
        if (load == 2)
            for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, acc, 0, 0, 0);
    }
    if (a + acc[0] == 12345.678f) *sink = a;
}

static int run_spin(const char* name, int grid, int threads, unsigned long long cycles, int load, float* sink, hipStream_t st) {
    const int N = 100;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin_k, dim3(grid), dim3(threads), 0, st, cycles, load, sink);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (5 * N);
    printf("%-44s grid %4d x %4d  spin %6llu ticks : %6.2f us per launch  => counter at %.2f GHz (1.6 us floor taken off)\n", name, grid,
           threads, cycles, us, cycles / (us - 1.6) / 1e3);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return 0;
}

template <int NT_STORE>
static int run(const char* name, int grid, int threads, size_t lds, int per_thread, int rd, uint4* buf, hipStream_t st) {
    const int N = 200;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<NT_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k<NT_STORE>, dim3(grid), dim3(threads), lds, st, buf, per_thread, rd, lds > 0);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double mb = (double)grid * threads * per_thread * 16 / 1e6;
    printf("%-44s grid %4d x %4d  lds %6zu  %7.2f MB %s %s : %6.2f us per launch\n", name, grid, threads, lds, mb,
           rd ? "rd+wr" : "wr   ", NT_STORE ? "nontemporal" : "plain      ", ms * 1e3 / (5 * N));
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return 0;
}

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint4* buf;
    CK(hipMalloc(&buf, (size_t)256 << 20));
    CK(hipMemset(buf, 0, (size_t)256 << 20));
    run<0>("one thread", 1, 1, 0, 0, 0, buf, st);
    run<0>("256 wg, no work", 256, 512, 0, 0, 0, buf, st);
    run<0>("256 wg, 100 KB LDS", 256, 512, 100 << 10, 0, 0, buf, st);
    run<0>("128 wg, 100 KB LDS", 128, 512, 100 << 10, 0, 0, buf, st);
    run<0>("512 wg, 64 KB LDS", 512, 512, 64 << 10, 0, 0, buf, st);
    run<0>("256 wg x 1024 thr, 50 KB LDS", 256, 1024, 50 << 10, 0, 0, buf, st);
    // level-3 size: 0.5 MB out; level 0: 16.8 MB out (64 KB per workgroup = 8 pieces per thread)
    run<0>("128 wg, 4 KB out each", 128, 512, 100 << 10, 1, 0, buf, st);          // 8 KB actually (512 x 16 B)
    run<1>("128 wg, 4 KB out each", 128, 512, 100 << 10, 1, 0, buf, st);
    run<0>("256 wg, 64 KB out each", 256, 512, 100 << 10, 8, 0, buf, st);
    run<1>("256 wg, 64 KB out each", 256, 512, 100 << 10, 8, 0, buf, st);
    run<0>("256 wg, 64 KB in + out each", 256, 512, 100 << 10, 8, 1, buf, st);
    run<1>("256 wg, 64 KB in + out each", 256, 512, 100 << 10, 8, 1, buf, st);
    run<0>("256 wg, 256 KB in + out each", 256, 512, 100 << 10, 32, 1, buf, st);
    run<1>("256 wg, 256 KB in + out each", 256, 512, 100 << 10, 32, 1, buf, st);
    float* sink = reinterpret_cast<float*>(buf);
    run_spin("spin, idle lanes, 1 wg", 1, 64, 24000, 0, sink, st);
    run_spin("spin, idle lanes, 256 wg", 256, 512, 24000, 0, sink, st);
    run_spin("spin, idle lanes, 256 wg, short", 256, 512, 6000, 0, sink, st);
    run_spin("spin, VALU busy, 256 wg", 256, 512, 24000, 1, sink, st);
    run_spin("spin, MFMA busy, 256 wg", 256, 512, 24000, 2, sink, st);
    run_spin("spin, MFMA busy, 256 wg, long", 256, 512, 240000, 2, sink, st);
    return 0;
}
