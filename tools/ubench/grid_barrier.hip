// What a dependent "seam" between two layers costs on MI355X, two ways (the question behind DESIGN.md 3.7):
//   (a) a kernel boundary: a chain of N launches captured in a hipGraph, each workgroup writes a small record and the next
//       launch's workgroup reads ANOTHER workgroup's record (so the data really crosses CUs / XCDs);
//   (b) the same N phases inside ONE persistent launch (one workgroup per CU), separated by a grid barrier that makes the
//       records visible: agent-scope release by lane 0 -> arrive on a counter -> relaxed sc1 poll with s_sleep -> agent-scope
//       acquire.  Two barrier forms: one monotonic counter, and XCD-hierarchical (per-XCC counter, the XCC's last arriver
//       goes to the top counter and publishes a per-XCC generation) as in MI355X_MICROARCH.md's price list.
// Every phase checks the value it reads (a stale read = a broken barrier) and the program prints the number of mismatches.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int REC = 64;        // floats per workgroup record (256 B)

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one phase of "work": write my record for phase ph, later read workgroup (wg + 37) % n's record of phase ph
__device__ __forceinline__ void write_rec(float* buf, int n, int ph, int wg) {
    if (threadIdx.x < REC) buf[((size_t)(ph & 1) * n + wg) * REC + threadIdx.x] = (float)(ph * 1000 + wg);
}
__device__ __forceinline__ int check_rec(const float* buf, int n, int ph, int wg) {
    const int src = (wg + 37) % n;
    int bad = 0;
    if (threadIdx.x < REC) bad = buf[((size_t)(ph & 1) * n + src) * REC + threadIdx.x] != (float)(ph * 1000 + src);
    return bad;
}

// (a) one launch per phase: reads the previous launch's records, writes its own
__global__ void phase_kernel(float* buf, int ph, int* errors) {
    const int n = gridDim.x, wg = blockIdx.x;
    if (ph > 0 && check_rec(buf, n, ph - 1, wg)) atomicAdd(errors, 1);
    write_rec(buf, n, ph, wg);
}

// (b1) persistent, single monotonic counter
__global__ void persistent_counter(float* buf, unsigned* counter, int nphases, int* errors) {
    const int n = gridDim.x, wg = blockIdx.x;
    int bad = 0;
    for (int ph = 0; ph < nphases; ++ph) {
        write_rec(buf, n, ph, wg);
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(ph + 1) * (unsigned)n;
            while (ld_relaxed(counter) < want) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        bad |= check_rec(buf, n, ph, wg);
    }
    if (bad) atomicAdd(errors, 1);
}

// (b2) persistent, XCD-hierarchical: xcc counters [8], top counter, per-xcc generation words [8]
__global__ void persistent_xcd(float* buf, unsigned* xcc_cnt, unsigned* top, unsigned* gen, int nphases, int* errors) {
    const int n = gridDim.x, wg = blockIdx.x;
    const int xcc = wg & 7;                                  // workgroup b runs on XCD b % 8 (observed placement)
    const unsigned per_xcc = (unsigned)((n - xcc + 7) / 8);
    int bad = 0;
    for (int ph = 0; ph < nphases; ++ph) {
        write_rec(buf, n, ph, wg);
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(&xcc_cnt[xcc * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1 == (unsigned)(ph + 1) * per_xcc) {     // last arriver of this XCC: go to the top counter
                __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (ld_relaxed(top) < (unsigned)(ph + 1) * 8u) __builtin_amdgcn_s_sleep(1);
                __hip_atomic_store(&gen[xcc * 32], (unsigned)(ph + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (ld_relaxed(&gen[xcc * 32]) < (unsigned)(ph + 1)) __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        bad |= check_rec(buf, n, ph, wg);
    }
    if (bad) atomicAdd(errors, 1);
}

int main() {
    const int n = 256, threads = 256, N = 200;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    float* buf; unsigned* ctr; int* err;
    CK(hipMalloc(&buf, (size_t)2 * n * REC * 4));
    CK(hipMalloc(&ctr, 4096 * 4));
    CK(hipMalloc(&err, 4));
    CK(hipMemset(err, 0, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms;
    int herr = 0;

    // (a) graph of N dependent launches
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(phase_kernel, dim3(n), dim3(threads), 0, st, buf, i, err);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("graph of %d dependent launches (256 WGs x 256 thr, 256-B record per WG): %.2f us per launch boundary, %d stale reads\n",
           N, ms * 1000.f / (5 * N), herr);

    // (b1)
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemsetAsync(ctr, 0, 4096 * 4, st));
        CK(hipMemsetAsync(err, 0, 4, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(persistent_counter, dim3(n), dim3(threads), 0, st, buf, ctr, N, err);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("persistent, one counter      : %.2f us per phase (barrier + record hand-off), %d workgroups saw a stale record\n",
           ms * 1000.f / N, herr);

    // (b2)
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemsetAsync(ctr, 0, 4096 * 4, st));
        CK(hipMemsetAsync(err, 0, 4, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(persistent_xcd, dim3(n), dim3(threads), 0, st, buf, ctr, ctr + 1024, ctr + 2048, N, err);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("persistent, XCD-hierarchical : %.2f us per phase (barrier + record hand-off), %d workgroups saw a stale record\n",
           ms * 1000.f / N, herr);
    return 0;
}
