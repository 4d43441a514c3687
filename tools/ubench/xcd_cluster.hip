// What an all-gather between the 16 workgroups that would share one image of the 32x2 / 64x4 levels costs on MI355X when they
// sit on ONE XCD (the persistent "trunk" question, asked again with XCD-local clusters instead of a grid barrier).
// 256 workgroups x 512 threads, one per CU (100 KB of LDS); workgroup b runs on XCD b % 8 (observed placement; checked here with
// s_getreg XCC_ID), so cluster = (XCD, half of its 32 workgroups): 16 clusters of 16.  One phase = a layer seam of such a kernel:
// every workgroup publishes its 2 KB slice (64 pixels x 16 channels bf16), arrives on its cluster's counter, waits for the other
// 15, and reads the cluster's 32 KB into LDS; every word is checked.
//   plain : plain 16-byte stores -> s_waitcnt vmcnt(0) -> relaxed agent atomic add; consumer: relaxed sc1 poll -> sc1 loads
//           (the stores stay in the XCD's L2, sc1 loads bypass the reader's L1: valid only if the cluster really shares an L2)
//   wt    : sc0 sc1 (write-through) stores -> vmcnt(0) -> atomic add; consumer: poll -> sc1 loads (valid at any placement)
//   launch: the same exchange across a kernel boundary (graph of dependent launches), for reference
// build + run:  hipcc --offload-arch=gfx950 -O3 tools/ubench/xcd_cluster.hip -o /tmp/xcd_cluster && /tmp/xcd_cluster
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int NT = 512, CL = 16, SLICE = 2048, LDS_BYTES = 100 * 1024;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 ld_sc1(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_wt(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }

__device__ __forceinline__ void cluster_ids(int wg, int* cluster, int* rank) {
#ifdef CROSS_XCD      // 16 consecutive block ids = one cluster: two workgroups on each of the 8 XCDs
    *cluster = wg >> 4;
    *rank = wg & 15;
#else
    const int x = wg & 7, j = wg >> 3;
    *cluster = x * 2 + (j >> 4);
    *rank = j & 15;
#endif
}
__device__ __forceinline__ unsigned pattern(int phase, int cluster, int rank, int i) { return (unsigned)(phase * 65536 + cluster * 1024 + rank * 32 + (i & 31)); }

template <int MODE>   // 0 plain, 1 write-through
__global__ void __launch_bounds__(NT, 1) persistent(unsigned char* buf, unsigned* counters, int nphases, int* errors, int* misplaced) {
    extern __shared__ unsigned char smem[];
    const int wg = blockIdx.x, tid = threadIdx.x;
    int cluster, rank;
    cluster_ids(wg, &cluster, &rank);
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));       // XCC_ID field: bits 3:0
    // (the round-robin pointer carries over from earlier dispatches: workgroup b sits on XCD (b + k) % 8 for a launch-dependent k,
    //  so "same b % 8" = "same XCD" is what is checked, through workgroup 0's id)
    __shared__ unsigned xcc0;
    if (tid == 0) {
        if (wg == 0) misplaced[1] = (int)xcc;
    }
    (void)xcc0;
    int bad = 0;
    for (int ph = 0; ph < nphases; ++ph) {
        unsigned char* base = buf + (size_t)((ph & 1) * 16 + cluster) * CL * SLICE;
        if (tid < SLICE / 16) {                  // publish my slice: 128 threads x 16 bytes
            const unsigned v = pattern(ph, cluster, rank, tid);
            const u32x4 val = {v, v + 1, v + 2, v + 3};
            void* dst = base + rank * SLICE + tid * 16;
            if (MODE == 0) *reinterpret_cast<u32x4*>(dst) = val; else st_wt(dst, val);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(&counters[cluster * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int polls = 0;
            while (__hip_atomic_load(&counters[cluster * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(ph + 1) * CL) {
                __builtin_amdgcn_s_sleep(1);
                if (++polls > (1 << 22)) { atomicAdd(errors, 1000000); break; }
            }
        }
        __syncthreads();
        // gather the cluster's 32 KB: 512 threads x 4 x 16 bytes, into LDS
        u32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ld_sc1(base + (size_t)(tid + i * NT) * 16);
        // (the asm loads are invisible to the compiler's wait insertion: the wait names the registers, so no use moves above it)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + i * NT, r = q / (SLICE / 16), t = q % (SLICE / 16);
            bad |= v[i].x != pattern(ph, cluster, r, t) || v[i].w != pattern(ph, cluster, r, t) + 3;
            *reinterpret_cast<u32x4*>(smem + q * 16) = v[i];
        }
        __syncthreads();
    }
    if (bad) atomicAdd(errors, 1);
}

__global__ void __launch_bounds__(NT, 1) per_launch(unsigned char* buf, int ph, int* errors) {
    extern __shared__ unsigned char smem[];
    const int wg = blockIdx.x, tid = threadIdx.x;
    int cluster, rank;
    cluster_ids(wg, &cluster, &rank);
    int bad = 0;
    if (ph > 0) {
        const unsigned char* base = buf + (size_t)(((ph - 1) & 1) * 16 + cluster) * CL * SLICE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + i * NT, r = q / (SLICE / 16), t = q % (SLICE / 16);
            const u32x4 v = *reinterpret_cast<const u32x4*>(base + (size_t)q * 16);
            bad |= v.x != pattern(ph - 1, cluster, r, t);
            *reinterpret_cast<u32x4*>(smem + q * 16) = v;
        }
    }
    __syncthreads();
    unsigned char* base = buf + (size_t)((ph & 1) * 16 + cluster) * CL * SLICE;
    if (tid < SLICE / 16) {
        const unsigned v = pattern(ph, cluster, rank, tid);
        const u32x4 val = {v, v + 1, v + 2, v + 3};
        *reinterpret_cast<u32x4*>(base + rank * SLICE + tid * 16) = val;
    }
    if (bad) atomicAdd(errors, 1);
}

int main() {
    const int n = 256, N = 200;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    unsigned char* buf; unsigned* ctr; int *err, *mis;
    CK(hipMalloc(&buf, (size_t)2 * 16 * CL * SLICE));
    CK(hipMalloc(&ctr, 16 * 32 * 4));
    CK(hipMalloc(&err, 4)); CK(hipMalloc(&mis, 8));
    CK(hipMemset(err, 0, 4)); CK(hipMemset(mis, 0, 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(persistent<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(persistent<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(per_launch), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    int herr = 0, hmis = 0;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(ctr, 0, 16 * 32 * 4, st));
            CK(hipMemsetAsync(err, 0, 4, st));
            CK(hipEventRecord(e0, st));
            if (mode == 0) hipLaunchKernelGGL(persistent<0>, dim3(n), dim3(NT), LDS_BYTES, st, buf, ctr, N, err, mis);
            else hipLaunchKernelGGL(persistent<1>, dim3(n), dim3(NT), LDS_BYTES, st, buf, ctr, N, err, mis);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&hmis, mis + 1, 4, hipMemcpyDeviceToHost));
        printf("persistent, %-5s stores: %.2f us per seam (publish 2 KB + arrive + wait for 15 + gather 32 KB), %d workgroups saw a wrong word (workgroup 0 on XCC %d)\n",
               mode == 0 ? "plain" : "wt", ms * 1000.f / N, herr, hmis);
    }
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipMemsetAsync(err, 0, 4, st));
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(per_launch, dim3(n), dim3(NT), LDS_BYTES, st, buf, i, err);
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
        printf("graph of dependent launches   : %.2f us per seam (gather 32 KB + publish 2 KB per launch), %d wrong\n", ms * 1000.f / (5 * N), herr);
    }
    return 0;
}
