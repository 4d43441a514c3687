// Can a dependent launch hide its boundary + prologue behind the previous kernel on MI355X?  
//   (a) the usual chain: N kernels captured on ONE stream -- every launch waits for the previous one to drain (barrier bit), then
//       fetches its arguments / weights / addresses, then reads the previous kernel's output;
//   (b) the same N kernels captured on TWO streams alternately (even / odd), so kernel n + 1 is not ordered behind kernel n by the
//       hardware: its workgroups become resident as CUs free up, run their independent prologue (weights -> registers), and then
//       wait in software for kernel n: every workgroup of n ends with `s_waitcnt vmcnt(0)` + one agent-scope atomic add on n's
//       counter; n + 1 polls it (relaxed sc1 loads + s_sleep).  The data crosses with sc0 sc1 (write-through) stores and sc1 loads:
//       no cache maintenance instruction on either side (MI355X_MICROARCH.md, inter-workgroup visibility, valid forms).
// Every workgroup keeps LDS_BYTES of LDS (one workgroup per CU, like the conv kernels) and grids are <= the CU count, so a waiting
// workgroup never occupies a slot the kernel it waits for still needs.  Every value read is checked.
// build + run:  hipcc --offload-arch=gfx950 -O3 tools/ubench/soft_dep.hip -o /tmp/soft_dep && /tmp/soft_dep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int NT = 512;
constexpr int ACT_PER_WG = 4096;          // floats each workgroup writes (16 KB) and reads from another workgroup
constexpr int W_PER_WG = 8192;            // "weights": floats read in the prologue (32 KB), independent of the previous kernel
constexpr int LDS_BYTES = 100 * 1024;

typedef float f32x4 __attribute__((ext_vector_type(4)));     // (HIP's float4 is a struct: not an asm register operand)
__device__ __forceinline__ float4 ld_sc1(const float4* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_sc1(float4* p, float4 v) {
    const f32x4 u = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(u) : "memory");
}

// work: `spin` dependent FMAs per element (a few microseconds of "K loop")
template <bool SOFT>
__global__ void __launch_bounds__(NT, 1) layer(const float* __restrict__ w, const float* act_in, float* act_out, unsigned* flags,
                                               int idx, int nwg_prev, int spin, int* errors) {
    extern __shared__ float smem[];
    const int wg = blockIdx.x, n = gridDim.x, tid = threadIdx.x;
    // ---- prologue: independent of the previous kernel (weights) ----
    float4 wv[W_PER_WG / 4 / NT];
#pragma unroll
    for (int i = 0; i < W_PER_WG / 4 / NT; ++i)
        wv[i] = reinterpret_cast<const float4*>(w + (size_t)((wg * 7 + idx) % n) * W_PER_WG)[tid + i * NT];
    float wsum = 0.f;
#pragma unroll
    for (int i = 0; i < W_PER_WG / 4 / NT; ++i) wsum += wv[i].x + wv[i].y + wv[i].z + wv[i].w;
    smem[tid] = wsum;
    // ---- wait for the previous layer ----
    if (SOFT && idx > 0) {
        if (tid == 0) {
            int polls = 0;                               // bounded: a protocol error must not hang the device
            while (__hip_atomic_load(&flags[idx - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nwg_prev) {
                __builtin_amdgcn_s_sleep(1);
                if (++polls > (1 << 20)) { atomicAdd(errors, 1000); break; }
            }
        }
    }
    __syncthreads();
    // ---- dependent phase: read another workgroup's output of the previous layer, check, work, write ----
    const int src = (wg + 37) % n;
    int bad = 0;
    float4 a[ACT_PER_WG / 4 / NT];
    if (idx > 0) {
#pragma unroll
        for (int i = 0; i < ACT_PER_WG / 4 / NT; ++i) {
            const float4* p = reinterpret_cast<const float4*>(act_in + (size_t)src * ACT_PER_WG) + tid + i * NT;
            a[i] = SOFT ? ld_sc1(p) : *p;
        }
        if (SOFT) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0].x), "+v"(a[0].y), "+v"(a[0].z), "+v"(a[0].w), "+v"(a[1].x), "+v"(a[1].y), "+v"(a[1].z), "+v"(a[1].w)::"memory");
#pragma unroll
        for (int i = 0; i < ACT_PER_WG / 4 / NT; ++i) {
            const float want = (float)((idx - 1) * 1000 + src);
            bad |= (a[i].x != want) | (a[i].y != want) | (a[i].z != want) | (a[i].w != want);
        }
    }
    float acc = smem[(tid + 1) % NT] * 1e-30f;
    for (int s = 0; s < spin; ++s) acc = acc * 1.0000001f + 1e-30f;
    const float outv = (float)(idx * 1000 + wg) + acc * 0.f;
#pragma unroll
    for (int i = 0; i < ACT_PER_WG / 4 / NT; ++i) {
        float4* p = reinterpret_cast<float4*>(act_out + (size_t)wg * ACT_PER_WG) + tid + i * NT;
        const float4 v = make_float4(outv, outv, outv, outv);
        if (SOFT) st_sc1(p, v); else *p = v;
    }
    if (bad) atomicAdd(errors, 1);
    if (SOFT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's write-through stores have been acknowledged
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&flags[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv) {
    const int N = 100;
    int dev_cus = 256;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    dev_cus = prop.multiProcessorCount;
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    float *w, *act[3];
    unsigned* flags;
    int* err;
    CK(hipMalloc(&w, (size_t)dev_cus * W_PER_WG * 4));
    CK(hipMemset(w, 0, (size_t)dev_cus * W_PER_WG * 4));
    for (int i = 0; i < 3; ++i) CK(hipMalloc(&act[i], (size_t)dev_cus * ACT_PER_WG * 4));
    CK(hipMalloc(&flags, N * 4));
    CK(hipMalloc(&err, 4));
    CK(hipMemset(err, 0, 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipEvent_t e0, e1, fork, join;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    printf("%d CUs; %d layers per graph; each workgroup: %d KB prologue read, %d KB hand-off, %d KB LDS\n", dev_cus, N,
           W_PER_WG * 4 / 1024, ACT_PER_WG * 4 / 1024, LDS_BYTES / 1024);
    for (int grid : {dev_cus, dev_cus / 2}) {
        for (int spin : {0, 2000, 8000}) {
            float ms_a = 0.f, ms_b = 0.f;
            int herr_a = 0, herr_b = 0;
            // (a) one stream
            {
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
                for (int i = 0; i < N; ++i)
                    hipLaunchKernelGGL(layer<false>, dim3(grid), dim3(NT), LDS_BYTES, s1, w, act[(i + 2) % 3], act[i % 3], flags, i, grid, spin, err);
                CK(hipStreamEndCapture(s1, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, s1));
                CK(hipStreamSynchronize(s1));
                CK(hipEventRecord(e0, s1));
                for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s1));
                CK(hipEventRecord(e1, s1));
                CK(hipStreamSynchronize(s1));
                CK(hipEventElapsedTime(&ms_a, e0, e1));
                CK(hipMemcpy(&herr_a, err, 4, hipMemcpyDeviceToHost));
                CK(hipMemset(err, 0, 4));
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
            // (b) two streams, software dependency
            {
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
                CK(hipMemsetAsync(flags, 0, N * 4, s1));
                // layer 0 runs alone (hardware edge 0 -> 1): two kernels released at the same moment could interleave their
                // workgroups, and a waiting workgroup of layer 1 would then hold a CU layer 0 still needs
                for (int i = 0; i < N; ++i) {
                    hipLaunchKernelGGL(layer<true>, dim3(grid), dim3(NT), LDS_BYTES, (i & 1) ? s2 : s1, w, act[(i + 2) % 3], act[i % 3], flags, i, grid, spin, err);
                    if (i == 0) {
                        CK(hipEventRecord(fork, s1));
                        CK(hipStreamWaitEvent(s2, fork, 0));
                    }
                }
                CK(hipEventRecord(join, s2));
                CK(hipStreamWaitEvent(s1, join, 0));
                CK(hipStreamEndCapture(s1, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, s1));
                CK(hipStreamSynchronize(s1));
                CK(hipEventRecord(e0, s1));
                for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s1));
                CK(hipEventRecord(e1, s1));
                CK(hipStreamSynchronize(s1));
                CK(hipEventElapsedTime(&ms_b, e0, e1));
                CK(hipMemcpy(&herr_b, err, 4, hipMemcpyDeviceToHost));
                CK(hipMemset(err, 0, 4));
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
            printf("grid %3d, spin %5d: one stream %.2f us / layer (%d bad) | two streams + software dependency %.2f us / layer (%d bad)\n",
                   grid, spin, ms_a * 1000.f / (5 * N), herr_a, ms_b * 1000.f / (5 * N), herr_b);
            fflush(stdout);
        }
    }
    return 0;
}
