// Cluster seam of the persistent trunk launch (trunk.hip), shared by the phase bodies (conv_small_body.h, attention_body.h).
#pragma once
#include "kernels.h"

namespace rldm {

// ---- cluster seam of the persistent trunk (tools/ubench/xcd_cluster.hip: 1.4 us for 16 workgroups of one XCD) --------------------
// The workgroups of one image sit on ONE XCD (block -> XCD round-robin; checked by the host once per plan), so a plain store that
// has been acknowledged (s_waitcnt vmcnt(0)) is in the L2 they share, and a load that does not hit a stale line of the reader's L1
// sees it: the consumer invalidates its CU's vector L1 once, behind the wait (buffer_inv sc0: the L1 only -- sc1 would also walk the
// L2 and made the whole step 30 % slower), and then uses ordinary cached loads.
constexpr int kTrunkPrefetch = RLDM_TRUNK_PREFETCH;
#ifndef RLDM_TRUNK_CL_PREFETCH
#define RLDM_TRUNK_CL_PREFETCH 0    /* ... by the multi-tile clusters of trunk variant 1 (round 4 experiment: 12 fits since the kernel sits at 196 VGPRs,
                                       measured 230.8 -> 229.6 img/s -- 252 VGPRs, 144 spilled SGPRs, and the K loop's wait is bandwidth, not the first round trip) */
#endif
constexpr int kClusterPrefetch = RLDM_TRUNK_CL_PREFETCH;            // weight fragments per wave requested one phase ahead (registers carried across phases)
struct TrunkSeam {
    unsigned* counter;              // arrivals of this image's cluster (monotonic over the launch; zeroed by the launch before)
    unsigned wait_for;              // arrivals that must have happened before this phase reads activations
    int has_wait;                   // (0: the launch's first phase -- its inputs crossed a kernel boundary)
    unsigned next_rec;              // lane l: word l of the next phase's record (kernels.h TrunkWord), or 0 behind the last phase
    int next_rank_kg;               // this wave's weight stream of a layer = (channel tile * k-groups + k-group)
    int rank, ranks;                // this workgroup's place in the image's cluster
    int* error;                     // device flag: a bounded poll gave up (the host refuses the plan's results)
    unsigned long long* ts;         // ABLATE builds: 16 stamp slots of this phase (workgroup 0) or null
    // the phase's time-embedding row (the launch's arguments, not the phase record: the table belongs to the caller of the plan)
    const float* temb;              // table + this layer's channel offset, or null
    const int* step_ptr;
    int temb_rows_per_step, temb_per_sample, temb_ld;
};
#ifndef RLDM_TRUNK_PLAIN_LOADS
#define RLDM_TRUNK_PLAIN_LOADS 1   // (0: L1-bypassing nontemporal loads instead of one L1 invalidate per phase -- measured 0.7 % slower)
#endif
#ifndef RLDM_TRUNK_INV
#define RLDM_TRUNK_INV "buffer_inv sc0"
#endif
__device__ __forceinline__ void trunk_wait(const TrunkSeam& s, int tid) {
    if (tid == 0 && s.has_wait && !RLDM_EXP_NOWAIT) {
        int polls = 0;
        while ((int)(__hip_atomic_load(s.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - s.wait_for) < 0) {   // (wrap-safe)
            __builtin_amdgcn_s_sleep(1);
            if (++polls > (1 << 22)) { *s.error = 1; break; }      // bounded: a protocol error must not hang the device
        }
    }
    // (a barrier that waits for LDS only: global requests of the phase issued ahead of the wait -- the weight ring of a conv_stream
    //  phase -- stay in flight across it; the "memory" clobber keeps the phase's loads behind it)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#if RLDM_TRUNK_PLAIN_LOADS
    // the CU's vector L1 is invalidated ONCE per phase, behind the wait: the phase's activation loads are then ordinary cached loads
    // (first touch from the XCD's L2, re-reads of a halo line from the L1) instead of L1-bypassing ones
    asm volatile(RLDM_TRUNK_INV ::: "memory");
#endif
}
__device__ __forceinline__ void trunk_arrive(const TrunkSeam& s, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this thread's stores are in the cluster's L2
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(s.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ---- L2 warm-up of the NEXT phase's weights (round 5) ------------------------------------------------------------------------------
// With image <-> XCD every XCD pulls every layer's weights through its own L2, and a layer's first touch is an Infinity-Cache round trip
// (~4 k cycles) per ring slot: the K loop of a 64x4 cluster phase ran 17-22 k cycles against 5.6 k of matrix-pipe time
// (profiles/round5_trunk_timeline_before.txt).  Each workgroup of the cluster therefore touches its 1/ranks share of the next phase's
// weight bytes -- one dword per 128-byte line, kWarmLoads loads per thread -- once its own K loop has ended: the loads land under the
// epilogue (the arrive waits for vmcnt(0) anyway), nothing younger than them is waited for before that, and the next phase's ring
// then hits the L2.  The values are kept alive until trunk_warm_done (an unused load result would be a register the compiler reuses
// while the load is still in flight).
#ifndef RLDM_TRUNK_WARM
#define RLDM_TRUNK_WARM 3           /* loads per thread (0: off): 16 ranks x 512 threads x 3 x 128 B = 3 MiB >= the largest layer */
#endif
#ifndef RLDM_TRUNK_WARM0
#define RLDM_TRUNK_WARM0 0          /* 1: the image-owning phases of variant 0 warm their successor's weights too */
#endif
constexpr int kWarmLoads = RLDM_TRUNK_WARM;
struct TrunkWarm { unsigned v[kWarmLoads > 0 ? kWarmLoads : 1]; };
__device__ __forceinline__ void trunk_warm_next(const TrunkSeam& s, int tid, int nthreads, TrunkWarm& w) {
    if constexpr (kWarmLoads > 0) {
        const unsigned nr = s.next_rec;
        const unsigned bytes = (unsigned)__builtin_amdgcn_readlane((int)nr, TW_WBYTES);
        const unsigned long long wp = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)nr, TW_WPK + 1) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)nr, TW_WPK);
        typedef __attribute__((address_space(1))) const unsigned char* global_bytes_t;
        const unsigned char* base = (const unsigned char*)(global_bytes_t)wp;
        const unsigned lines = bytes >> 7;
        const unsigned share = (lines + (unsigned)s.ranks - 1u) / (unsigned)s.ranks;
        const unsigned first = (unsigned)s.rank * share;
#pragma unroll
        for (int j = 0; j < kWarmLoads; ++j) {
            const unsigned l = (unsigned)tid + (unsigned)(j * nthreads);
            w.v[j] = 0u;
            if (l < share && first + l < lines) w.v[j] = *reinterpret_cast<const unsigned*>(base + ((size_t)(first + l) << 7));
        }
    }
}
__device__ __forceinline__ void trunk_warm_done(TrunkWarm& w) {
    if constexpr (kWarmLoads > 0) {
#pragma unroll
        for (int j = 0; j < kWarmLoads; ++j) asm volatile("" :: "v"(w.v[j]));
    }
}

// activation loads: past the L1 inside the trunk (another CU of the cluster wrote the line during this launch)
template <bool BYPASS> __device__ __forceinline__ uint4 ld_act16(const void* p) {
    if constexpr (BYPASS && !RLDM_TRUNK_PLAIN_LOADS) {
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    } else {
        return *reinterpret_cast<const uint4*>(p);
    }
}
template <bool BYPASS> __device__ __forceinline__ float2 ld_act8(const float2* p) {
    if constexpr (BYPASS && !RLDM_TRUNK_PLAIN_LOADS) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x2_t*>(p));
        return make_float2(v.x, v.y);
    } else {
        return *p;
    }
}

// 16 bytes of bf16 activations as an MFMA operand
template <bool BYPASS> __device__ __forceinline__ bf16x8 ld_act_frag(const bf16_t* p) {
    if constexpr (BYPASS && !RLDM_TRUNK_PLAIN_LOADS) return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
    else return *reinterpret_cast<const bf16x8*>(p);
}

}  // namespace rldm
