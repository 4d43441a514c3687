// Persistent "trunk" launch for the levels whose images fit one workgroup tile (<= 64 pixels: the 32x2 level and the mid block
// of the KITTI network): a run of consecutive convs -- ResnetBlock conv1 / conv2, attention output projections -- executes as the
// PHASES of one launch instead of one launch each (ldm/pipelines.py:353-362 -> UNet2DModel.forward).
//
// Why this is legal without a grid barrier: with producer-side GroupNorm (conv_small_body.h) a workgroup (image b, 32 output
// channels) needs, for the next layer, only what the OTHER channel tiles of the SAME image wrote -- GroupNorm statistics never
// leave the image.  So the dependence between two layers is a cluster of N / 32 workgroups per image, not the grid.  The
// workgroups of an image are given block ids with equal (id % 8): they land on one XCD and share its L2, where a plain store is
// visible to an L1-bypassing load once it has been acknowledged.  Measured (tools/ubench/xcd_cluster.hip): publish + arrive + wait
// for the cluster + gather 32 KB = 1.4 us, against 1.9 us for the same hand-off across a launch boundary -- and the boundary also
// costs the next launch its argument fetch, its first weight round trip and its address set-up, which here run under the
// previous phase (the next phase's weight fragments are requested behind the K loop).
//
// A phase = conv_small_body<..., TRUNK = true> on a ConvParams record in device memory.  Nothing here orders workgroups of
// DIFFERENT images: they drift apart freely (so activations of a segment are never recycled inside it: runtime.hip defers the
// arena releases to the segment's end).
//
// Second generation: the same seam for images of SEVERAL tiles.  At 13..16 images an image is 16 workgroups at every level --
// 16 pixel tiles of conv_stream's 256 x 128 instance at 256x16, 4 x 4 tiles of conv_small's 64 x 64 instance at 64x4 -- so
// 16 images x 16 = 256 workgroups fill the chip once, two images per XCD, and halos, channel slices and GroupNorm statistics
// partials (folded by the CONSUMER phase, as in the stand-alone kernels) still never leave the image.  Four kernels (template V),
// one per set of phase bodies, so that each set has its own register allocation; the bodies are the stand-alone kernels' code
// (conv_small_body.h / conv_stream_body.h / attention_body.h, TRUNK = true) plus gn_apply_phase below.  DESIGN.md section 3.7.
#include "conv_small_body.h"
#include "conv_stream_body.h"
#include "attention_body.h"

namespace rldm {

// A phase record is 64 dwords (kernels.h, TrunkWord): every wave loads it with ONE instruction (lane l = word l) a phase ahead,
// and v_readlane moves the fields into SGPRs -- where the kernel-argument copy of a stand-alone launch lives.  (Reading the record
// field by field would be ~60 dependent VECTOR loads per phase: the launch also writes device memory, so the compiler may not use
// scalar loads for it.)
__device__ __forceinline__ unsigned rl(unsigned rec, int word) { return (unsigned)__builtin_amdgcn_readlane((int)rec, word); }
// The pointer is built as a GLOBAL-address-space pointer and only then converted to the generic type the bodies take: the compiler
// then proves every access through it global and emits global_load / global_store.  Built from an integer alone it is a FLAT pointer:
// every load and store of every phase became a flat_* instruction, which counts on lgkmcnt as well as vmcnt -- each LDS wait of a K
// loop then also waited for the whole weight ring in flight, i.e. the ring was no ring (round 3: the "open question" of DESIGN.md 3.7,
// a 128x8 phase's K loop 16.3 k cycles against 12.8 k in the stand-alone launch; tools/l1_probe.sh).
template <class T> __device__ __forceinline__ T* rl_ptr(unsigned rec, int word) {
#ifdef RLDM_TRUNK_FLAT          // (A/B builds: the round-2 behaviour)
    return reinterpret_cast<T*>(((unsigned long long)rl(rec, word + 1) << 32) | rl(rec, word));
#else
    typedef __attribute__((address_space(1))) T* global_ptr_t;
    return (T*)(global_ptr_t)(((unsigned long long)rl(rec, word + 1) << 32) | rl(rec, word));
#endif
}
__device__ __forceinline__ void unpack_phase(ConvParams& q, unsigned rec) {
    q.x0 = rl_ptr<const bf16_t>(rec, TW_X0);
    q.r0 = rl_ptr<const bf16_t>(rec, TW_R0);
    q.r1 = rl_ptr<const bf16_t>(rec, TW_R1);
    q.wpk = rl_ptr<const bf16_t>(rec, TW_WPK);
    q.bias = rl_ptr<const float>(rec, TW_BIAS);
    q.y = rl_ptr<bf16_t>(rec, TW_Y);
    q.y_stats = rl_ptr<float2>(rec, TW_YSTATS);
    q.res = rl_ptr<const bf16_t>(rec, TW_RES);
    q.R0 = (int)rl(rec, TW_R0C); q.R1 = (int)rl(rec, TW_R1C);
    q.Win = (int)rl(rec, TW_WIN); q.Hin = (int)rl(rec, TW_HIN); q.Wout = (int)rl(rec, TW_WOUT); q.Hout = (int)rl(rec, TW_HOUT);
    q.TW = (int)rl(rec, TW_TW); q.TH = (int)rl(rec, TW_TH); q.colb = (int)rl(rec, TW_COLB); q.th_shift = (int)rl(rec, TW_THSHIFT);
    q.N = (int)rl(rec, TW_N); q.y_ld = (int)rl(rec, TW_YLD); q.nviews = (int)rl(rec, TW_NVIEWS);
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int at = TW_NV0 + v * TW_NVSTRIDE;
        q.nv[v].y = rl_ptr<bf16_t>(rec, at);
        q.nv[v].gamma = rl_ptr<const float>(rec, at + 2);
        q.nv[v].beta = rl_ptr<const float>(rec, at + 4);
        q.nv[v].ld = (int)rl(rec, at + 6);
        q.nv[v].cpg_shift = (int)rl(rec, at + 7);
        q.nv[v].inv_n = __uint_as_float(rl(rec, at + 8));
        q.nv[v].eps = __uint_as_float(rl(rec, at + 9));
        q.nv[v].silu = (int)rl(rec, at + 10);
    }
    q.nv[2] = q.nv[1];                          // (trunk phases write at most two copies)
    // what a trunk phase never has / what its tile implies
    q.x1 = nullptr; q.C0 = 0; q.C1 = 0; q.st0 = nullptr; q.st1 = nullptr; q.P0 = 0; q.P1 = 0; q.temb = nullptr; q.step_ptr = nullptr;
    q.ts = nullptr; q.up = 1; q.stride = 1; q.tiles_h = 1; q.tiles_img = 1; q.dbg = 0; q.silu = 0; q.B = 0;
}
// a phase of a multi-tile cluster (kind >= 8): no views; the words they would occupy carry the consumer-side GroupNorm of the input
__device__ __forceinline__ void unpack_cluster_phase(ConvParams& q, unsigned rec) {
    unpack_phase(q, rec);
    q.nviews = 0;
    q.st0 = rl_ptr<const float2>(rec, TW_ST0);
    q.gn_gamma = rl_ptr<const float>(rec, TW_GAMMA);
    q.gn_beta = rl_ptr<const float>(rec, TW_BETA);
    q.P0 = (int)rl(rec, TW_P0);
    q.gn_groups = (int)rl(rec, TW_GROUPS);
    q.magic_cpg = (int)rl(rec, TW_MAGIC_CPG);
    q.gn_inv_n = __uint_as_float(rl(rec, TW_INVN));
    q.gn_eps = __uint_as_float(rl(rec, TW_EPS));
    q.silu = (int)rl(rec, TW_SILU);
    q.tiles_h = (int)rl(rec, TW_TILES_H);
    q.tiles_img = (int)rl(rec, TW_TILES_IMG);
    q.up = max((int)rl(rec, TW_UP), 1);         // (nearest x2 folded into the staging of an up-sampler's conv)
    // (round 4) a concatenated input normalised by the phase itself: second tensor, its statistics, the split (C1 == 0: one tensor)
    q.x1 = rl_ptr<const bf16_t>(rec, TW_X1);
    q.st1 = rl_ptr<const float2>(rec, TW_ST1);
    q.C0 = (int)rl(rec, TW_C0); q.C1 = (int)rl(rec, TW_C1); q.P1 = (int)rl(rec, TW_P1);
}

// a conv_stream phase (kind TK_STREAM): the cluster words + the second input tensor of a concatenation, nearest-x2, halo divisor
__device__ __forceinline__ void unpack_stream_phase(ConvParams& q, unsigned rec) {
    unpack_cluster_phase(q, rec);
    q.x1 = rl_ptr<const bf16_t>(rec, TW_X1);
    q.st1 = rl_ptr<const float2>(rec, TW_ST1);
    q.C0 = (int)rl(rec, TW_C0); q.C1 = (int)rl(rec, TW_C1); q.P1 = (int)rl(rec, TW_P1);
    q.magic_thv = (int)rl(rec, TW_MAGIC_THV);
}

// GroupNorm (+ SiLU) of cat[x0, x1] as a phase (a concatenated conv input is normalised once, not by every channel tile of the conv:
// norm.hip's gn_apply_kernel, same arithmetic per element): rank r of the image's cluster takes pixels [r, r + 1) * npix / ranks, all
// channels (<= 512: one thread per channel folds its group).
__device__ __forceinline__ void gn_apply_phase(const ConvParams& cp, const int rank, const int ranks, const int b, const TrunkSeam& seam) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));              // (opaque per phase: conv_small_body.h)
    const int tid = tid_;
    constexpr int NT = 512;
    const int C0 = cp.R0, C1 = cp.R1, Cin = C0 + C1, npix = cp.Win;
    const bf16_t* const gx0 = cp.x0;
    const bf16_t* const gx1 = cp.r0;
    const float2* const gs0 = cp.st0;
    const float2* const gs1 = reinterpret_cast<const float2*>(cp.res);
    const int nP0 = cp.P0, nP1 = cp.tiles_h;
    double* sD = reinterpret_cast<double*>(smem);               // [2][512]
    float* sGa = reinterpret_cast<float*>(sD + 2 * 512);        // [512]
    float* sGs = sGa + 512;
    float gam = 0.f, bet = 0.f;
    if (tid < Cin) { gam = cp.gn_gamma[tid]; bet = cp.gn_beta[tid]; }
    trunk_wait(seam, tid);
    double S = 0.0, SS = 0.0;
    if (tid < Cin) {
        const bool first = tid < C0;
        const int c = first ? tid : tid - C0;
        const int C = first ? C0 : C1;
        const int P = first ? nP0 : nP1;
        const float2* src = (first ? gs0 : gs1) + (size_t)b * P * C + c;
        int q = 0;
        for (; q + 4 <= P; q += 4) {
            float2 u[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j] = ld_act8<true>(src + (size_t)(q + j) * C);
#pragma unroll
            for (int j = 0; j < 4; ++j) { S += (double)u[j].x; SS += (double)u[j].y; }
        }
        for (; q < P; ++q) {
            const float2 u = ld_act8<true>(src + (size_t)q * C);
            S += (double)u.x;
            SS += (double)u.y;
        }
    }
    const int n8 = Cin >> 3, ppr = npix / ranks, px0 = rank * ppr, total = ppr * n8;
    constexpr int NB = 4;
    uint4 v[NB];
    int cl[NB];
    size_t dst[NB];
    auto load_batch = [&](int q0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int q = q0 + j * NT;
            const int pl = q / n8;
            cl[j] = (q - pl * n8) * 8;
            const size_t pix = (size_t)b * npix + px0 + pl;
            dst[j] = pix * Cin + cl[j];
            v[j] = make_uint4(0u, 0u, 0u, 0u);
            if (q < total) {
                const bool first = cl[j] < C0;
                v[j] = ld_act16<true>(first ? gx0 + pix * C0 + cl[j] : gx1 + pix * C1 + (cl[j] - C0));
            }
        }
    };
    load_batch(tid);
    if (tid < Cin) {
        sD[tid] = S;
        sD[512 + tid] = SS;
    }
    __syncthreads();
    if (tid < Cin) {                            // every channel's thread folds its own group (no serial phase)
        const int cpg = Cin / cp.gn_groups;
        const int gb = (tid / cpg) * cpg;
        double GS = 0.0, GSS = 0.0;
        for (int i = 0; i < cpg; ++i) {
            GS += sD[gb + i];
            GSS += sD[512 + gb + i];
        }
        const double inv_n = (double)cp.gn_inv_n;
        const double mean = GS * inv_n;
        double var = GSS * inv_n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const float a = gam * __builtin_amdgcn_rsqf((float)var + cp.gn_eps);
        sGa[tid] = a;
        sGs[tid] = bet - (float)mean * a;
    }
    __syncthreads();
    for (int q0 = tid; q0 < total; q0 += NT * NB) {
        if (q0 != tid) load_batch(q0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (q0 + j * NT >= total) continue;
            const float4 a0 = *reinterpret_cast<const float4*>(sGa + cl[j]), a1 = *reinterpret_cast<const float4*>(sGa + cl[j] + 4);
            const float4 s0 = *reinterpret_cast<const float4*>(sGs + cl[j]), s1 = *reinterpret_cast<const float4*>(sGs + cl[j] + 4);
            float f0 = bf16lo(v[j].x) * a0.x + s0.x, f1 = bf16hi(v[j].x) * a0.y + s0.y;
            float f2 = bf16lo(v[j].y) * a0.z + s0.z, f3 = bf16hi(v[j].y) * a0.w + s0.w;
            float f4 = bf16lo(v[j].z) * a1.x + s1.x, f5 = bf16hi(v[j].z) * a1.y + s1.y;
            float f6 = bf16lo(v[j].w) * a1.z + s1.z, f7 = bf16hi(v[j].w) * a1.w + s1.w;
            if (cp.silu) {
                silu_x8(f0, f1, f2, f3, f4, f5, f6, f7);
            }
            uint4 o;
            o.x = pack_bf16x2(f0, f1); o.y = pack_bf16x2(f2, f3);
            o.z = pack_bf16x2(f4, f5); o.w = pack_bf16x2(f6, f7);
            *reinterpret_cast<uint4*>(cp.y + dst[j]) = o;
        }
    }
    trunk_arrive(seam, tid);
}

// CL = false: image-owning tiles (kinds 0..6 + attention over a pre-normalised x); CL = true: multi-tile clusters (kinds 8..13).
// Two kernels, so that each set of instances gets its own register allocation.
// V == 4 (round 4): the full-resolution level on conv_stream's 4-wave 128 x 128 instance -- 32 workgroups of 256 threads per image,
// TWO workgroups per CU (512 in all at 16 images: __launch_bounds__' second argument is waves per SIMD).  The two workgroups of a
// CU belong to different images (block ids below 256 are images 0..7, the rest images 8..15), i.e. to clusters that are never
// ordered against each other: one's statistics round trip / fold / first halo chunk / epilogue runs under the other's K loop.
template <int V>
__global__ void __launch_bounds__(V == 4 ? 256 : 512, V == 4 ? 2 : 1) trunk_kernel(const TrunkParams tp) {
    constexpr bool CL = V == 1, ST = V >= 2;
    // block id -> (image, channel tile): ids with the same (id % 8) share an XCD; an image's `ranks` tiles are 8 apart
    const int wg = blockIdx.x;
    const int ranks = tp.ranks;
    const int t = wg >> 3;
    const int rank = t % ranks, b = (t / ranks) * 8 + (wg & 7);
    if (b >= tp.B) return;                      // (batch not a multiple of 8: the surplus workgroups have no cluster)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // a workgroup's place in its image: channel tile nt (NWN 32-channel tiles wide) of pixel tile mt; its waves' weight streams
    const int nwn = tp.nwn, nt = rank % tp.ntile_n, mt = rank / tp.ntile_n;
    const int stream_id = (nt * nwn + wave % nwn) * (8 / nwn) + wave / nwn;     // (NWN == 1: rank * 8 + k-group)
    unsigned* const counter = tp.counters + b * 32;
    const unsigned* const recs = reinterpret_cast<const unsigned*>(tp.phases);

    // the cluster's arrival counter is never reset: this launch waits for arrivals past `base` = launches so far x arrivals per
    // launch ([3] is advanced by rank 0 at the very end: visible to the next launch across the kernel boundary)
    const unsigned epoch = counter[3];
    const unsigned base = epoch * (unsigned)(tp.nphases * ranks);
    // placement check (the protocol's only assumption): rank 0 publishes its XCC id ahead of its first arrive, every other
    // rank compares once it has waited for that arrive
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    if (rank == 0 && tid == 0) __hip_atomic_store(counter + 1, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    if constexpr (V == 4) {
        // the second image group starts tp.skew x 1024 cycles late: the two workgroups of a CU then alternate -- one in its K loop while
        // the other is between two of them -- instead of asking for the matrix pipes at the same time (they share them fairly: an
        // offset, once there, stays)
        if (t / ranks != 0)
            for (int i = 0; i < tp.skew; ++i) __builtin_amdgcn_s_sleep(16);
    }
    bf16x8 wpf[kTrunkPrefetch];
    unsigned rec = recs[lane];                  // phase 0's record
    auto wave_stream = [&](unsigned r) __attribute__((always_inline)) {
        return reinterpret_cast<const unsigned char*>(rl_ptr<const bf16_t>(r, TW_WPK)) + ((size_t)stream_id * rl(r, TW_NMINE)) * 1024;
    };
    // (round 4: the multi-tile clusters of variant 1 carry kClusterPrefetch fragments across phases too -- the kernel sits at 196 VGPRs since
    //  round 2's hoisting fix, and its K loops start on weights that every XCD pulls from the Infinity Cache: 8.7 k cycles against 3 k stand-alone)
    constexpr bool PFV = V == 0 || (V == 1 && kClusterPrefetch > 0);
    if constexpr (PFV) {
        const unsigned char* w0 = wave_stream(rec);
        const int g0 = (int)rl(rec, TW_G);
#pragma unroll
        for (int j = 0; j < kTrunkPrefetch; ++j)
            if (j < g0) wpf[j] = *reinterpret_cast<const bf16x8*>(w0 + (unsigned)(j * 1024 + lane * 16));
    }
    for (int i = 0; i < tp.nphases; ++i) {
        const unsigned nrec = i + 1 < tp.nphases ? recs[(i + 1) * TW_WORDS + lane] : 0u;     // requested a phase ahead
        ConvParams cp;                          // NOT zeroed (zeroing measured -0.4 %): fields a phase record does not carry are UNDEFINED, so
                                                // a body compiled with TRUNK = true must not test one (guard such code with `if constexpr
                                                // (!TRUNK)`: the compiler otherwise resolves the branch as it likes -- it once dropped a
                                                // whole phase body)
        const int kind = (int)rl(rec, TW_KIND);
        if constexpr (ST) unpack_stream_phase(cp, rec);
        else if constexpr (CL) unpack_cluster_phase(cp, rec);
        else if (kind == TK_GN_APPLY) unpack_cluster_phase(cp, rec);
        else unpack_phase(cp, rec);
        TrunkSeam seam;
        seam.counter = counter;
        seam.wait_for = base + (unsigned)(i * ranks);
        seam.has_wait = i != 0;
        seam.error = tp.error;
        seam.ts = (tp.ts && wg == 0 && i < 16) ? tp.ts + i * 16 : nullptr;
        const int toff = (int)rl(rec, TW_TEMBOFF);
        seam.temb = toff >= 0 ? tp.temb + toff : nullptr;
        seam.step_ptr = tp.step_ptr;
        seam.temb_rows_per_step = tp.temb_rows_per_step;
        seam.temb_per_sample = tp.temb_per_sample;
        seam.temb_ld = tp.temb_ld;
        // (the next record has long arrived when the K loop ends: its weight stream is resolved inside the body, behind the K loop)
        seam.next_rec = nrec;
        seam.next_rank_kg = stream_id;
        seam.rank = rank;
        seam.ranks = ranks;
        bool conv_done = true;
        if constexpr (V == 2) {
            conv_stream_body<2, 4, true>(cp, nt, mt, b, seam);      // full-resolution level: 16 tiles of 256 pixels x 128 channels per image
        } else if constexpr (V == 3) {
            conv_stream_body<1, 2, true>(cp, nt, mt, b, seam);      // 128x8 level: 8 tiles of 128 pixels x 2 channel tiles of 64
        } else if constexpr (V == 5) {
            // (round 5) 128x8 level on round 4's tile: 16 tiles of 64 pixels (8 x 8) x 128 channels x 2 k-groups per image
            conv_stream_body<1, 4, true, 8, 2>(cp, nt, mt, b, seam);
        } else if constexpr (V == 4) {
            // full-resolution level: 32 tiles of 128 pixels x 128 channels per image, 4 waves; the level's up-sampler conv in its sub-pixel
            // form: 8 INPUT tiles x 4 parities
            const unsigned form = rl(rec, TW_SUB);                      // (the host's choice of instance: 0 | 1 sub-pixel | 2 tiles as tall as the image)
            if (form == 1u) conv_stream_body<1, 4, true, 4, 4, 1, true>(cp, rank & 3, rank >> 2, b, seam);
            else if (form == 2u) conv_stream_body<1, 4, true, 4, 4, 1, false, false, true>(cp, nt, mt, b, seam);
            else conv_stream_body<1, 4, true, 4>(cp, nt, mt, b, seam);
        } else if constexpr (!CL) {
            switch (kind) {
                case 0: conv_small_body<1, 2, 9, 2, true>(cp, rank, 0, b, wpf, seam); break;
                case 1: conv_small_body<1, 4, 9, 2, true>(cp, rank, 0, b, wpf, seam); break;
                case 2: conv_small_body<1, 2, 1, 2, true>(cp, rank, 0, b, wpf, seam); break;
                // 32-pixel images (the 32x1 level of the nuScenes network): the same three layers on one 32-pixel tile
                case 4: conv_small_body<1, 2, 9, 1, true>(cp, rank, 0, b, wpf, seam); break;
                case 5: conv_small_body<1, 4, 9, 1, true>(cp, rank, 0, b, wpf, seam); break;
                case 6: conv_small_body<1, 2, 1, 1, true>(cp, rank, 0, b, wpf, seam); break;
                // (round 5) 16-channel tiles: 16 workgroups per image (C_in 256 / 512 for the 3x3, 256 for the pointwise conv)
                case TK_H16 + 0: conv_small_body<1, 1, 9, 2, true, kTrunkPrefetch, true>(cp, rank, 0, b, wpf, seam); break;
                case TK_H16 + 1: conv_small_body<1, 2, 9, 2, true, kTrunkPrefetch, true>(cp, rank, 0, b, wpf, seam); break;
                case TK_H16 + 2: conv_small_body<1, 1, 1, 2, true, kTrunkPrefetch, true>(cp, rank, 0, b, wpf, seam); break;
                case TK_GN_APPLY: {             // (a phase without weights: the next conv's first fragments are requested here)
                    gn_apply_phase(cp, rank, ranks, b, seam);
                    const int next_g = (int)rl(nrec, TW_G);
                    if (next_g > 0) {
                        const unsigned char* nw = wave_stream(nrec);
#pragma unroll
                        for (int j = 0; j < kTrunkPrefetch; ++j)
                            if (j < next_g) wpf[j] = *reinterpret_cast<const bf16x8*>(nw + (unsigned)(j * 1024 + lane * 16));
                    }
                } break;
                default: conv_done = false; break;
            }
        } else {
            // multi-tile clusters (the 64x4 level: 4 pixel tiles x 4 channel tiles of 64 per image): GroupNorm + SiLU of the input
            // folded into the staging as in the stand-alone launch, from the statistics the previous phase published
            switch (kind) {
                // (round 4: 128 input channels -- the first conv of the level, behind the stride-2 down-sampler; the kernel has had the registers
                //  for its 18-fragment ring since round 2's hoisting fix)
                case TK_CL_3x3_128: conv_small_body<2, 2, 9, 2, true, kClusterPrefetch>(cp, nt, mt, b, wpf, seam); break;
                case TK_CL_3x3_256: conv_small_body<2, 4, 9, 2, true, kClusterPrefetch>(cp, nt, mt, b, wpf, seam); break;
                case TK_CL_3x3_384: conv_small_body<2, 6, 9, 2, true, kClusterPrefetch>(cp, nt, mt, b, wpf, seam); break;
                case TK_CL_3x3_512: conv_small_body<2, 8, 9, 2, true, kClusterPrefetch>(cp, nt, mt, b, wpf, seam); break;
                case TK_CL_1x1_256: conv_small_body<2, 4, 1, 2, true, kClusterPrefetch>(cp, nt, mt, b, wpf, seam); break;
                case TK_GN_APPLY: {
                    gn_apply_phase(cp, rank, ranks, b, seam);
                    if constexpr (kClusterPrefetch > 0) {       // (a phase without weights: the next conv's first fragments are requested here)
                        const int next_g = (int)rl(nrec, TW_G);
                        if (next_g > 0) {
                            const unsigned char* nw = wave_stream(nrec);
#pragma unroll
                            for (int j = 0; j < kClusterPrefetch; ++j)
                                if (j < next_g) wpf[j] = *reinterpret_cast<const bf16x8*>(nw + (unsigned)(j * 1024 + lane * 16));
                        }
                    }
                } break;
                default: conv_done = false; break;
            }
        }
        if (!ST && !conv_done) {
                // attention core of the block (GroupNorm already applied by the producer of x): this workgroup's heads / ranks
                // heads of image b, one query tile per wave; its output projection is the next phase (a 1x1 conv)
                AttnQkvParams ap;
                ap.x = cp.x0; ap.st = nullptr; ap.P = 0; ap.gamma = nullptr; ap.beta = nullptr; ap.eps = 0.f; ap.groups = 1;
                ap.inv_n = 0.f; ap.magic_cpg = 0;
                ap.wfrag = cp.wpk; ap.bias = cp.bias; ap.out = cp.y;
                ap.B = tp.B; ap.L = cp.Win; ap.C = cp.N; ap.ts = nullptr; ap.ts_L = 0;
                const int Lp = (ap.L + 31) & ~31;
                if (CL && kind == TK_ATTN_FOLD) {   // raw x + the producer's statistics (cluster words), two query tiles per wave
                    ap.st = cp.st0; ap.P = cp.P0; ap.gamma = cp.gn_gamma; ap.beta = cp.gn_beta; ap.eps = cp.gn_eps;
                    ap.groups = cp.gn_groups; ap.inv_n = cp.gn_inv_n; ap.magic_cpg = cp.magic_cpg;
                    if constexpr (CL) attention_qkv2_body<1, true>(ap, 8, Lp, (ap.C >> 3) / ranks, b, rank, seam);
                } else if constexpr (V == 0) {
                    attention_qkv2_body<0, true>(ap, 8, Lp, (ap.C >> 3) / ranks, b, rank, seam);
                }
                // the next phase's first weight fragments (what a conv phase requests behind its K loop)
                const int next_g = PFV ? (int)rl(nrec, TW_G) : 0;
                if (next_g > 0) {
                    const unsigned char* nw = wave_stream(nrec);
#pragma unroll
                    for (int j = 0; j < kTrunkPrefetch; ++j)
                        if (j < next_g) wpf[j] = *reinterpret_cast<const bf16x8*>(nw + (unsigned)(j * 1024 + lane * 16));
                }
        }
        if (i == 1 && rank != 0 && tid == 0) {  // (rank 0's first arrive has been waited for: its XCC id is there)
            const unsigned x0 = __hip_atomic_load(counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (x0 != xcc + 1u) *tp.error = 2;  // not on rank 0's XCD: this workgroup's reads may be stale
        }
        rec = nrec;
    }
    if (rank == 0 && tid == 0) counter[3] = epoch + 1u;
}

// workgroups of trunk_kernel<variant> the runtime will keep resident on ONE CU with `lds` bytes of dynamic LDS each (< 0: the query
// failed).  The plan builder asks before it commits a segment to a persistent launch: a spin-waiting grid that is not co-resident
// only fails through the bounded-wait self-check, which is a slow way to find out (variant 4 needs TWO per CU: 2 x 80 KiB is the
// whole LDS, so any static LDS, scratch or a partitioned CU would halve it).
int trunk_max_resident(int variant, size_t lds) {
    if (variant < 0 || variant > 5) return -1;
    static DynLdsLimit lds_limit[6];
    const int cl = variant;
    auto kern = cl == 0 ? trunk_kernel<0> : (cl == 1 ? trunk_kernel<1> : (cl == 2 ? trunk_kernel<2> : (cl == 3 ? trunk_kernel<3> : (cl == 4 ? trunk_kernel<4> : trunk_kernel<5>))));
    if (lds_limit[cl].ensure(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return -1;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kern), cl == 4 ? 256 : 512, lds) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return n;
}

int launch_trunk(const TrunkParams& tp, size_t lds, hipStream_t stream) {
    RLDM_REQUIRE(tp.variant >= 0 && tp.variant <= 5, "trunk: bad kernel variant");
    const int per_cu = tp.variant == 4 ? 2 : 1;  // (variant 4: 256-thread workgroups, two per CU)
    RLDM_REQUIRE(tp.nphases >= 1 && tp.ranks >= 1 && tp.ranks <= 16 * per_cu && tp.B >= 1 && (tp.nwn == 1 || tp.nwn == 2 || tp.nwn == 4) &&
                     tp.ntile_n >= 1 && tp.ranks % tp.ntile_n == 0, "trunk: bad parameters");
    RLDM_REQUIRE(lds <= (size_t)160 * 1024 / per_cu, "trunk: LDS tile too large");
    const int groups = (tp.B + 7) / 8;
    const int grid = 8 * tp.ranks * groups;
    RLDM_REQUIRE(grid <= 256 * per_cu, "trunk: the grid must be co-resident (one workgroup per CU; two of the 4-wave variant)");
    static DynLdsLimit lds_limit[6];             // per instantiation and device, thread safe
    const int cl = tp.variant;
    auto kern = cl == 0 ? trunk_kernel<0> : (cl == 1 ? trunk_kernel<1> : (cl == 2 ? trunk_kernel<2> : (cl == 3 ? trunk_kernel<3> : (cl == 4 ? trunk_kernel<4> : trunk_kernel<5>))));
    RLDM_HIP_CHECK(lds_limit[cl].ensure(reinterpret_cast<const void*>(kern), lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(cl == 4 ? 256 : 512), lds, stream, tp);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace rldm
