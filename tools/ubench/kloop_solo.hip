// micro-benchmark (tuning aid, not product code): conv_stream's K-loop instruction mix -- per k-step MI MFMAs 32x32x16 on MI independent
// accumulators, MI ds_read_b128 (pixel fragments, requested PFD steps ahead) and one 1-KiB weight fragment from global memory (ring of G) --
// with ONE or TWO waves per SIMD, to find what bounds a wave that is alone on its SIMD (17 k cycles for 9.2 k of MFMAs).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/kloop_solo.hip -o /tmp/kloop_solo && /tmp/kloop_solo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// MODE bits: 1 = LDS fragment reads, 2 = global weight loads, 4 = reads interleaved behind each MFMA (else behind the step's last MFMA),
//   8 = the chunk's staging (6 halo pieces: global load -> a*x+s -> SiLU -> bf16 -> ds_write_b128) as ONE burst behind step 12 of every 36,
//   16 = the same staging spread: one sixth of it behind every sixth k-step, 64 = ... one piece's arithmetic spread over its six steps
//   (element pairs), 32 = s_barrier every 36 steps
__device__ __forceinline__ float silu1(float x) { return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * x)); }
__device__ __forceinline__ unsigned pk(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 r; r[0] = (__bf16)a; r[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ uint4 stage_piece(uint4 v, float a, float s) {
    float f0 = silu1(lo(v.x) * a + s), f1 = silu1(hi(v.x) * a + s), f2 = silu1(lo(v.y) * a + s), f3 = silu1(hi(v.y) * a + s);
    float f4 = silu1(lo(v.z) * a + s), f5 = silu1(hi(v.z) * a + s), f6 = silu1(lo(v.w) * a + s), f7 = silu1(hi(v.w) * a + s);
    return make_uint4(pk(f0, f1), pk(f2, f3), pk(f4, f5), pk(f6, f7));
}
template <int NW, int MI, int PFD, int MODE, int PAT = 0>
__global__ void __launch_bounds__(64 * NW, NW / 4) k(const unsigned char* __restrict__ w, float* out, int steps, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int G = 12;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)(i & 255) * 1e-3f;
    __syncthreads();
    const int l31 = lane & 31, kh = lane >> 5;
    int xo[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xo[mi] = ((mi * 32 + l31) * 144 + kh * 16) % 60000;
    const unsigned char* wp = w + (size_t)(blockIdx.x % 4 * 8 + wave % 4) * (steps + 32) * 1024 + lane * 16;
    f32x16 acc[MI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 xr[PFD][MI], wr[G];
#pragma unroll
    for (int j = 0; j < G; ++j) wr[j] = *reinterpret_cast<const bf16x8*>(wp + j * 1024);
#pragma unroll
    for (int d = 0; d < PFD; ++d)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) xr[d][mi] = *reinterpret_cast<const bf16x8*>(smem + xo[mi] + d * 32);
    wp += G * 1024;
    uint4 hp[6];
    const unsigned char* hsrc = w + (size_t)tid * 16;
    unsigned char* hdst = smem + 32768 + tid * 16;
#pragma unroll
    for (int i = 0; i < 6; ++i) hp[i] = *reinterpret_cast<const uint4*>(hsrc + i * 8192);
    const float ga = 1.0009765625f, gs = 0.01f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c0 = 0; c0 < steps; c0 += 3 * G) {
#pragma unroll
      for (int row = 0; row < 3; ++row) {       // (unrolled: the halo piece a step works on is a compile-time register)
        const int s0 = c0 + row * G;
#pragma unroll
        for (int j = 0; j < G; ++j) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[j], xr[j % PFD][mi], acc[mi], 0, 0, 0);
                if (MODE & 1) xr[j % PFD][mi] = *reinterpret_cast<const bf16x8*>(smem + xo[mi] + ((j + PFD) % 9) * 32 + ((j + PFD) / 9) * 1440);
            }
            if (MODE & 2) wr[j] = *reinterpret_cast<const bf16x8*>(wp + j * 1024);
            if (MODE & 64) {
                // one element PAIR of the current halo piece per k-step (steps 0..3 of the piece's six), its store + the next request at step 4:
                // the ~13 VALU instructions share the step's scheduling region with the MFMAs (placed between them below)
                const int i = (row * 2 + j / 6) % 6, part = j % 6;
                uint4 v = hp[0];
#pragma unroll
                for (int q = 1; q < 6; ++q) if (q == i) v = hp[q];
                if (part < 4) {
                    const unsigned u = part == 0 ? v.x : (part == 1 ? v.y : (part == 2 ? v.z : v.w));
                    const unsigned r = pk(silu1(lo(u) * ga + gs), silu1(hi(u) * ga + gs));
                    if (part == 0) v.x = r; else if (part == 1) v.y = r; else if (part == 2) v.z = r; else v.w = r;
                } else if (part == 4) {
                    *reinterpret_cast<uint4*>(hdst + i * 4096) = v;
                    v = *reinterpret_cast<const uint4*>(hsrc + ((s0 + j) & 63) * 4096);
                }
#pragma unroll
                for (int q = 0; q < 6; ++q) if (q == i) hp[q] = v;
            }
            if (MODE & 64) {
                if (PAT == 0) {                 // MFMA, read, 4 VALU -- four times
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (MODE & 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    }
                } else if (PAT == 1) {          // MFMA, 3 VALU, read
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                        if (MODE & 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                } else if (PAT == 2) {          // the compiler's own order inside the step
                } else if (PAT == 3) {          // MFMA, read, 2 VALU, 1 transcendental
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (MODE & 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
                    }
                } else if (PAT == 4) {          // MFMA, read, 3 VALU
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (MODE & 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    }
                }
            } else if (MODE & 4) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (MODE & 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, MI, 0);
                if (MODE & 1) __builtin_amdgcn_sched_group_barrier(0x100, MI, 0);
            }
            if (MODE & 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if ((MODE & 16) && (j % 6) == 5) {          // one piece behind every sixth step (6 pieces per 36 steps)
                const int i = (row * 2 + j / 6) % 6;
                uint4 v = hp[0];
#pragma unroll
                for (int q = 1; q < 6; ++q) if (q == i) v = hp[q];
                v = stage_piece(v, ga, gs);
                *reinterpret_cast<uint4*>(hdst + i * 4096) = v;
                const uint4 nv = *reinterpret_cast<const uint4*>(hsrc + ((s0 + j) & 63) * 4096);
#pragma unroll
                for (int q = 0; q < 6; ++q) if (q == i) hp[q] = nv;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if ((MODE & 8) && row == 0) {                   // the burst: all six pieces behind the first row of taps
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const uint4 v = stage_piece(hp[i], ga, gs);
                *reinterpret_cast<uint4*>(hdst + i * 4096) = v;
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) hp[i] = *reinterpret_cast<const uint4*>(hsrc + ((s0 + i) & 63) * 4096);
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((MODE & 32) && row == 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        wp += G * 1024;
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < MI; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NW, int MI, int PFD, int MODE, int PAT = 0>
static void run(const unsigned char* w, float* out, unsigned long long* cyc, int blocks, const char* what) {
    const int steps = 72 * 8;
    auto kern = k<NW, MI, PFD, MODE, PAT>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * NW), 64 * 1024, 0, w, out, 24, cyc);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * NW), 64 * 1024, 0, w, out, steps, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per_step = (double)c / steps;
    const int waves_per_simd = NW * (blocks / 256) / 4;
    printf("%-34s waves/SIMD %d  MI %d  read-ahead %d  mode %d: %7.1f cycles per k-step and wave = %5.1f per MFMA on the SIMD (%.0f %% of the pipe), %.1f us\n",
           what, waves_per_simd, MI, PFD, MODE, per_step, per_step / (MI * waves_per_simd), 100.0 * 32.0 * MI * waves_per_simd / per_step, ms * 1e3);
}

int main() {
    unsigned char* w; float* out; unsigned long long* cyc;
    const size_t wbytes = (size_t)32 * (72 * 8 + 64) * 1024;
    hipMalloc(&w, wbytes); hipMemset(w, 0x11, wbytes);
    hipMalloc(&out, 512 * 512 * 4); hipMalloc(&cyc, 8);
    // one 4-wave workgroup per CU (a wave alone on its SIMD)
    run<4, 4, 2, 0>(w, out, cyc, 256, "MFMA only");
    run<4, 4, 2, 1>(w, out, cyc, 256, "+ LDS reads (behind the step)");
    run<4, 4, 2, 5>(w, out, cyc, 256, "+ LDS reads (interleaved)");
    run<4, 4, 2, 2>(w, out, cyc, 256, "+ weight loads");
    run<4, 4, 2, 3>(w, out, cyc, 256, "+ both (behind)");
    run<4, 4, 2, 7>(w, out, cyc, 256, "+ both (interleaved)");
    run<4, 4, 3, 7>(w, out, cyc, 256, "+ both (interleaved), 3 sets");
    run<4, 4, 2, 7 + 32>(w, out, cyc, 256, "+ barrier per chunk");
    run<4, 4, 2, 7 + 32 + 8>(w, out, cyc, 256, "+ barrier + staging BURST");
    run<4, 4, 2, 7 + 32 + 16>(w, out, cyc, 256, "+ barrier + staging per 6th step");
    run<4, 4, 2, 7 + 32 + 64>(w, out, cyc, 256, "+ barrier + staging IN the MFMA gaps");
    run<4, 4, 2, 7 + 32 + 64, 1>(w, out, cyc, 256, "  ... MFMA, 3 VALU, read");
    run<4, 4, 2, 7 + 32 + 64, 2>(w, out, cyc, 256, "  ... compiler's order");
    run<4, 4, 2, 7 + 32 + 64, 3>(w, out, cyc, 256, "  ... MFMA, read, 2 VALU, 1 trans");
    run<4, 4, 2, 7 + 32 + 64, 4>(w, out, cyc, 256, "  ... MFMA, read, 3 VALU");
    run<4, 8, 1, 7>(w, out, cyc, 256, "MI 8, one set, interleaved");
    run<4, 8, 2, 7>(w, out, cyc, 256, "MI 8, two sets, interleaved");
    // two 4-wave workgroups per CU / one 8-wave workgroup (two waves per SIMD)
    run<4, 4, 2, 0>(w, out, cyc, 512, "2 WGs: MFMA only");
    run<4, 4, 2, 7>(w, out, cyc, 512, "2 WGs: + both (interleaved)");
    run<4, 4, 2, 3>(w, out, cyc, 512, "2 WGs: + both (behind)");
    run<8, 4, 2, 7>(w, out, cyc, 256, "8 waves: + both (interleaved)");
    run<8, 4, 2, 3>(w, out, cyc, 256, "8 waves: + both (behind)");
    run<8, 4, 2, 7 + 32 + 8>(w, out, cyc, 256, "8 waves: + barrier + BURST");
    run<8, 4, 2, 7 + 32 + 16>(w, out, cyc, 256, "8 waves: + barrier + per 6th step");
    run<8, 4, 2, 7 + 32 + 64>(w, out, cyc, 256, "8 waves: + barrier + in the MFMA gaps");
    run<8, 4, 2, 7 + 32 + 64, 1>(w, out, cyc, 256, "8 waves:  ... MFMA, 3 VALU, read");
    run<8, 4, 2, 7 + 32 + 64, 2>(w, out, cyc, 256, "8 waves:  ... compiler's order");
    run<8, 4, 2, 7 + 32 + 64, 3>(w, out, cyc, 256, "8 waves:  ... MFMA, read, 2 VALU, 1 trans");
    run<8, 4, 2, 7 + 32 + 64, 4>(w, out, cyc, 256, "8 waves:  ... MFMA, read, 3 VALU");
    return 0;
}
