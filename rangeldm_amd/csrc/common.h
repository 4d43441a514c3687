// Shared device helpers and host-side error plumbing for librangeldm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <string>

typedef unsigned short bf16_t;  // raw bfloat16 bits in HBM / LDS
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define RLDM_WAVE 64

namespace rldm {

void set_error(const std::string& msg);

#define RLDM_HIP_CHECK(expr)                                                                        \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            rldm::set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ +    \
                            ":" + std::to_string(__LINE__) + ")");                                  \
            return 1;                                                                               \
        }                                                                                           \
    } while (0)

#define RLDM_REQUIRE(cond, msg)                                                                     \
    do {                                                                                            \
        if (!(cond)) {                                                                              \
            rldm::set_error(std::string(msg) + " [" #cond "] (" + __FILE__ + ":" +                   \
                            std::to_string(__LINE__) + ")");                                        \
            return 1;                                                                               \
        }                                                                                           \
    } while (0)

// Largest dynamic-LDS size a kernel has been enabled for, PER DEVICE (hipFuncSetAttribute is a per-device setting) and
// safe to call from several host threads.  One object per kernel instantiation (function-local static at the launch site).
struct DynLdsLimit {
    static constexpr int kMaxDevices = 16;
    std::atomic<size_t> set[kMaxDevices];
    DynLdsLimit() { for (auto& v : set) v.store(0); }
    // returns hipSuccess without a runtime call when `bytes` is already covered on the current device
    hipError_t ensure(const void* kernel, size_t bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const bool tracked = dev >= 0 && dev < kMaxDevices;
        if (tracked && bytes <= set[dev].load(std::memory_order_acquire)) return hipSuccess;
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        if (tracked) {                                   // monotone maximum; racing threads at worst repeat the call
            size_t cur = set[dev].load(std::memory_order_relaxed);
            while (cur < bytes && !set[dev].compare_exchange_weak(cur, bytes, std::memory_order_release)) {}
        }
        return hipSuccess;
    }
};

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) -------------------------------------------------------
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
    union { float f; uint32_t u; } v;
    v.f = f;
    uint32_t u = v.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__host__ __device__ inline float bf16_to_f32(bf16_t h) {
    union { float f; uint32_t u; } v;
    v.u = ((uint32_t)h) << 16;
    return v.f;
}
// device: hardware v_cvt_pk_bf16_f32 (round-to-nearest-even) via the native __bf16 vector conversion
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ inline float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ inline float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// x * sigmoid(x) on the hardware exp2 / rcp units (1 ulp class; the result is rounded to bf16 right after)
__device__ inline float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// SiLU of eight values (the unit every staging path normalises: 16 bytes of bf16).  Measured alternatives that trade transcendentals
// for plain VALU work (one reciprocal per pair / per quad of values, Newton reciprocal) are all SLOWER end to end (-0.9 / -1.4 / -2.8 %):
// v_exp_f32 / v_rcp_f32 are not what the staging is short of, issue slots are (the compiler already packs the affine into v_pk_fma_f32).
__device__ __forceinline__ void silu_x8(float& f0, float& f1, float& f2, float& f3, float& f4, float& f5, float& f6, float& f7) {
    f0 = silu_f(f0); f1 = silu_f(f1); f2 = silu_f(f2); f3 = silu_f(f3);
    f4 = silu_f(f4); f5 = silu_f(f5); f6 = silu_f(f6); f7 = silu_f(f7);
}

// The sampler's step index (device int behind a uniform pointer) as a VECTOR load.  As a scalar load it would share lgkmcnt with
// the kernel-argument loads, which return out of order: the first use of any later argument then waits for it too, and every
// wave of the launch stalls one cold memory round trip at entry.  The lane offset is an opaque zero, so the compiler keeps the
// load in the vector queue, where it is the first (in-order) request and its consumer far away waits for nothing.
__device__ __forceinline__ int load_step_vector(const int* step_ptr) {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return step_ptr[z];
}

// XCD-aware block remap (guide T1): blocks land on XCD (bid % 8); give each XCD a contiguous range of logical ids
// so neighbouring tiles (which share input halos / weight panels) hit the same private L2.  Bijective for any n.
__device__ inline int xcd_remap(int bid, int nblocks) {
    const int nx = 8;
    int q = nblocks / nx, r = nblocks % nx;
    int xcd = bid % nx, k = bid / nx;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

}  // namespace rldm
