// calibrate.hip -- what THIS box does, measured in the same process as the benchmark (bench.py `calibration`).
//
// The boxes of a pool differ by several per cent and the shader clock sags under sustained matrix load (2.4 GHz nominal, 1.8 - 2.2
// measured under conv_stream); a throughput line that carries neither cannot tell a slow box from a slow build.  Three figures:
//   * a pure-MFMA loop on every SIMD of the chip (v_mfma_f32_32x32x16_bf16 on register operands, eight independent accumulators per
//     wave, one wave per SIMD): TFLOP/s by HIP events, and the shader clock while it runs (s_memtime = shader-clock ticks,
//     s_memrealtime = the 100 MHz constant counter, read by the same wave at both ends);
//   * a device-to-device copy of 1 GiB (16 bytes per lane, grid-stride): GB/s read + written;
//   * clock stamps a caller puts around ANY work on a stream (rldm_calib_clock_stamp): one workgroup per XCD records (XCC id,
//     s_memtime, s_memrealtime); the difference of two stamps of the same XCD is the mean shader clock over that work -- the
//     headline batch replayed from its graphs, for `roofline.frac_clock_adjusted`.
#include "common.h"
#include "../../include/rangeldm_hip.h"

namespace {

__global__ void __launch_bounds__(256) calib_mfma_kernel(int iters, float* sink, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)(1.f + 0.0078125f * (float)((lane + e) & 7));
        b[e] = (__bf16)(0.5f - 0.0078125f * (float)((lane * 3 + e) & 7));
    }
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;                       // (keeps the accumulators alive)
    if (threadIdx.x == 0 && blockIdx.x < 256) {
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

__global__ void __launch_bounds__(256) calib_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

__global__ void __launch_bounds__(64) calib_stamp_kernel(unsigned long long* slots) {
    if (threadIdx.x != 0) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    const unsigned long long c = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
    unsigned long long* s = slots + 4 * blockIdx.x;
    s[0] = xcc; s[1] = c; s[2] = r; s[3] = 1ull;
}

}  // namespace

extern "C" {

int rldm_calibrate(double* mfma_tflops, double* mfma_clock_mhz, double* copy_gbs, void* stream) {
    RLDM_REQUIRE(mfma_tflops && mfma_clock_mhz && copy_gbs, "null argument");
    hipStream_t st = (hipStream_t)stream;
    int dev = 0, cus = 256;
    RLDM_HIP_CHECK(hipGetDevice(&dev));
    RLDM_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    float* sink = nullptr;
    unsigned long long* clk = nullptr;
    RLDM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&sink), 256));
    RLDM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&clk), 512 * sizeof(unsigned long long)));
    RLDM_HIP_CHECK(hipMemsetAsync(clk, 0, 512 * sizeof(unsigned long long), st));
    hipEvent_t e0, e1;
    RLDM_HIP_CHECK(hipEventCreate(&e0));
    RLDM_HIP_CHECK(hipEventCreate(&e1));
    // ~20 ms of matrix work (160 k x 8 MFMAs of 32 cycles per wave): long enough for the clock to settle where sustained MFMA load
    // puts it (the first, shorter, launch warms up)
    const int iters = 160000, grid = cus;                   // one workgroup of 4 waves per CU = one wave per SIMD
    calib_mfma_kernel<<<grid, 256, 0, st>>>(iters / 8, sink, clk);
    RLDM_HIP_CHECK(hipEventRecord(e0, st));
    calib_mfma_kernel<<<grid, 256, 0, st>>>(iters, sink, clk);
    RLDM_HIP_CHECK(hipEventRecord(e1, st));
    RLDM_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    RLDM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *mfma_tflops = (double)grid * 4 * iters * 8 * (2.0 * 32 * 32 * 16) / (ms * 1e-3) / 1e12;
    unsigned long long h[512];
    RLDM_HIP_CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    double ticks = 0, real = 0;
    for (int i = 0; i < 256 && i < grid; ++i) { ticks += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
    *mfma_clock_mhz = real > 0 ? ticks / real * 100.0 : 0.0;       // (s_memrealtime: 100 MHz)
    // 1 GiB copy
    const size_t bytes = (size_t)1 << 30;
    uint4 *src = nullptr, *dst = nullptr;
    RLDM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&src), bytes));
    RLDM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dst), bytes));
    RLDM_HIP_CHECK(hipMemsetAsync(src, 1, bytes, st));
    calib_copy_kernel<<<cus * 8, 256, 0, st>>>(src, dst, bytes / 16);
    RLDM_HIP_CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < 3; ++r) calib_copy_kernel<<<cus * 8, 256, 0, st>>>(src, dst, bytes / 16);
    RLDM_HIP_CHECK(hipEventRecord(e1, st));
    RLDM_HIP_CHECK(hipEventSynchronize(e1));
    RLDM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *copy_gbs = 3.0 * 2.0 * (double)bytes / (ms * 1e-3) / 1e9;
    (void)hipFree(src); (void)hipFree(dst); (void)hipFree(sink); (void)hipFree(clk);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}

int rldm_calib_clock_stamp(unsigned long long* slots, void* stream) {
    RLDM_REQUIRE(slots != nullptr, "null argument");
    calib_stamp_kernel<<<RLDM_CALIB_STAMP_BLOCKS, 64, 0, (hipStream_t)stream>>>(slots);
    RLDM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"
