// Stand-alone probe: sustained rate of the bf16 MFMA forms on this box (dependency-free accumulators, every CU busy).
//   hipcc -O3 --offload-arch=gfx950 tools/probe_bf16_peak.hip -o tools/bin/probe_bf16_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
    f32x4 acc[NACC];
    for (int k = 0; k < NACC; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned w = 0x3f803f80u + (threadIdx.x & 7);
    const u32x4 av = {w, w, w, w}, bv = {w ^ 1u, w, w ^ 2u, w};
    const bf16x8 a = __builtin_bit_cast(bf16x8, av), b = __builtin_bit_cast(bf16x8, bv);
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
    f32x4 s = acc[0];
    for (int k = 1; k < NACC; ++k) s += acc[k];
    if (s.x + s.y + s.z + s.w == -1.f) out[0] = s.x;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k)
        for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
    const unsigned w = 0x3f803f80u + (threadIdx.x & 7);
    const u32x4 av = {w, w, w, w}, bv = {w ^ 1u, w, w ^ 2u, w};
    const bf16x8 a = __builtin_bit_cast(bf16x8, av), b = __builtin_bit_cast(bf16x8, bv);
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
    float s = 0.f;
    for (int k = 0; k < NACC; ++k)
        for (int j = 0; j < 16; ++j) s += acc[k][j];
    if (s == -1.f) out[0] = s;
}
template <class F>
static float time_ms(hipStream_t st, F fn) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    fn(); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < 5; ++i) fn();
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 5;
}
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    float* sink; CK(hipMalloc(&sink, 4096));
    const int iters = 20000;
    for (int wgs = 2; wgs <= 8; wgs *= 2) {
        const int blocks = 256 * wgs;
        float ms = time_ms(st, [&] { hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(256), 0, st, sink, iters); });
        printf("16x16x32 bf16, 8 accumulators/wave, %d waves/SIMD: %7.1f TFLOP/s\n", wgs, (double)blocks * 4 * iters * 8 * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12);
        ms = time_ms(st, [&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, st, sink, iters); });
        printf("16x16x32 bf16, 4 accumulators/wave, %d waves/SIMD: %7.1f TFLOP/s\n", wgs, (double)blocks * 4 * iters * 4 * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12);
        ms = time_ms(st, [&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, st, sink, iters); });
        printf("32x32x16 bf16, 4 accumulators/wave, %d waves/SIMD: %7.1f TFLOP/s\n", wgs, (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
