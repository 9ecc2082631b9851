// Peak probes for the two rooflines the path is priced against (SURVEY 8d asks for the data-sheet peaks AND what this
// box actually delivers): a float4 streaming copy (HBM) and a dependent-free fp32 MFMA loop (matrix cores).
#include "common.hpp"

__global__ __launch_bounds__(256) void probe_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {          // four loads in flight per lane
        const f32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

extern "C" int cffm_probe_copy(const void* src, void* dst, int64_t bytes, void* stream) {
    if (bytes <= 0 || (bytes & 15)) return CFFM_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(probe_copy_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst,
                       bytes / 16);
    CFFM_CHECK_LAUNCH();
    return 0;
}

// every wave keeps 8 independent 16x16x4 fp32 accumulators busy: 8 * 2*16*16*4 = 16384 flop per iteration per wave
__global__ __launch_bounds__(256) void probe_mfma_kernel(float* __restrict__ out, int iters) {
    f32x4 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + (float)(threadIdx.x & 7) * 1e-3f, b = 1.0f - (float)(threadIdx.x & 3) * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = mfma16(a, b, acc[k]);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += acc[k];
    if (s.x + s.y + s.z + s.w == -1.f) out[0] = s.x;        // never true: keeps the loop alive
}

// returns the flop count of one launch through *flops
extern "C" int cffm_probe_mfma(float* out, int32_t iters, int64_t* flops, void* stream) {
    if (iters <= 0 || !out) return CFFM_ERR_BAD_SHAPE;
    const int blocks = 256 * 8;                              // 8 workgroups of 4 waves per CU
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, (int)iters);
    CFFM_CHECK_LAUNCH();
    if (flops) *flops = (int64_t)blocks * 4 * iters * 8 * (2ll * 16 * 16 * 4);
    return 0;
}
