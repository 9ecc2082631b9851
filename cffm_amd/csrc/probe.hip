// Peak probes for the two rooflines the path is priced against (SURVEY 8d asks for the data-sheet peaks AND what this
// box actually delivers): a float4 streaming copy (HBM) and a dependent-free fp32 MFMA loop (matrix cores).
#include "common.hpp"

// Each workgroup moves contiguous 16 KB tiles (4 x 16 B per lane, all four loads issued before the first store); tiles are
// dealt round-robin over a grid that fills the chip 8 workgroups deep.  (The first version strode its four loads 16 MB
// apart and reached 4.3-4.6 TB/s; contiguous tiles are what the 6.3 TB/s float4-copy figure of the microarchitecture
// guide is measured with.)
__global__ __launch_bounds__(256) void probe_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n4) {
    const int64_t tiles = n4 >> 10;                          // 1024 float4 per tile
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t i = (t << 10) + threadIdx.x;
        const f32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + 256),
                    c = __builtin_nontemporal_load(src + i + 512), d = __builtin_nontemporal_load(src + i + 768);
        __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + 256);
        __builtin_nontemporal_store(c, dst + i + 512); __builtin_nontemporal_store(d, dst + i + 768);
    }
    for (int64_t i = (tiles << 10) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

extern "C" int cffm_probe_copy(const void* src, void* dst, int64_t bytes, void* stream) {
    if (bytes <= 0 || (bytes & 15)) return CFFM_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(probe_copy_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst,
                       bytes / 16);
    CFFM_CHECK_LAUNCH();
    return 0;
}

// read-only stream: what the HBM READ roofline of a gather is priced against.  Nothing is written (the compare on the
// running sum keeps the loads alive; it practically never holds)
__global__ __launch_bounds__(256) void probe_read_kernel(const f32x4* __restrict__ src, float* __restrict__ sink, int64_t n4) {
    const int64_t tiles = n4 >> 10;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t i = (t << 10) + threadIdx.x;
        const f32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + 256),
                    c = __builtin_nontemporal_load(src + i + 512), d = __builtin_nontemporal_load(src + i + 768);
        s += (a + b) + (c + d);
    }
    if (s.x + s.y + s.z + s.w == 123456.789f) sink[blockIdx.x] = s.x;     // practically never: keeps the loop alive
}

extern "C" int cffm_probe_read(const void* src, void* sink, int64_t bytes, void* stream) {
    if (bytes <= 0 || (bytes & 15) || !sink) return CFFM_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(probe_read_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (float*)sink,
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

// the same probe on the bf16 pipe: 8 independent v_mfma_f32_16x16x32_bf16 accumulators per wave (2 * 16 * 16 * 32 = 16384 flop each):
// what the bf16x3 conv loops (conv.hip) are priced against - the data sheet says 2.5 PFLOP/s dense
typedef __bf16 probe_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned probe_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void probe_mfma_bf16_kernel(float* __restrict__ out, int iters) {
    f32x4 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned w = 0x3f803f80u + (threadIdx.x & 7);                 // (1.0, 1.0) in bf16, a few low bits varied
    const probe_u32x4 av = {w, w, w, w}, bv = {w ^ 1u, w, w ^ 2u, w};
    const probe_bf16x8 a = __builtin_bit_cast(probe_bf16x8, av), b = __builtin_bit_cast(probe_bf16x8, bv);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += acc[k];
    if (s.x + s.y + s.z + s.w == -1.f) out[0] = s.x;        // never true: keeps the loop alive
}
extern "C" int cffm_probe_mfma_bf16(float* out, int32_t iters, int64_t* flops, void* stream) {
    if (iters <= 0 || !out) return CFFM_ERR_BAD_SHAPE;
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(probe_mfma_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, (int)iters);
    CFFM_CHECK_LAUNCH();
    if (flops) *flops = (int64_t)blocks * 4 * iters * 8 * (2ll * 16 * 16 * 32);
    return 0;
}
