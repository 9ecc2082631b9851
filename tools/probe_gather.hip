// Stand-alone probe (not part of libcffm_hip.so): what bounds a read-only, consumer-fused embedding gather on this box.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_gather.hip -o tools/bin/probe_gather
// A: VALU issue rate of plain vs packed fp32 (v_fma_f32 / v_mul_f32 / v_max_f32 against v_pk_fma_f32 / v_pk_mul_f32)
// B: random whole-record reads of a [M][stride] fp32 table (M = 1M), 8192 x 32 lookups, records summed in registers:
//    record strides 512 / 528 / 640 B, pieces read per record 32 (rows only) / 33 (rows + bias in the record), and the
//    bias as a separate 4-byte gather (per-lookup, or in sorted-id order)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <functional>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- A: VALU ---------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void valu_kernel(float* out, int iters, float s) {
    float a[8];
    f32x2 p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = 1.f + threadIdx.x * 1e-6f + k; p[k] = (f32x2){a[k], a[k] + 0.5f}; }
    const f32x2 s2 = (f32x2){s, s * 1.0001f}, t2 = (f32x2){1e-7f, 2e-7f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(s), "v"(s2.y));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(s2), "v"(t2));
            if (MODE == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(s));
            if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(s2));
            if (MODE == 4) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(s));
            if (MODE == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(t2));
            if (MODE == 6) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[k]) : "v"(s));
            if (MODE == 7) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(s), "v"(s2.y));
            if (MODE == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(s));
            if (MODE == 9) asm volatile("v_max_f32 %0, 0, %0" : "+v"(a[k]));
            if (MODE == 10) {   // the unit body of gather_inner_fwd_wide_kernel: 10 instructions
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(s2));
                asm volatile("v_max_f32 %0, 0, %0" : "+v"(p[k].x));
                asm volatile("v_max_f32 %0, 0, %0" : "+v"(p[k].y));
                f32x2 z;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(z) : "v"(p[k]), "v"(s2), "v"(t2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(z) : "v"(p[k]), "v"(t2));
                float m;
                asm volatile("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(p[k].x), "v"(p[k].y));
                asm volatile("v_max_f32 %0, 0, %0" : "+v"(z.x));
                asm volatile("v_max_f32 %0, 0, %0" : "+v"(z.y));
                f32x2 mm = (f32x2){m, m};
                asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(z) : "v"(mm));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(z), "v"(s2));
            }
        }
    }
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) r += a[k] + p[k].x + p[k].y;
    if (r == -1.f) out[0] = r;
}

// ---- B: gather -------------------------------------------------------------------------------------------------------------
// Work = 16-byte pieces.  Lookup l (= example * F + field) has PPR pieces; a workgroup owns a contiguous range of lookups and
// walks its pieces THREADS at a time, U independent loads in flight per lane before they are consumed.
template <int THREADS, int U, int PPR, int BIAS>   // BIAS: 0 none, 1 separate 4-byte gather per lookup (lane per lookup)
__global__ __launch_bounds__(THREADS) void gather_kernel(const char* __restrict__ table, int64_t stride, const float* __restrict__ bias,
                                                         const int32_t* __restrict__ ids, int n_lookups, int per_wg,
                                                         float* __restrict__ sink) {
    const int l0 = blockIdx.x * per_wg, l1 = min(l0 + per_wg, n_lookups);
    const int total = (l1 - l0) * PPR;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int g0 = threadIdx.x; g0 < total; g0 += THREADS * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int g = g0 + u * THREADS;
            v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (g < total) {
                const int l = PPR == 32 ? (g >> 5) : g / PPR;
                const int pc = g - l * PPR;
                const int id = ids[l0 + l];
                if (BIAS >= 2) v[u] = *reinterpret_cast<const f32x4*>(table + (pc >= 16 ? (int64_t)256000000 : 0) + (int64_t)id * 256 + (pc & 15) * 16);
                else v[u] = *reinterpret_cast<const f32x4*>(table + (int64_t)id * stride + pc * 16);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    float b = 0.f;
    if (BIAS == 1 || BIAS == 3) {
        for (int l = l0 + threadIdx.x; l < l1; l += THREADS) b += bias[ids[l]];
    }
    const float r = acc.x + acc.y + acc.z + acc.w + b;
    if (r == 123456.789f) sink[blockIdx.x] = r;
}

// bias in sorted-id order: sorted [n] int32 ids, one lane per lookup
__global__ __launch_bounds__(256) void bias_sorted_kernel(const float* __restrict__ bias, const int32_t* __restrict__ sorted, int n,
                                                          float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = bias[sorted[i]];
}

// rows into LDS (ds_write_b128) instead of registers; one barrier per group of examples
template <int THREADS, int U, int PPR>
__global__ __launch_bounds__(THREADS) void gather_lds_kernel(const char* __restrict__ table, int64_t stride,
                                                             const int32_t* __restrict__ ids, int n_lookups, int per_wg,
                                                             float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* L = reinterpret_cast<f32x4*>(smem);
    const int l0 = blockIdx.x * per_wg, l1 = min(l0 + per_wg, n_lookups);
    const int total = (l1 - l0) * PPR;
    float r = 0.f;
    for (int g0 = threadIdx.x; g0 < total; g0 += THREADS * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int g = g0 + u * THREADS;
            v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (g < total) {
                const int l = PPR == 32 ? (g >> 5) : g / PPR;
                const int pc = g - l * PPR;
                const int id = ids[l0 + l];
                v[u] = *reinterpret_cast<const f32x4*>(table + (int64_t)id * stride + pc * 16);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) L[u * THREADS + threadIdx.x] = v[u];
        __syncthreads();
        r += smem[(threadIdx.x * 4) & 1023];
        __syncthreads();
    }
    if (r == 123456.789f) sink[blockIdx.x] = r;
}

static float time_ms(hipStream_t st, int iters, const std::function<void(int)>& fn) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) fn(i);
    CK(hipEventRecord(a, st));
    for (int i = 0; i < iters; ++i) fn(i);
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    float* sink;
    CK(hipMalloc(&sink, 1 << 20));
    // ---- A
    {
        const char* names[11] = {"v_fma_f32", "v_pk_fma_f32", "v_mul_f32", "v_pk_mul_f32", "v_max_f32", "v_pk_add_f32", "v_max_i32",
                                 "v_max3_f32", "v_add_f32", "v_max_f32 0,x", "unit body (10 instr)"};
        for (int wps = 8; wps >= 2; wps >>= 1)
        for (int m = 0; m < 11; ++m) {
            const int iters = 4096, blocks = 256 * wps;
            float ms = time_ms(st, 5, [&](int) {
                switch (m) {
                    case 0: hipLaunchKernelGGL(valu_kernel<0>, dim3(blocks), dim3(256), 0, st, sink, iters, 1.0000001f); break;
                    case 1: hipLaunchKernelGGL(valu_kernel<1>, dim3(blocks), dim3(256), 0, st, sink, iters, 1.0000001f); break;
                    case 2: hipLaunchKernelGGL(valu_kernel<2>, dim3(blocks), dim3(256), 0, st, sink, iters, 1.0000001f); break;
                    case 3: hipLaunchKernelGGL(valu_kernel<3>, dim3(blocks), dim3(256), 0, st, sink, iters, 1.0000001f); break;
                    case 4: hipLaunchKernelGGL(valu_kernel<4>, dim3(blocks), dim3(256), 0, st, sink, iters, 1.0000001f); break;
                    case 5: hipLaunchKernelGGL(valu_kernel<5>, dim3(blocks), dim3(256), 0, st, sink, iters, 1.0000001f); break;
                    case 6: hipLaunchKernelGGL(valu_kernel<6>, dim3(blocks), dim3(256), 0, st, sink, iters, 1.0000001f); break;
                    case 7: hipLaunchKernelGGL(valu_kernel<7>, dim3(blocks), dim3(256), 0, st, sink, iters, 1.0000001f); break;
                    case 8: hipLaunchKernelGGL(valu_kernel<8>, dim3(blocks), dim3(256), 0, st, sink, iters, 1.0000001f); break;
                    case 9: hipLaunchKernelGGL(valu_kernel<9>, dim3(blocks), dim3(256), 0, st, sink, iters, 1.0000001f); break;
                    default: hipLaunchKernelGGL(valu_kernel<10>, dim3(blocks), dim3(256), 0, st, sink, iters, 1.0000001f); break;
                }
            });
            const double winstr = (double)blocks * 4 * iters * 8 * (m == 10 ? 10 : 1);   // wave-instructions
            const double lane_elems = (double)blocks * 256 * iters * 8 * ((m & 1) ? 2 : 1);
            printf("VALU %d waves/SIMD %-22s %8.3f ms  %7.2f T wave-lane-instr/s  %6.2f ns per wave-instr per SIMD\n", wps, names[m], ms,
                   winstr * 64 / (ms * 1e-3) / 1e12, ms * 1e6 / (winstr / 1024.0));
            (void)lane_elems;
        }
    }
    if (getenv("PROBE_VALU_ONLY")) return 0;
    // ---- B
    const int M = 1000000, F = 32, B = 8192, NSET = 8;
    const int n = B * F;
    const int64_t max_stride = 640;
    char* table;
    CK(hipMalloc(&table, (size_t)M * max_stride + 4096));
    CK(hipMemset(table, 0, (size_t)M * max_stride + 4096));
    float* bias;
    CK(hipMalloc(&bias, (size_t)M * 4));
    CK(hipMemset(bias, 0, (size_t)M * 4));
    std::vector<int32_t> h((size_t)NSET * n), hs((size_t)NSET * n);
    srand(2021);
    const int w = (M + F - 1) / F;
    for (int s = 0; s < NSET; ++s)
        for (int b = 0; b < B; ++b)
            for (int f = 0; f < F; ++f) {
                const int lo = f * w, hi = std::min((f + 1) * w, M);
                h[(size_t)s * n + b * F + f] = lo + (int)(((uint64_t)rand() * 2147483648ull + rand()) % (uint64_t)(hi - lo));
            }
    hs = h;
    for (int s = 0; s < NSET; ++s) std::sort(hs.begin() + (size_t)s * n, hs.begin() + (size_t)(s + 1) * n);
    int32_t *ids, *sorted;
    CK(hipMalloc(&ids, h.size() * 4)); CK(hipMalloc(&sorted, h.size() * 4));
    CK(hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sorted, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    float* bout;
    CK(hipMalloc(&bout, (size_t)n * 4));
    const double alg = (double)n * 520.0;

#define RUN(NAME, KERNEL, THREADS, PER_WG, STRIDE, LDS)                                                                      \
    do {                                                                                                                      \
        const int per_wg = (PER_WG);                                                                                          \
        const int blocks = (n + per_wg - 1) / per_wg;                                                                         \
        float ms = time_ms(st, 20, [&](int i) {                                                                               \
            hipLaunchKernelGGL(KERNEL, dim3(blocks), dim3(THREADS), LDS, st, (const char*)table, (int64_t)(STRIDE),            \
                               ids + (size_t)(i % NSET) * n, n, per_wg, sink);                                                \
        });                                                                                                                   \
        CK(hipGetLastError());                                                                                                \
        printf("GATHER %-58s %7.2f us  %6.0f GB/s alg  frac %.3f\n", NAME, ms * 1e3, alg / (ms * 1e-3) / 1e9,                 \
               alg / (ms * 1e-3) / 8e12);                                                                                     \
    } while (0)
#define KB(T, U, PPR, BIAS) (gather_kernel<T, U, PPR, BIAS>)
#define RUNB(NAME, KERNEL, THREADS, PER_WG, STRIDE)                                                                           \
    do {                                                                                                                      \
        const int per_wg = (PER_WG);                                                                                          \
        const int blocks = (n + per_wg - 1) / per_wg;                                                                         \
        float ms = time_ms(st, 20, [&](int i) {                                                                               \
            hipLaunchKernelGGL(KERNEL, dim3(blocks), dim3(THREADS), 0, st, (const char*)table, (int64_t)(STRIDE),              \
                               (const float*)bias, ids + (size_t)(i % NSET) * n, n, per_wg, sink);                            \
        });                                                                                                                   \
        CK(hipGetLastError());                                                                                                \
        printf("GATHER %-58s %7.2f us  %6.0f GB/s alg  frac %.3f\n", NAME, ms * 1e3, alg / (ms * 1e-3) / 1e9,                 \
               alg / (ms * 1e-3) / 8e12);                                                                                     \
    } while (0)

    // rows only, stride 512: threads x in-flight x lookups per workgroup
    RUNB("s512 p32 nobias  T256 U4  wg=32 lookups (1 example)", KB(256, 4, 32, 0), 256, 32, 512);
    RUNB("s512 p32 nobias  T256 U8  wg=64", KB(256, 8, 32, 0), 256, 64, 512);
    RUNB("s512 p32 nobias  T256 U8  wg=128", KB(256, 8, 32, 0), 256, 128, 512);
    RUNB("s512 p32 nobias  T256 U16 wg=128", KB(256, 16, 32, 0), 256, 128, 512);
    RUNB("s512 p32 nobias  T512 U4  wg=128", KB(512, 4, 32, 0), 512, 128, 512);
    RUNB("s512 p32 nobias  T512 U8  wg=128", KB(512, 8, 32, 0), 512, 128, 512);
    RUNB("s512 p32 nobias  T1024 U2 wg=256 (8 examples)", KB(1024, 2, 32, 0), 1024, 256, 512);
    RUNB("s512 p32 nobias  T1024 U4 wg=256", KB(1024, 4, 32, 0), 1024, 256, 512);
    RUNB("s512 p32 nobias  T1024 U4 wg=1024 (32 examples, 256 wgs)", KB(1024, 4, 32, 0), 1024, 1024, 512);
    RUNB("s512 p32 nobias  T1024 U8 wg=1024", KB(1024, 8, 32, 0), 1024, 1024, 512);
    RUNB("s512 p32 nobias  T1024 U8 wg=512", KB(1024, 8, 32, 0), 1024, 512, 512);
    // separate 4-byte bias gather per lookup
    RUNB("s512 p32 +bias4B T1024 U4 wg=256", KB(1024, 4, 32, 1), 1024, 256, 512);
    RUNB("s512 p32 +bias4B T256 U8  wg=128", KB(256, 8, 32, 1), 256, 128, 512);
    // two separate tables of 256-byte rows (inner [M][64], outer [M][64]) instead of one 512-byte record
    RUNB("2x256 separate tables nobias T1024 U4 wg=256", KB(1024, 4, 32, 2), 1024, 256, 512);
    RUNB("2x256 separate tables nobias T256 U8 wg=128", KB(256, 8, 32, 2), 256, 128, 512);
    RUNB("2x256 separate tables +bias4B T1024 U4 wg=256", KB(1024, 4, 32, 3), 1024, 256, 512);
    RUNB("2x256 separate tables +bias4B T256 U8 wg=128", KB(256, 8, 32, 3), 256, 128, 512);
    RUNB("2x256 separate tables +bias4B T512 U8 wg=128", KB(512, 8, 32, 3), 512, 128, 512);
    // bias inside the record
    RUNB("s528 p33 inrec   T1024 U4 wg=256", KB(1024, 4, 33, 0), 1024, 256, 528);
    RUNB("s528 p33 inrec   T256 U8  wg=128", KB(256, 8, 33, 0), 256, 128, 528);
    RUNB("s640 p33 inrec   T1024 U4 wg=256", KB(1024, 4, 33, 0), 1024, 256, 640);
    RUNB("s640 p33 inrec   T256 U8  wg=128", KB(256, 8, 33, 0), 256, 128, 640);
    RUNB("s576 p33 inrec   T1024 U4 wg=256", KB(1024, 4, 33, 0), 1024, 256, 576);
    // into LDS
    RUN("s512 p32 LDS     T1024 U4 wg=256", (gather_lds_kernel<1024, 4, 32>), 1024, 256, 512, 1024 * 4 * 16);
    RUN("s512 p32 LDS     T256 U8  wg=128", (gather_lds_kernel<256, 8, 32>), 256, 128, 512, 256 * 8 * 16);
    // bias alone, sorted order
    {
        float ms = time_ms(st, 20, [&](int i) {
            hipLaunchKernelGGL(bias_sorted_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)bias,
                               sorted + (size_t)(i % NSET) * n, n, bout);
        });
        printf("BIAS sorted order, 262144 lookups: %7.2f us\n", ms * 1e3);
        ms = time_ms(st, 20, [&](int i) {
            hipLaunchKernelGGL(bias_sorted_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)bias,
                               ids + (size_t)(i % NSET) * n, n, bout);
        });
        printf("BIAS batch order,  262144 lookups: %7.2f us\n", ms * 1e3);
    }
    return 0;
}
