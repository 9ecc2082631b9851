// Stand-alone probe (not part of libcffm_hip.so): can the 1x2 conv of the inner branch ([x0, x1, 1] -> [z0, z1], CFFM.py:327)
// ride on the idle MFMA pipe of gather_inner_fwd_wide_kernel?
//   hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_unit.hip -o tools/bin/probe_mfma_unit
// A: register / lane layout of v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4 lanes: D[m][n] = C[m][n] + A[m] * B[n]) - checked, not assumed
// B: issue rate of the unit body in three forms, 1024-thread workgroups, one per CU (4 waves per SIMD, the kernel's occupancy):
//      10 VALU (round 3) | 8 VALU + 2 MFMA 4x4x1 | 8 VALU alone (what is left when the MFMAs cost nothing)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void layout_kernel(float* out) {
    const int lane = threadIdx.x;
    const float a = (float)(lane + 1), b = (float)(100 * (lane + 1));
    f32x4 c = (f32x4){0.5f, 0.25f, 0.125f, 0.0625f};
    f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int m = 0; m < 4; ++m) out[lane * 4 + m] = d[m];
}

template <int MODE>
__global__ __launch_bounds__(1024) void unit_kernel(float* out, int iters, float s) {
    f32x2 p[8], acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { p[k] = (f32x2){1.f + threadIdx.x * 1e-6f + k, 1.5f + k}; acc[k] = (f32x2){0.f, 0.f}; }
    const f32x2 s2 = (f32x2){s, s * 1.0001f}, t2 = (f32x2){1e-7f, 2e-7f};
    const int lane = threadIdx.x & 63;
    const float aw0 = (lane & 3) == 0 ? s : ((lane & 3) == 1 ? s * 1.0001f : 0.f);
    const float aw1 = (lane & 3) == 0 ? 1e-7f : ((lane & 3) == 1 ? 2e-7f : 0.f);
    const f32x4 cbv = (f32x4){1e-3f, 2e-3f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            f32x2 x = p[k];
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(s2));
            asm volatile("v_max_f32 %0, 0, %0" : "+v"(x.x));
            asm volatile("v_max_f32 %0, 0, %0" : "+v"(x.y));
            f32x2 z;
            if (MODE == 0) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(z) : "v"(x), "v"(s2), "v"(t2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(z) : "v"(x), "v"(t2));
            } else if (MODE == 1) {
                f32x4 zz = __builtin_amdgcn_mfma_f32_4x4x1f32(aw0, x.x, cbv, 0, 0, 0);
                zz = __builtin_amdgcn_mfma_f32_4x4x1f32(aw1, x.y, zz, 0, 0, 0);
                z = (f32x2){zz[0], zz[1]};
            } else {
                z = x;
            }
            float m;
            asm volatile("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(x.x), "v"(x.y));
            asm volatile("v_max_f32 %0, 0, %0" : "+v"(z.x));
            asm volatile("v_max_f32 %0, 0, %0" : "+v"(z.y));
            f32x2 mm = (f32x2){m, m};
            asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(z) : "v"(mm));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(z), "v"(s2));
        }
    }
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) r += acc[k].x + acc[k].y;
    if (r == -1.f) out[0] = r;
}

static float time_ms(hipStream_t st, int reps, const std::function<void()>& fn) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    fn();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    float* sink;
    CK(hipMalloc(&sink, 1 << 20));
    // ---- A
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, st, sink);
    float h[256];
    CK(hipMemcpyAsync(h, sink, sizeof(h), hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    int bad = 0;
    const float c0[4] = {0.5f, 0.25f, 0.125f, 0.0625f};
    for (int lane = 0; lane < 64; ++lane)
        for (int m = 0; m < 4; ++m) {
            const int blk = lane / 4, n = lane % 4;
            const float want = c0[m] + (float)(4 * blk + m + 1) * (float)(100 * (4 * blk + n + 1));   // A from lane 4b+m, B from lane 4b+n
            if (h[lane * 4 + m] != want) ++bad;
        }
    printf("LAYOUT v_mfma_f32_4x4x1_16b_f32: D[m][n] of block b in lane 4b+n, register m; A[m] from lane 4b+m, B[n] from lane 4b+n: %s\n",
           bad ? "MISMATCH" : "confirmed");
    if (bad)
        for (int lane = 0; lane < 8; ++lane)
            printf("  lane %d: %g %g %g %g\n", lane, h[lane * 4], h[lane * 4 + 1], h[lane * 4 + 2], h[lane * 4 + 3]);
    // ---- B
    const char* names[3] = {"10 VALU (pk_fma x2 for the 1x2 conv)", "8 VALU + 2 MFMA 4x4x1", "8 VALU alone"};
    for (int m = 0; m < 3; ++m) {
        const int iters = 2048, blocks = 256;
        float ms = time_ms(st, 5, [&]() {
            if (m == 0) hipLaunchKernelGGL(unit_kernel<0>, dim3(blocks), dim3(1024), 0, st, sink, iters, 1.0000001f);
            if (m == 1) hipLaunchKernelGGL(unit_kernel<1>, dim3(blocks), dim3(1024), 0, st, sink, iters, 1.0000001f);
            if (m == 2) hipLaunchKernelGGL(unit_kernel<2>, dim3(blocks), dim3(1024), 0, st, sink, iters, 1.0000001f);
        });
        const double units_per_simd = (double)blocks * 16 * iters * 8 / 1024.0;      // unit-waves per SIMD
        printf("UNIT 4 waves/SIMD %-40s %8.3f ms  %6.2f ns per unit-wave per SIMD  (stress shape: 1984 unit-waves per SIMD = %5.1f us)\n",
               names[m], ms, ms * 1e6 / units_per_simd, ms * 1e6 / units_per_simd * 1984 * 1e-3);
    }
    return 0;
}
