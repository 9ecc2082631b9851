// evaluate()'s metric sums on the device (CFFM.py:607-615): the predictions of a split never travel to the host.
//
//   predictions_bounded = min(max(y_pred, min(y_true)), max(y_true))            CFFM.py:607-609
//   RMSE = sqrt(mean((y_true - predictions_bounded)^2))                         CFFM.py:610-612
//   R2   = 1 - sum((y_true - bounded)^2) / sum((y_true - mean(y_true))^2)       CFFM.py:614 (sklearn r2_score)
//
// The three sums the two metrics need - sum (y - p)^2, sum y, sum y^2 - are accumulated in float64 (the reference
// computes both metrics in float64 on the host).  Two stages, both in a fixed order: EVAL_BLOCKS workgroups leave one
// partial each, the last stage adds the partials in block order onto the running sums, so a split swept in several
// blocks of rows gives the same bits on every run.
#include "internal.hpp"

#define EVAL_BLOCKS 256

__global__ __launch_bounds__(256) void eval_partial_kernel(const float* __restrict__ pred, const float* __restrict__ y,
                                                           int64_t n, float lo, float hi, double* __restrict__ part) {
    __shared__ double red[3][4];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)EVAL_BLOCKS * 256) {
        // np.maximum(..., min) then np.minimum(..., max), which PROPAGATE NaN (fmaxf/fminf would map a NaN prediction to
        // lo and a diverged model would report a finite, plausible RMSE): the sums then become NaN, as sklearn's metrics
        // would refuse the reference's NaN predictions (CFFM.py:607-614)
        const float raw = pred[i];
        const float p = raw != raw ? raw : fminf(fmaxf(raw, lo), hi);
        const double yt = (double)y[i], d = yt - (double)p;
        s0 += d * d; s1 += yt; s2 += yt * yt;
    }
    // fixed-order reduction: lanes by xor butterfly (every lane ends with the same value), waves in wave order
    for (int m = 32; m >= 1; m >>= 1) {
        s0 += __shfl_xor(s0, m, 64); s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = s0; red[1][w] = s1; red[2][w] = s2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        part[(int64_t)blockIdx.x * 3 + k] = ((red[k][0] + red[k][1]) + red[k][2]) + red[k][3];
    }
}

__global__ __launch_bounds__(64) void eval_final_kernel(const double* __restrict__ part, double* __restrict__ sums) {
    if (threadIdx.x < 3) {
        double s = 0.0;
        for (int b = 0; b < EVAL_BLOCKS; ++b) s += part[(int64_t)b * 3 + threadIdx.x];
        sums[threadIdx.x] += s;
    }
}

extern "C" int64_t cffm_eval_scratch_bytes(void) { return (int64_t)EVAL_BLOCKS * 3 * sizeof(double); }

extern "C" int cffm_eval_sums(const float* pred, const float* y, int64_t n, float lo, float hi, void* scratch, double* sums,
                              void* stream) {
    if (n <= 0) return 0;
    if (!pred || !y || !scratch || !sums) return CFFM_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(eval_partial_kernel, dim3(EVAL_BLOCKS), dim3(256), 0, st, pred, y, n, lo, hi, (double*)scratch);
    CFFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(eval_final_kernel, dim3(1), dim3(64), 0, st, (const double*)scratch, sums);
    CFFM_CHECK_LAUNCH();
    return 0;
}
