// Head of the CFFM graph: sum pooling over the conv stack (CFFM.py:381, :390-396), the two dense
// layers of the outer branch (:409-414), the linear-attention first-order term (:422-446), add_n (:453),
// the loss terms (:486-514) and all of their gradients.
//
// Everything here is tiny per example except the pooling sweep, which re-reads the conv outputs once
// (contiguous S*Pp-float rows, 16-byte loads) - HBM/L2-bound and bitwise reproducible (fixed tree).
#include "internal.hpp"

#include "head_body.hpp"

__global__ __launch_bounds__(256) void head_fwd_kernel(HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    head_fwd_body<4, -1, true>(a, blockIdx.x, smem);
}

// deterministic single-workgroup sum of n floats -> dst[0] (and dst[3] when mirror != 0)
__global__ __launch_bounds__(1024) void sum_kernel(const float* __restrict__ x, int64_t n, float* dst, int mirror) {
    __shared__ float red[16];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += x[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        dst[0] = s;
        if (mirror) dst[3] = s;
    }
}

__global__ __launch_bounds__(256) void head_bwd_kernel(HeadBwdArgs a) {
    __shared__ float dh1s[CFFM_HEAD_UNITS];
    __shared__ float dt1s[1024];
    __shared__ float red[4];
    HeadBwdState st;
    head_bwd_begin(a, blockIdx.x, st);
    const float L = head_bwd_loss(a, blockIdx.x == 0, red);
    const float invB = 1.f / (float)a.Bg;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x)
        head_bwd_example(a, blockIdx.x, st, b, head_dout(a.loss, a.out[b], a.y[b], invB, L), dh1s, dt1s);
    head_bwd_end(a, blockIdx.x, st);
}

extern "C" int cffm_head_fwd(const cffm_shape_t* s, const float* theta, void* ws, const float* y, int32_t B, void* stream) {
    return cffm_head_fwd_impl(s, theta, ws, y, B, true, (hipStream_t)stream);
}

int cffm_head_fwd_impl(const cffm_shape_t* s, const float* theta, void* ws, const float* y, int32_t B, bool do_sum,
                       hipStream_t stream) {
    return cffm_head_fwd_impl2(s, theta, ws, y, B, do_sum, false, stream);
}

int cffm_head_fwd_impl2(const cffm_shape_t* s, const float* theta, void* ws, const float* y, int32_t B, bool do_sum, bool s0_ready,
                        hipStream_t stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (B <= 0) return 0;
    if (2 * s->D - 2 > 1024) return CFFM_ERR_UNSUPPORTED;
    cffm_theta_layout_t tl; cffm_ws_layout_t wl;
    cffm_theta_layout(s, &tl); cffm_ws_layout(s, B, &wl);
    char* w = (char*)ws;
    HeadArgs a;
    a.g = make_geo(s); a.B = B;
    a.Eo = (const float*)(w + wl.Eo); a.fb = (const float*)(w + wl.fb);
    a.inner_out = (const float*)(w + wl.inner_out);
    for (int l = 0; l < CFFM_MAX_LAYERS; ++l) a.C[l] = (const float*)(w + wl.C[l]);
    a.d1_w = theta + tl.d1_w; a.d1_b = theta + tl.d1_b; a.d2_w = theta + tl.d2_w; a.d2_b = theta + tl.d2_b;
    a.att_W = theta + tl.att_W; a.att_b = theta + tl.att_b; a.lin_w = theta + tl.lin_w; a.lin_b = theta + tl.lin_b;
    a.bias = theta + tl.bias;
    a.y = y;
    a.t1 = (float*)(w + wl.t1); a.h1 = (float*)(w + wl.h1); a.att = (float*)(w + wl.att);
    a.out = (float*)(w + wl.out); a.sqerr = (float*)(w + wl.sqerr);
    a.loss = s->loss; a.inner_conv = s->inner_conv; a.outer_conv = s->outer_conv;
    a.s0_ready = (s0_ready && s->outer_conv && s->D <= 256) ? 1 : 0;
    for (int l = 0; l < CFFM_MAX_LAYERS; ++l) {
        a.pool_np[l] = wl.pool_np[l];
        a.pool[l] = wl.pool_np[l] > 0 ? (const float*)(w + wl.pool[l]) : nullptr;
    }
    hipLaunchKernelGGL(head_fwd_kernel, dim3(B), dim3(256), head_fwd_lds(a.g), (hipStream_t)stream, a);
    CFFM_CHECK_LAUNCH();
    if (y && do_sum) {
        hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)(w + wl.sqerr),
                           (int64_t)B, (float*)(w + wl.scalars), 1);
        CFFM_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int cffm_head_bwd(const cffm_shape_t* s, const float* theta, void* ws, const float* y, int32_t B,
                             int64_t B_global, void* stream) {
    return cffm_head_bwd_impl(s, theta, ws, y, B, B_global, false, nullptr, (hipStream_t)stream);
}

int cffm_head_bwd_impl(const cffm_shape_t* s, const float* theta, void* ws, const float* y, int32_t B, int64_t B_global,
                       bool local_sum, float* loss_out, hipStream_t stream, bool unscaled) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (B <= 0) return 0;
    if (2 * s->D - 2 > 1024) return CFFM_ERR_UNSUPPORTED;
    cffm_theta_layout_t tl; cffm_ws_layout_t wl;
    cffm_theta_layout(s, &tl); cffm_ws_layout(s, B, &wl);
    char* w = (char*)ws;
    HeadBwdArgs a;
    fill_head_bwd_args(s, theta, ws, y, B, B_global, local_sum, loss_out, unscaled, &a);
    hipLaunchKernelGGL(head_bwd_kernel, dim3(small_slabs(B)), dim3(256), 0, (hipStream_t)stream, a);
    CFFM_CHECK_LAUNCH();
    return 0;
}
