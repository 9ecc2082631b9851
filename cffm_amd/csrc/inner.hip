// Inner (element-wise product) branch of the CFFM graph, CFFM.py:301-343, and its gradient.
//
// Per example: P pairwise products of K-vectors -> act -> 1x2/stride-2 conv (1 -> 2 channels) + bias
// -> relu -> act -> + max-pool(2) of the activated products -> flatten (p, t, ch) -> dense(1).
// Everything is element-wise followed by ONE dot product, so the whole branch is a single pass over
// the gathered rows held in LDS with a wavefront/block reduction at the end: nothing of the
// [B,P,K/2,2] intermediate (1 GB at F=32,K=64,B=8192) is ever written.
#include "internal.hpp"

#include "inner_body.hpp"

__global__ __launch_bounds__(256) void inner_fwd_kernel(InnerFwdArgs ia) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    inner_fwd_body(ia, blockIdx.x, smem);
}

__global__ __launch_bounds__(256) void inner_bwd_kernel(InnerBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    inner_bwd_body(a, blockIdx.x, gridDim.x, smem);
}

extern "C" int cffm_inner_fwd(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, void* stream) {
    return cffm_inner_fwd_impl(s, theta, ws, B, nullptr, nullptr, (hipStream_t)stream);
}

int cffm_inner_fwd_impl(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, const cffm_tables_t* tab,
                        const int32_t* ids, hipStream_t stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (B <= 0 || !s->inner_conv) return 0;
    cffm_theta_layout_t tl; cffm_ws_layout_t wl;
    cffm_theta_layout(s, &tl); cffm_ws_layout(s, B, &wl);
    const Geo g = make_geo(s);
    char* w = (char*)ws;
    const size_t lds = inner_fwd_lds(g);
    FusedGather fg;
    fg.ids = tab ? ids : nullptr;
    if (tab) {
        fg.inner = tab->inner_emb; fg.outer = tab->outer_emb; fg.fbias = tab->feat_bias;
        fg.Ei = (float*)(w + wl.Ei); fg.Eo = (float*)(w + wl.Eo); fg.fb = (float*)(w + wl.fb);
        fg.keys = (unsigned long long*)(w + wl.sort_keys);
        fg.M = s->M; fg.D = s->D;
    }
    InnerFwdArgs ia;
    ia.g = g; ia.Ei = (const float*)(w + wl.Ei); ia.cw = theta + tl.inner_cw; ia.cb = theta + tl.inner_cb;
    ia.wd = theta + tl.inner_dw; ia.bd = theta + tl.inner_db; ia.inner_out = (float*)(w + wl.inner_out); ia.fg = fg;
    hipLaunchKernelGGL(inner_fwd_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, ia);
    CFFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cffm_inner_bwd(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, void* stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (B <= 0 || !s->inner_conv) return 0;
    cffm_theta_layout_t tl; cffm_ws_layout_t wl;
    cffm_theta_layout(s, &tl); cffm_ws_layout(s, B, &wl);
    const Geo g = make_geo(s);
    char* w = (char*)ws;
    const size_t lds = inner_bwd_lds(g);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)inner_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    InnerBwdArgs a;
    const int nslab = fill_inner_bwd_args(s, theta, ws, B, &a);
    hipLaunchKernelGGL(inner_bwd_kernel, dim3(nslab), dim3(256), lds, (hipStream_t)stream, a);
    CFFM_CHECK_LAUNCH();
    return 0;
}
