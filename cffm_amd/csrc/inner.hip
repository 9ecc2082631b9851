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

// One workgroup per gradient slab; it walks examples slab, slab + NSLAB, ...  Thread t always owns the
// same (p, t) units, so the dense-kernel gradient is accumulated by plain read-modify-write in the
// workgroup's own slab (no atomics, fixed order).  dEi of one example is accumulated in per-wavefront
// private LDS copies (ds_add_f32) that are merged in wavefront order: bitwise reproducible.
__global__ __launch_bounds__(256) void inner_bwd_kernel(Geo g, int B, const float* __restrict__ Ei,
                                                        const float* __restrict__ dout,
                                                        const float* __restrict__ cw_g, const float* __restrict__ cb_g,
                                                        const float* __restrict__ wd, float* __restrict__ dEi,
                                                        float* __restrict__ slab_cw, float* __restrict__ slab_cb,
                                                        float* __restrict__ slab_dw, float* __restrict__ slab_db,
                                                        int64_t slab_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    slab_cw += blockIdx.x * slab_stride; slab_cb += blockIdx.x * slab_stride;
    slab_dw += blockIdx.x * slab_stride; slab_db += blockIdx.x * slab_stride;
    const int FK = g.F * g.K;
    float* E = reinterpret_cast<float*>(smem);                 // [F*K]
    float* dE = E + FK;                                        // [4][F*K]
    uint32_t* lut = reinterpret_cast<uint32_t*>(dE + 4 * FK);  // [Pp]
    float* red = reinterpret_cast<float*>(lut + g.Pp);         // [4]
    const int wave = threadIdx.x >> 6;
    build_pair_lut(lut, g.F, g.Pp);
    // alignment gaps of this slab range must read as zeros in the reduction
    if (threadIdx.x < 8) slab_cw[threadIdx.x] = 0.f;          // inner_cw (4) + inner_cb (2 + 2 pad)
    if (threadIdx.x < 4) slab_db[threadIdx.x] = 0.f;          // inner_db (1 + 3 pad)
    float cw[4] = {cw_g[0], cw_g[1], cw_g[2], cw_g[3]};
    float cb[2] = {cb_g[0], cb_g[1]};
    const int K2 = g.K / 2, units = g.P * K2;
    const float invK2 = 1.f / (float)K2;
    float gcw[4] = {0.f, 0.f, 0.f, 0.f}, gcb[2] = {0.f, 0.f}, gdb = 0.f;
    bool first = true;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        const float4* src = reinterpret_cast<const float4*>(Ei + (int64_t)b * FK);
        for (int i = threadIdx.x; i < FK / 4; i += blockDim.x) reinterpret_cast<float4*>(E)[i] = src[i];
        for (int i = threadIdx.x; i < 4 * FK; i += blockDim.x) dE[i] = 0.f;
        __syncthreads();
        const float db = dout[b];
        float* myE = dE + wave * FK;
        for (int u = threadIdx.x; u < units; u += blockDim.x) {
            const int p = fast_div(u, invK2), t = u - p * K2;
            const InnerUnit v = inner_unit(E, lut, p, t, g.K, cw, cb, g.act);
            const int64_t wi = (int64_t)p * g.K + 2 * t;
            const float2 w2 = *reinterpret_cast<const float2*>(&wd[wi]);
            // dense(1) kernel gradient: flat * dout
            float2 acc2 = first ? make_float2(0.f, 0.f) : *reinterpret_cast<float2*>(&slab_dw[wi]);
            acc2.x += v.s0 * db; acc2.y += v.s1 * db;
            *reinterpret_cast<float2*>(&slab_dw[wi]) = acc2;
            const float ds0 = db * w2.x, ds1 = db * w2.y;
            const float dz0 = ds0 * act_relu_grad(fmaxf(v.z0, 0.f), g.act);
            const float dz1 = ds1 * act_relu_grad(fmaxf(v.z1, 0.f), g.act);
            gcw[0] += dz0 * v.x0; gcw[1] += dz1 * v.x0; gcw[2] += dz0 * v.x1; gcw[3] += dz1 * v.x1;
            gcb[0] += dz0; gcb[1] += dz1;
            const float dmp = ds0 + ds1;
            const bool firstmax = v.x0 >= v.x1;                       // max-pool grad: first element on ties
            const float dx0 = dz0 * cw[0] + dz1 * cw[1] + (firstmax ? dmp : 0.f);
            const float dx1 = dz0 * cw[2] + dz1 * cw[3] + (firstmax ? 0.f : dmp);
            const float dI0 = dx0 * act_grad_f(v.I0, g.act), dI1 = dx1 * act_grad_f(v.I1, g.act);
            atomicAdd(&myE[v.i * g.K + 2 * t], dI0 * v.ejx);
            atomicAdd(&myE[v.i * g.K + 2 * t + 1], dI1 * v.ejy);
            atomicAdd(&myE[v.j * g.K + 2 * t], dI0 * v.eix);
            atomicAdd(&myE[v.j * g.K + 2 * t + 1], dI1 * v.eiy);
        }
        if (threadIdx.x == 0) gdb += db;
        __syncthreads();
        for (int i = threadIdx.x; i < FK; i += blockDim.x)
            dEi[(int64_t)b * FK + i] = ((dE[i] + dE[FK + i]) + dE[2 * FK + i]) + dE[3 * FK + i];
        first = false;
    }
    if (first) {   // this slab saw no example: it must still read as zeros
        for (int64_t i = threadIdx.x; i < (int64_t)g.P * g.K; i += blockDim.x) slab_dw[i] = 0.f;
    }
    float r[7];
    for (int q = 0; q < 4; ++q) r[q] = block_sum(gcw[q], red);
    r[4] = block_sum(gcb[0], red);
    r[5] = block_sum(gcb[1], red);
    r[6] = block_sum(gdb, red);
    if (threadIdx.x == 0) {
        for (int q = 0; q < 4; ++q) slab_cw[q] = r[q];
        slab_cb[0] = r[4]; slab_cb[1] = r[5];
        slab_db[0] = r[6];
    }
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
    const size_t lds = (size_t)(5 * g.F * g.K + g.Pp + 8) * 4;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)inner_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    SlabPlan sp;
    make_slab_plan(s, B, tl, &sp);
    const SlabRange& sr = sp.r[sp.inner];
    float* base = (float*)(w + wl.gpart) + sr.base - sr.off;      // slab 0 of theta offset x lives at base + x
    hipLaunchKernelGGL(inner_bwd_kernel, dim3(sr.nslab), dim3(256), lds, (hipStream_t)stream, g, (int)B,
                       (const float*)(w + wl.Ei), (const float*)(w + wl.dout), theta + tl.inner_cw,
                       theta + tl.inner_cb, theta + tl.inner_dw, (float*)(w + wl.dEi), base + tl.inner_cw,
                       base + tl.inner_cb, base + tl.inner_dw, base + tl.inner_db, (int64_t)sr.len);
    CFFM_CHECK_LAUNCH();
    return 0;
}
