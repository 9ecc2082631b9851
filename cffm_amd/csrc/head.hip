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
    head_fwd_body(a, blockIdx.x, smem);
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

struct HeadBwdArgs {
    Geo g;
    int B;
    int64_t Bg;
    const float *fb, *t1, *h1, *att, *out, *y, *Ctop;
    const float *d1_w, *d2_w, *att_W, *lin_w;
    float* scalars;
    const float* sqerr;      // non-NULL: sum the B local loss terms here instead of reading scalars[3]
    float* loss_out;         // may be NULL
    float *dout, *dt1, *dfb, *dCtop;
    float *s_attW, *s_attb, *s_bias, *s_d1w, *s_d1b, *s_d2w, *s_d2b, *s_linw, *s_linb;   // slab 0 pointers
    int64_t stride_front, stride_back;    // slab strides of the head-front (att_W, att_b, bias) and head-back ranges
    int64_t front_len, back_len;
    int loss, outer_conv;
    int unscaled;            // 1: leave the 1/L of the RMSE-style loss out of dout (data-parallel late scaling)
};

__global__ __launch_bounds__(256) void head_bwd_kernel(HeadBwdArgs a) {
    __shared__ float dh1s[CFFM_HEAD_UNITS];
    __shared__ float dt1s[1024];
    __shared__ float red[4];
    const Geo& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t1w = 2 * g.D - 2, FF = g.F * g.F;
    const int64_t sof = (int64_t)blockIdx.x * a.stride_front, sob = (int64_t)blockIdx.x * a.stride_back;
    float* s_attW = a.s_attW + sof; float* s_d1w = a.s_d1w + sob;
    // zero this slab's two ranges first (alignment gaps and members this configuration never writes)
    for (int64_t e = tid; e < a.front_len; e += 256) s_attW[e] = 0.f;        // att_W is the first member of the range
    for (int64_t e = tid; e < a.back_len; e += 256) s_d1w[e] = 0.f;          // d1_w is the first member of the range
    __syncthreads();
    float sum = 0.f;
    float hybrid_log = 0.f;
    if (a.loss == CFFM_LOSS_HYBRID) {      // two sums with different normalisers: taken from out / y directly
        float p_sq = 0.f, p_log = 0.f;
        for (int i = tid; i < a.B; i += 256) {
            const float o = a.out[i], yy = a.y[i];
            p_sq += 0.5f * (yy - o) * (yy - o);
            p_log -= yy * logf(o + 1e-7f) + (1.f - yy) * logf(1.f - o + 1e-7f);
        }
        sum = block_sum(p_sq, red);
        __syncthreads();
        hybrid_log = block_sum(p_log, red);
        __syncthreads();
    } else if (a.unscaled) {
        sum = 0.f;           // not known yet: the caller all-reduces it together with the gradients
    } else if (a.sqerr) {    // same fixed-order sum in every workgroup
        float part = 0.f;
        for (int i = tid; i < a.B; i += 256) part += a.sqerr[i];
        sum = block_sum(part, red);
        __syncthreads();
    } else {
        sum = a.scalars[3];
    }
    const float invB = 1.f / (float)a.Bg;
    float L;
    if (a.loss == CFFM_LOSS_SQUARE_RMSE) L = a.unscaled ? 1.f : sqrtf(sum * invB + 1e-10f);   // CFFM.py:493
    else if (a.loss == CFFM_LOSS_SQUARE_L2) L = sum;            // data term only (the regularisers are not summed here)
    else if (a.loss == CFFM_LOSS_HYBRID) L = 0.5f * sum + 0.5f * hybrid_log * invB;   // CFFM.py:511-513
    else L = sum * invB;
    if (blockIdx.x == 0 && tid == 0) {
        a.scalars[1] = L;
        if (a.sqerr) { a.scalars[0] = sum; a.scalars[3] = sum; }
        if (a.loss_out) a.loss_out[0] = L;
    }
    float g_d2w = 0.f, g_d1b = 0.f, g_d2b = 0.f, g_bias = 0.f, g_linw = 0.f, g_linb = 0.f, g_attb = 0.f;
    bool first = true;
    const int off_top = [&] { int o = 0; for (int i = 0; i < g.live; ++i) o += g.D >> i; return o; }();
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float out = a.out[b], y = a.y[b];
        float d;
        switch (a.loss) {
            case CFFM_LOSS_SQUARE_RMSE: d = (out - y) * invB / L; break;
            case CFFM_LOSS_MSE: d = 2.f * (out - y) * invB; break;
            case CFFM_LOSS_MAE: d = (out > y ? 1.f : (out < y ? -1.f : 0.f)) * invB; break;
            case CFFM_LOSS_SQUARE_L2: d = out - y; break;        // d/dout of sum (y - out)^2 / 2
            case CFFM_LOSS_HYBRID:
                d = 0.5f * (out - y) - 0.5f * invB * (y / (out + 1e-7f) - (1.f - y) / (1.f - out + 1e-7f));
                break;
            default: {
                const float s = out;   // ws.out holds sigmoid(logit) for log_loss
                d = -(y / (s + 1e-7f) - (1.f - y) / (1.f - s + 1e-7f)) * invB * s * (1.f - s);
            }
        }
        if (tid == 0) { a.dout[b] = d; g_bias += d; }
        __syncthreads();
        if (a.outer_conv) {
            const float dd = d * g.beta_outer;
            if (tid < CFFM_HEAD_UNITS) {
                const float v = dd * a.d2_w[tid];
                dh1s[tid] = v;
                g_d2w += a.h1[(int64_t)b * CFFM_HEAD_UNITS + tid] * dd;
                g_d1b += v;
            }
            if (tid == 0) g_d2b += dd;
            __syncthreads();
            for (int k = tid; k < t1w; k += 256) {
                float s = 0.f;
                for (int q = 0; q < CFFM_HEAD_UNITS; ++q) s += dh1s[q] * a.d1_w[k * CFFM_HEAD_UNITS + q];
                dt1s[k] = s;
                a.dt1[(int64_t)b * t1w + k] = s;
            }
            for (int e = tid; e < t1w * CFFM_HEAD_UNITS; e += 256) {
                const int k = e / CFFM_HEAD_UNITS, q = e % CFFM_HEAD_UNITS;
                const float v = a.t1[(int64_t)b * t1w + k] * dh1s[q];
                s_d1w[e] = first ? v : s_d1w[e] + v;
            }
            __syncthreads();
            // gradient wrt the top live conv output: only its sum pool feeds the head
            const int ntop = 4 * g.Pp;
            for (int e = tid; e < ntop; e += 256) {
                const int yy = e / (2 * g.Pp);
                const int64_t idx = (int64_t)b * ntop + e;
                a.dCtop[idx] = dt1s[off_top + yy] * act_relu_grad(a.Ctop[idx], g.act);
            }
        }
        if (wave == 0) {
            const float fbv = lane < g.F ? a.fb[(int64_t)b * g.F + lane] : 0.f;
            if (g.linear_att) {
                const float at = lane < g.F ? a.att[(int64_t)b * g.F + lane] : 0.f;
                const float dg = lane < g.F ? d * a.lin_w[lane] : 0.f;
                const float da = dg * fbv;
                const float sda = wave_sum(da * at);
                const float dz = at * (da - sda) / g.lamda_att;
                float dfbv = dg * at;
                for (int gI = 0; gI < g.F; ++gI) {
                    const float dzg = __shfl(dz, gI, 64);
                    if (lane < g.F) {
                        dfbv += dzg * a.att_W[lane * g.F + gI];
                        const float v = fbv * dzg;
                        s_attW[lane * g.F + gI] = first ? v : s_attW[lane * g.F + gI] + v;
                    }
                }
                if (lane < g.F) a.dfb[(int64_t)b * g.F + lane] = dfbv;
                g_linw += fbv * at * d;
                g_attb += dz;
                if (lane == 0) g_linb += d;
            } else if (lane < g.F) {
                a.dfb[(int64_t)b * g.F + lane] = d;
            }
        }
        first = false;
    }
    (void)FF;
    if (tid < CFFM_HEAD_UNITS) {
        a.s_d2w[sob + tid] = g_d2w;
        a.s_d1b[sob + tid] = g_d1b;
    }
    if (tid < g.F) {
        a.s_linw[sob + tid] = g_linw;
        a.s_attb[sof + tid] = g_attb;
    }
    if (tid == 0) {
        a.s_d2b[sob] = g_d2b;
        a.s_bias[sof] = g_bias;
        a.s_linb[sob] = g_linb;
    }
}

extern "C" int cffm_head_fwd(const cffm_shape_t* s, const float* theta, void* ws, const float* y, int32_t B, void* stream) {
    return cffm_head_fwd_impl(s, theta, ws, y, B, true, (hipStream_t)stream);
}

int cffm_head_fwd_impl(const cffm_shape_t* s, const float* theta, void* ws, const float* y, int32_t B, bool do_sum,
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
    SlabPlan sp;
    make_slab_plan(s, B, tl, &sp);
    const SlabRange& rf = sp.r[sp.head_front];
    const SlabRange& rb = sp.r[sp.head_back];
    float* gf = (float*)(w + wl.gpart) + rf.base - rf.off;        // slab 0 of theta offset x lives at gf + x
    float* gb = (float*)(w + wl.gpart) + rb.base - rb.off;
    HeadBwdArgs a;
    a.g = make_geo(s); a.B = B; a.Bg = B_global;
    a.fb = (const float*)(w + wl.fb); a.t1 = (const float*)(w + wl.t1); a.h1 = (const float*)(w + wl.h1);
    a.att = (const float*)(w + wl.att); a.out = (const float*)(w + wl.out); a.y = y;
    const int top = a.g.live - 1;
    a.Ctop = (const float*)(w + wl.C[top]); a.dCtop = (float*)(w + wl.dC[top]);
    a.d1_w = theta + tl.d1_w; a.d2_w = theta + tl.d2_w; a.att_W = theta + tl.att_W; a.lin_w = theta + tl.lin_w;
    a.scalars = (float*)(w + wl.scalars);
    a.sqerr = local_sum ? (const float*)(w + wl.sqerr) : nullptr;
    a.loss_out = loss_out;
    a.dout = (float*)(w + wl.dout); a.dt1 = (float*)(w + wl.dt1); a.dfb = (float*)(w + wl.dfb);
    a.s_attW = gf + tl.att_W; a.s_attb = gf + tl.att_b; a.s_bias = gf + tl.bias;
    a.s_d1w = gb + tl.d1_w; a.s_d1b = gb + tl.d1_b; a.s_d2w = gb + tl.d2_w; a.s_d2b = gb + tl.d2_b;
    a.s_linw = gb + tl.lin_w; a.s_linb = gb + tl.lin_b;
    a.stride_front = rf.len; a.stride_back = rb.len; a.front_len = rf.len; a.back_len = rb.len;
    a.loss = s->loss; a.outer_conv = s->outer_conv; a.unscaled = unscaled ? 1 : 0;
    hipLaunchKernelGGL(head_bwd_kernel, dim3(small_slabs(B)), dim3(256), 0, (hipStream_t)stream, a);
    CFFM_CHECK_LAUNCH();
    return 0;
}
