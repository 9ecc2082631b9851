// Device body of the head forward (shared by head_fwd_kernel and the fused forward kernel).
#pragma once
#include "common.hpp"

struct HeadArgs {
    Geo g;
    int B;
    const float* Eo;          // [B,F,D]
    const float* fb;          // [B,F]
    const float* inner_out;   // [B]
    const float* C[CFFM_MAX_LAYERS];
    const float *d1_w, *d1_b, *d2_w, *d2_b, *att_W, *att_b, *lin_w, *lin_b, *bias;
    const float* y;           // may be NULL
    float *t1, *h1, *att, *out, *sqerr;
    int loss, inner_conv, outer_conv;
    int s0_ready = 0;         // 1: t1[b][0:D] already holds the s0 pool (cffm_gather_inner_fwd_wide wrote it) and Eo is not read
    const float* pool[CFFM_MAX_LAYERS] = {};   // wide shapes: partial sum pools left by the conv epilogues, [B][S_l][pool_np[l]]
    int pool_np[CFFM_MAX_LAYERS] = {};         // 0: sweep C_l here
};

__device__ __forceinline__ float loss_term(float out_raw, float y, int loss, float* out_eval) {
    switch (loss) {
        case CFFM_LOSS_MAE: *out_eval = out_raw; return fabsf(y - out_raw);
        case CFFM_LOSS_SQUARE_L2: *out_eval = out_raw; return 0.5f * (y - out_raw) * (y - out_raw);   // tf.nn.l2_loss
        case CFFM_LOSS_LOG: {
            const float s = 1.f / (1.f + expf(-out_raw));
            *out_eval = s;                                         // CFFM.py:496
            return -(y * logf(s + 1e-7f) + (1.f - y) * logf(1.f - s + 1e-7f));
        }
        default: *out_eval = out_raw; return (y - out_raw) * (y - out_raw);
    }
}

// One workgroup per example.  Everything that does not depend on an earlier phase is loaded first (the
// embedding tile, this thread's slice of the dense(32) kernel, the first-order inputs), so the kernel pays the
// L2/HBM latency once instead of once per phase; the pooling sweep keeps four rows in flight per lane.
#define HEAD_KPP 16      // preloaded dense(32) rows per thread: covers 2D-2 <= 128
static inline size_t head_fwd_lds(const Geo& g) { return (size_t)(1024 + 8 * CFFM_HEAD_UNITS + CFFM_MAX_FIELDS + 4 + g.F * g.D + g.F * g.F) * 4 + 16; }

// NW wavefronts: the pooling sweeps, the embedding tile and s0 use all of them; the dense(32) partials stay on the
// first 256 threads (8 parts x 32 units)
// CL (fused forward): LDS copies of this example's conv outputs (at fused_c_off(l, c0_off, c1_off, ..) inside smem) are read
// instead of the global ones
// POOLS: the stand-alone head of the wide shapes takes s0 and the layer pools as left by earlier kernels (s0_ready, pool[]).  The
// fused forward instantiates POOLS = false: a.pool[l] / a.pool_np[l] with a run-time l would be a dynamically indexed member of a
// by-value kernel argument, i.e. a private copy in scratch memory inside the single-launch forward (it cost fwd_all 1.4 us).
template <int NW = 4, int ACTC = -1, bool POOLS = false>
__device__ __forceinline__ void head_fwd_body(const HeadArgs& a, int b, char* smem, bool CL = false, int c0_off = 0, int c1_off = 0) {
    const int act = ACTC >= 0 ? ACTC : a.g.act;         // ACTC >= 0: compile-time activation id (README shapes)
    constexpr int NTH = 64 * NW, RPW = 16 / NW;                    // rows per wave in one pooling sweep
    float* t1s = reinterpret_cast<float*>(smem);                   // [1024]
    float (*hpart)[CFFM_HEAD_UNITS] = reinterpret_cast<float (*)[CFFM_HEAD_UNITS]>(t1s + 1024);   // [8][32]
    float* rs = t1s + 1024 + 8 * CFFM_HEAD_UNITS;                  // [CFFM_MAX_FIELDS]
    float* sc = rs + CFFM_MAX_FIELDS;                              // [4]
    float* Et = sc + 4;                                            // [F][D] embedding tile of this example
    float* aW = Et + a.g.F * a.g.D;                                // [F][F] attention matrix, staged and used by one wave only
    const Geo& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t1w = 2 * g.D - 2;
    const int q = tid & 31, part = tid >> 5, kpp = (t1w + 7) / 8;
    PHASE_MARK3(0);
    // ---- phase 0: independent loads ------------------------------------------------------------------------
    float w1r[HEAD_KPP];
    const bool pre = a.outer_conv && kpp <= HEAD_KPP;
    if (pre && tid < 256) {
#pragma unroll
        for (int i = 0; i < HEAD_KPP; ++i) {
            const int k = part * kpp + i;
            w1r[i] = (i < kpp && k < t1w) ? a.d1_w[k * CFFM_HEAD_UNITS + q] : 0.f;
        }
    }
    // every scalar / per-lane parameter of the later phases is requested now, so that no phase waits on L2 again
    float fbv = 0.f, attb = 0.f, linw = 0.f, linb = 0.f, d1b = 0.f, d2w = 0.f, d2b = 0.f, io = 0.f, biasv = 0.f, yv = 0.f;
    // the first-order term runs on the LAST wavefront: the first ones carry the pooling rows of the small top layers, and
    // the barrier behind the pools waited 1.9 us for wave 1 when it did both (phase timers, frappe)
    constexpr int FO = NW - 1;
    if (wave == FO) {
        if (lane < g.F) {
            fbv = a.fb[(int64_t)b * g.F + lane];
            linw = g.linear_att ? a.lin_w[lane] : 0.f;
            attb = g.linear_att ? a.att_b[lane] : 0.f;
        }
        if (g.linear_att) {
            linb = a.lin_b[0];
            for (int e = lane; e < g.F * g.F; e += 64) aW[e] = a.att_W[e];
        }
    }
    if (wave == 0) {
        if (a.outer_conv && lane < CFFM_HEAD_UNITS) { d1b = a.d1_b[lane]; d2w = a.d2_w[lane]; }
        if (a.outer_conv) d2b = a.d2_b[0];
        if (tid == 0) {
            io = a.inner_conv ? a.inner_out[b] : 0.f;
            biasv = a.bias[0];
            yv = a.y ? a.y[b] : 0.f;
        }
    }
    if (a.outer_conv && !(POOLS && a.s0_ready)) {
        const float4* E4 = reinterpret_cast<const float4*>(a.Eo + (int64_t)b * g.F * g.D);
        for (int i = tid; i < g.F * g.D / 4; i += NTH) reinterpret_cast<float4*>(Et)[i] = E4[i];
    }
    float s0v = 0.f;                                                 // s0_ready: this thread's element of the pool, requested now
    if (POOLS && a.outer_conv && a.s0_ready && tid < g.D) s0v = a.t1[(int64_t)b * t1w + tid];
    if (wave == FO) {                                                // first-order term, :422-446 (needs nothing from the other waves)
        float lin;
        if (g.linear_att) {
            float z = attb;
            for (int gI = 0; gI < g.F; ++gI) {
                const float fg = __shfl(fbv, gI, 64);
                if (lane < g.F) z += fg * aW[gI * g.F + lane];          // staged by this wavefront above: no barrier needed
            }
            z = lane < g.F ? z / g.lamda_att : -INFINITY;
            const float mx = wave_max(z);
            const float e = lane < g.F ? expf(z - mx) : 0.f;
            const float den = wave_sum(e);
            const float at = e / den;
            if (lane < g.F) a.att[(int64_t)b * g.F + lane] = at;
            lin = wave_sum(lane < g.F ? fbv * at * linw : 0.f) + linb;
        } else {
            lin = wave_sum(fbv);
        }
        if (lane == 0) sc[1] = lin;
    }
    PHASE_MARK3(1);
    float o = 0.f;
    if (a.outer_conv) {
        // pools of the live layers: s_{l+1}[y] = sum_{x,q} act(C_l[b,y,x,q])                      (:390-391)
        int off = g.D;
        for (int l = 0; l < g.live; ++l) {                          // only s_1 .. s_{Lc-1} reach t1 (:394-396)
            const int S = g.D >> (l + 1);
            if (POOLS && !CL && a.pool_np[l] > 0) {                 // the conv epilogue left the partials: add them up in index order
                const int np = a.pool_np[l];
                for (int y = tid; y < S; y += NTH) {
                    const float* pp = a.pool[l] + ((int64_t)b * S + y) * np;
                    float v = 0.f;
                    for (int k = 0; k < np; ++k) v += pp[k];
                    t1s[off + y] = v;
                }
                off += S;
                continue;
            }
            const int n4 = S * g.Pp / 4;
            const float4* base = CL ? reinterpret_cast<const float4*>(smem + fused_c_off(l, c0_off, c1_off, g.D, g.Pp))
                                    : reinterpret_cast<const float4*>(a.C[l] + (int64_t)b * S * S * g.Pp);
            // wave w owns rows w, w+4, ...; four rows are swept together so that four loads are in flight per lane
            for (int y0 = wave; y0 < S; y0 += 16) {
                float s4[RPW];
#pragma unroll
                for (int u = 0; u < RPW; ++u) s4[u] = 0.f;
                for (int i = lane; i < n4; i += 64) {
#pragma unroll
                    for (int u = 0; u < RPW; ++u) {
                        const int yy = y0 + NW * u;
                        if (yy < S) {
                            const float4 v = base[(int64_t)yy * n4 + i];
                            s4[u] += (act_pos(v.x, act) + act_pos(v.y, act)) + (act_pos(v.z, act) + act_pos(v.w, act));
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < RPW; ++u) {
                    const float t = wave_sum(s4[u]);
                    if (lane == 0 && y0 + NW * u < S) t1s[off + y0 + NW * u] = t;
                }
            }
            off += S;
        }
        PHASE_MARK3(2);
        lds_barrier();                                             // Et (and the pools) are in LDS
        PHASE_MARK3(3);
        if (POOLS && a.s0_ready) {                                   // kernel-uniform: every thread takes the same side
            if (tid < g.D) t1s[tid] = s0v;                           // (g.D <= 256 <= NTH on this path: the tiled layer 0 has D <= 64)
        } else {
        for (int f = wave; f < g.F; f += NW) {                      // row sums of the embedding tile
            float s = 0.f;
            for (int d = lane; d < g.D; d += 64) s += Et[f * g.D + d];
            s = wave_sum(s);
            if (lane == 0) rs[f] = s;
        }
        lds_barrier();
        // s0[h] = sum_{w,p} Eo[i_p][h] * Eo[j_p][w] = sum_i Eo[i][h] * sum_{j>i} rowsum(j)   (:381)
        for (int h = tid; h < g.D; h += NTH) {
            float s = 0.f, R = 0.f;
            for (int i = g.F - 2; i >= 0; --i) {
                R += rs[i + 1];
                s += Et[i * g.D + h] * R;
            }
            t1s[h] = s;
        }
        }
        lds_barrier();
        PHASE_MARK3(4);
        for (int k = tid; k < t1w; k += NTH) a.t1[(int64_t)b * t1w + k] = t1s[k];
        if (tid < 256) {                                             // dense(32), :409: 8 partial sums per unit
            float s = 0.f;
            if (pre) {
#pragma unroll
                for (int i = 0; i < HEAD_KPP; ++i) {
                    const int k = part * kpp + i;
                    if (i < kpp && k < t1w) s += t1s[k] * w1r[i];
                }
            } else {
                for (int k = part * kpp; k < min(t1w, (part + 1) * kpp); ++k) s += t1s[k] * a.d1_w[k * CFFM_HEAD_UNITS + q];
            }
            hpart[part][q] = s;
        }
        lds_barrier();
        PHASE_MARK3(5);
        if (wave == 0) {                                             // + bias, then dense(1) * beta, :410, :414
            float h = 0.f;
            if (lane < CFFM_HEAD_UNITS) {
                h = d1b;
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) h += hpart[pp][lane];
                a.h1[(int64_t)b * CFFM_HEAD_UNITS + lane] = h;
            }
            float v = lane < CFFM_HEAD_UNITS ? h * d2w : 0.f;
            v = wave_sum(v);
            if (lane == 0) sc[0] = g.beta_outer * (v + d2b);
        }
    }
    PHASE_MARK3(6);
    lds_barrier();
    if (tid == 0) {
        if (a.outer_conv) o = sc[0];
        float out = io;
        out += o;
        out += sc[1];
        out += biasv;                                                // :449-453
        float ev = out;
        if (a.y) a.sqerr[b] = loss_term(out, yv, a.loss, &ev);
        else if (a.loss == CFFM_LOSS_LOG) ev = 1.f / (1.f + expf(-out));
        a.out[b] = ev;
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

// The backward of the head is written as begin / loss / example / end so that the stand-alone kernel (one workgroup per
// slab looping over its examples) and the fused top-of-backward kernel share it.
struct HeadBwdState {
    float g_d2w, g_d1b, g_d2b, g_bias, g_linw, g_linb, g_attb;
    bool first;
};

__device__ __forceinline__ void head_bwd_begin(const HeadBwdArgs& a, int slab, HeadBwdState& st) {
    const int tid = threadIdx.x;
    float* s_attW = a.s_attW + (int64_t)slab * a.stride_front;
    float* s_d1w = a.s_d1w + (int64_t)slab * a.stride_back;
    // zero this slab's two ranges first (alignment gaps and members this configuration never writes)
    for (int64_t e = tid; e < a.front_len; e += 256) s_attW[e] = 0.f;        // att_W is the first member of the range
    for (int64_t e = tid; e < a.back_len; e += 256) s_d1w[e] = 0.f;          // d1_w is the first member of the range
    st.g_d2w = st.g_d1b = st.g_d2b = st.g_bias = st.g_linw = st.g_linb = st.g_attb = 0.f;
    st.first = true;
    __syncthreads();
}

// The loss L (CFFM.py:486-513) from the per-example terms, summed in the same fixed order by every workgroup; workgroup
// `publisher` also writes it out.  red: LDS [4].
__device__ __forceinline__ float head_bwd_loss(const HeadBwdArgs& a, bool publisher, float* red) {
    const int tid = threadIdx.x;
    float sum = 0.f, hybrid_log = 0.f;
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
    if (publisher && tid == 0) {
        a.scalars[1] = L;
        if (a.sqerr) { a.scalars[0] = sum; a.scalars[3] = sum; }
        if (a.loss_out) a.loss_out[0] = L;
    }
    return L;
}

// dh1s [CFFM_HEAD_UNITS] and dt1s [2D-2] are LDS scratch of the caller
template <int ACTC = -1>
__device__ __forceinline__ void head_bwd_example(const HeadBwdArgs& a, int slab, HeadBwdState& st, int b, float d,
                                                 float* dh1s, float* dt1s) {
    const Geo& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t1w = 2 * g.D - 2;
    float* s_attW = a.s_attW + (int64_t)slab * a.stride_front;
    float* s_d1w = a.s_d1w + (int64_t)slab * a.stride_back;
    const bool first = st.first;
    int off_top = 0;
    for (int i = 0; i < g.live; ++i) off_top += g.D >> i;
    if (tid == 0) { a.dout[b] = d; st.g_bias += d; }
    __syncthreads();
    if (a.outer_conv) {
        const float dd = d * g.beta_outer;
        if (tid < CFFM_HEAD_UNITS) {
            const float v = dd * a.d2_w[tid];
            dh1s[tid] = v;
            st.g_d2w += a.h1[(int64_t)b * CFFM_HEAD_UNITS + tid] * dd;
            st.g_d1b += v;
        }
        if (tid == 0) st.g_d2b += dd;
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
            a.dCtop[idx] = dt1s[off_top + yy] * act_relu_grad(a.Ctop[idx], ACTC >= 0 ? ACTC : g.act);
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
            st.g_linw += fbv * at * d;
            st.g_attb += dz;
            if (lane == 0) st.g_linb += d;
        } else if (lane < g.F) {
            a.dfb[(int64_t)b * g.F + lane] = d;
        }
    }
    st.first = false;
}

__device__ __forceinline__ void head_bwd_end(const HeadBwdArgs& a, int slab, const HeadBwdState& st) {
    const int tid = threadIdx.x;
    const int64_t sof = (int64_t)slab * a.stride_front, sob = (int64_t)slab * a.stride_back;
    if (tid < CFFM_HEAD_UNITS) {
        a.s_d2w[sob + tid] = st.g_d2w;
        a.s_d1b[sob + tid] = st.g_d1b;
    }
    if (tid < a.g.F) {
        a.s_linw[sob + tid] = st.g_linw;
        a.s_attb[sof + tid] = st.g_attb;
    }
    if (tid == 0) {
        a.s_d2b[sob] = st.g_d2b;
        a.s_bias[sof] = st.g_bias;
        a.s_linb[sob] = st.g_linb;
    }
}

static inline void fill_head_bwd_args(const cffm_shape_t* s, const float* theta, void* ws, const float* y, int32_t B,
                                      int64_t B_global, bool local_sum, float* loss_out, bool unscaled, HeadBwdArgs* out) {
    cffm_theta_layout_t tl; cffm_ws_layout_t wl;
    cffm_theta_layout(s, &tl); cffm_ws_layout(s, B, &wl);
    char* w = (char*)ws;
    SlabPlan sp;
    make_slab_plan(s, B, tl, &sp);
    const SlabRange& rf = sp.r[sp.head_front];
    const SlabRange& rb = sp.r[sp.head_back];
    float* gf = (float*)(w + wl.gpart) + rf.base - rf.off;        // slab 0 of theta offset x lives at gf + x
    float* gb = (float*)(w + wl.gpart) + rb.base - rb.off;
    HeadBwdArgs& a = *out;
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
}
