// Device body of the inner-branch forward (shared by inner_fwd_kernel and the fused forward kernel).
#pragma once
#include "common.hpp"

struct InnerUnit {
    float x0, x1, I0, I1, z0, z1, s0, s1, eix, eiy, ejx, ejy;
    int i, j;
};

__device__ __forceinline__ InnerUnit inner_unit(const float* E, const uint32_t* lut, int p, int t, int K,
                                                const float* cw, const float* cb, int act) {
    InnerUnit u;
    const uint32_t ij = lut[p];
    u.i = ij & 0xffff; u.j = ij >> 16;
    const float2 ei = *reinterpret_cast<const float2*>(&E[u.i * K + 2 * t]);
    const float2 ej = *reinterpret_cast<const float2*>(&E[u.j * K + 2 * t]);
    u.eix = ei.x; u.eiy = ei.y; u.ejx = ej.x; u.ejy = ej.y;
    u.I0 = ei.x * ej.x; u.I1 = ei.y * ej.y;                       // CFFM.py:310
    u.x0 = act_f(u.I0, act); u.x1 = act_f(u.I1, act);             // :319
    u.z0 = u.x0 * cw[0] + u.x1 * cw[2] + cb[0];                   // :327  cw[tap*2+ch]
    u.z1 = u.x0 * cw[1] + u.x1 * cw[3] + cb[1];
    const float mp = fmaxf(u.x0, u.x1);                           // :331
    u.s0 = act_pos(fmaxf(u.z0, 0.f), act) + mp;                   // :478, :330, :332
    u.s1 = act_pos(fmaxf(u.z1, 0.f), act) + mp;
    return u;
}

// Fused-step variant (fg.ids != NULL): this kernel IS the embedding gather.  The workgroup of example b
// fetches its F rows of all three tables straight from HBM (16-byte pieces, row pieces of one slot on
// consecutive lanes), keeps the inner rows in LDS for its own use, and leaves Ei/Eo/fb (and the packed sort keys
// of the sparse update) in the workspace for the later stages - one launch and one pass less than a separate gather.
struct FusedGather {
    const int32_t* ids;                    // NULL: the rows were gathered by cffm_gather already
    const float *inner, *outer, *fbias;
    float *Ei, *Eo, *fb;
    unsigned long long* keys;
    int M, D;
};

struct InnerFwdArgs {
    Geo g;
    const float *Ei, *cw, *cb, *wd, *bd;
    float* inner_out;
    FusedGather fg;
};
static inline size_t inner_fwd_lds(const Geo& g) { return (size_t)(g.F * g.K + g.Pp + 16) * 4; }   // + up to 16 wavefront partials

// ACTC >= 0: the activation id as a compile-time constant (the README shapes): every act switch folds away
// EoL != nullptr (fused forward): the gather also leaves the outer rows of the example in LDS, [F][EoLp] floats, where the
// factorised layer 0 reads them - that phase then starts without a trip to global memory.
// mid(): called once by every thread between the first batch of gather loads and their use - the fused forward parks the
// layer-0 filter in LDS there (its loads were issued before the gather's and return first).
struct NoMid { __device__ __forceinline__ void operator()() const {} };
template <int ACTC = -1, class Mid = NoMid>
__device__ __forceinline__ void inner_fwd_body(const InnerFwdArgs& ia, int b, char* smem, float* EoL = nullptr, int EoLp = 0,
                                               Mid mid = Mid()) {
    const Geo& g = ia.g;
    const int act = ACTC >= 0 ? ACTC : g.act;
    const float* Ei = ia.Ei; const float* cw_g = ia.cw; const float* cb_g = ia.cb; const float* wd = ia.wd;
    const float* bd = ia.bd; float* inner_out = ia.inner_out; const FusedGather& fg = ia.fg;
    float* E = reinterpret_cast<float*>(smem);                                   // [F*K]
    uint32_t* lut = reinterpret_cast<uint32_t*>(E + g.F * g.K);                 // [Pp]
    float* red = reinterpret_cast<float*>(lut + g.Pp);                          // [4]
    const int FK4 = g.F * g.K / 4;
    if (fg.ids != nullptr) {
        const int K4 = g.K / 4, D4 = fg.D / 4;
        const int32_t* idb = fg.ids + (int64_t)b * g.F;
        const int n_piece = g.F * (K4 + D4);
        float4 v_first = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((int)threadIdx.x < n_piece) {                  // the first piece of every thread is in flight across mid()
            const int i = threadIdx.x;
            const bool in = i < g.F * K4;
            const int j = in ? i : i - g.F * K4, per = in ? K4 : D4, f = j / per, c = j - f * per;
            int id = idb[f];
            id = id < 0 ? 0 : (id >= fg.M ? fg.M - 1 : id);
            v_first = reinterpret_cast<const float4*>(in ? fg.inner : fg.outer)[(int64_t)id * per + c];
        }
        mid();
        for (int i = threadIdx.x; i < n_piece; i += blockDim.x) {
            const bool in = i < g.F * K4;
            const int j = in ? i : i - g.F * K4, per = in ? K4 : D4, f = j / per, c = j - f * per;
            float4 v = v_first;
            if (i != (int)threadIdx.x) {
                int id = idb[f];
                id = id < 0 ? 0 : (id >= fg.M ? fg.M - 1 : id);
                v = reinterpret_cast<const float4*>(in ? fg.inner : fg.outer)[(int64_t)id * per + c];
            }
            if (in) {
                reinterpret_cast<float4*>(E)[j] = v;
                reinterpret_cast<float4*>(fg.Ei + (int64_t)b * g.F * g.K)[j] = v;
            } else {
                reinterpret_cast<float4*>(fg.Eo + (int64_t)b * g.F * fg.D)[j] = v;
                if (EoL != nullptr) {
                    float* e = EoL + f * EoLp + 4 * c;
                    e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w;
                }
            }
        }
        if (threadIdx.x < g.F) {
            const int raw = idb[threadIdx.x];
            const int id = raw < 0 ? 0 : (raw >= fg.M ? fg.M - 1 : raw);
            const int64_t slot = (int64_t)b * g.F + threadIdx.x;
            fg.fb[slot] = fg.fbias[id];
            fg.keys[slot] = ((unsigned long long)(unsigned)((raw < 0 || raw >= fg.M) ? fg.M : raw) << 32) | (unsigned long long)slot;   // bad id -> key M
        }
    } else {
        mid();
        const float4* src = reinterpret_cast<const float4*>(Ei + (int64_t)b * g.F * g.K);
        for (int i = threadIdx.x; i < FK4; i += blockDim.x) reinterpret_cast<float4*>(E)[i] = src[i];
    }
    build_pair_lut(lut, g.F, g.Pp);
    float cw[4] = {cw_g[0], cw_g[1], cw_g[2], cw_g[3]};
    float cb[2] = {cb_g[0], cb_g[1]};
    const int K2 = g.K / 2, units = g.P * K2;
    // dense(1) weights of unit u = p*K2 + t: flat index p*K + t*2 + ch (:333) = 2u + ch; the first pair of every thread is
    // fetched before the barrier, behind the gather
    const float2* wd2 = reinterpret_cast<const float2*>(wd);
    const float2 w2_first = (int)threadIdx.x < units ? wd2[threadIdx.x] : make_float2(0.f, 0.f);
    __syncthreads();
    const float invK2 = 1.f / (float)K2;
    float part = 0.f;
    for (int u = threadIdx.x; u < units; u += blockDim.x) {
        const int p = fast_div(u, invK2), t = u - p * K2;
        const InnerUnit v = inner_unit(E, lut, p, t, g.K, cw, cb, act);
        const float2 w2 = u == (int)threadIdx.x ? w2_first : wd2[u];
        part += v.s0 * w2.x + v.s1 * w2.y;
    }
    const float tot = block_sum(part, red);
    if (threadIdx.x == 0) inner_out[b] = tot + bd[0];                            // :339
}

// One workgroup per gradient slab; it walks examples slab, slab + NSLAB, ...  Thread t always owns the
// same (p, t) units, so the dense-kernel gradient is accumulated by plain read-modify-write in the
// workgroup's own slab (no atomics, fixed order).  dEi of one example is accumulated in per-wavefront
// private LDS copies (ds_add_f32) that are merged in wavefront order: bitwise reproducible.
struct InnerBwdArgs {
    Geo g;
    int B;
    const float *Ei, *dout, *cw, *cb, *wd;
    const float *out, *y;                             // dout == NULL: dL/dout = head_dout(loss, out[b], y[b], invB, L) on the fly
    int loss;
    float invB;
    float* dEi;
    float *slab_cw, *slab_cb, *slab_dw, *slab_db;     // slab 0
    int64_t slab_stride;
    const int32_t* idx = nullptr;                     // non-NULL: Ei is the inner TABLE [M][K] and row (b, f) is idx[b*F+f] (RowSrc)
    int idxM = 0;
    int idxStride = 0;                                // floats between rows of that table (0 = K): see RowSrc
};
static inline size_t inner_bwd_lds(const Geo& g) { return (size_t)(5 * g.F * g.K + g.Pp + 8) * 4; }

template <int ACTC = -1>
__device__ __forceinline__ void inner_bwd_body(const InnerBwdArgs& a, int slab, int nslab, char* smem, float L = 1.f) {
    const Geo& g = a.g;
    const int act = ACTC >= 0 ? ACTC : g.act;            // ACTC >= 0: compile-time activation id (README shapes)
    const int B = a.B;
    const float* __restrict__ Ei = a.Ei; const float* __restrict__ dout = a.dout;
    const float* __restrict__ cw_g = a.cw; const float* __restrict__ cb_g = a.cb; const float* __restrict__ wd = a.wd;
    float* __restrict__ dEi = a.dEi;
    float* __restrict__ slab_cw = a.slab_cw + slab * a.slab_stride; float* __restrict__ slab_cb = a.slab_cb + slab * a.slab_stride;
    float* __restrict__ slab_dw = a.slab_dw + slab * a.slab_stride; float* __restrict__ slab_db = a.slab_db + slab * a.slab_stride;
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
    // dense(1) weights of unit u = p*K2 + t sit at flat index p*K + 2t = 2u (:333) and do not depend on the example: the
    // first UPRE units of every thread are fetched once, before any barrier
    constexpr int UPRE = 4;
    const float2* wd2 = reinterpret_cast<const float2*>(wd);
    float2 w2p[UPRE];
#pragma unroll
    for (int k = 0; k < UPRE; ++k) {
        const int u = threadIdx.x + k * blockDim.x;
        w2p[k] = u < units ? wd2[u] : make_float2(0.f, 0.f);
    }
    bool first = true;
    for (int b = slab; b < B; b += nslab) {
        // dL/dout of the example: its loads are in flight together with the embedding rows
        const float db = dout ? dout[b] : head_dout(a.loss, a.out[b], a.y[b], a.invB, L);
        __syncthreads();
        if (a.idx == nullptr) {
            const float4* src = reinterpret_cast<const float4*>(Ei + (int64_t)b * FK);
            for (int i = threadIdx.x; i < FK / 4; i += blockDim.x) reinterpret_cast<float4*>(E)[i] = src[i];
        } else {                                       // wide shapes: the rows were never materialised, fetch them from the table
            const int K4 = g.K >> 2;
            const float invK4 = 1.f / (float)K4;
            for (int i = threadIdx.x; i < FK / 4; i += blockDim.x) {
                const int f = fast_div(i, invK4), c = i - f * K4;
                reinterpret_cast<float4*>(E)[i] =
                    reinterpret_cast<const float4*>(row_ptr(Ei, a.idx, a.idxM, (int64_t)b * g.F + f, g.K, a.idxStride))[c];
            }
        }
        for (int i = threadIdx.x; i < 4 * FK; i += blockDim.x) dE[i] = 0.f;
        __syncthreads();
        float* myE = dE + wave * FK;
        auto unit_step = [&](int u, float2 w2) {
            const int p = fast_div(u, invK2), t = u - p * K2;
            const InnerUnit v = inner_unit(E, lut, p, t, g.K, cw, cb, act);
            const int64_t wi = (int64_t)p * g.K + 2 * t;
            // dense(1) kernel gradient: flat * dout
            float2 acc2 = first ? make_float2(0.f, 0.f) : *reinterpret_cast<float2*>(&slab_dw[wi]);
            acc2.x += v.s0 * db; acc2.y += v.s1 * db;
            *reinterpret_cast<float2*>(&slab_dw[wi]) = acc2;
            const float ds0 = db * w2.x, ds1 = db * w2.y;
            const float dz0 = ds0 * act_relu_grad(fmaxf(v.z0, 0.f), act);
            const float dz1 = ds1 * act_relu_grad(fmaxf(v.z1, 0.f), act);
            gcw[0] += dz0 * v.x0; gcw[1] += dz1 * v.x0; gcw[2] += dz0 * v.x1; gcw[3] += dz1 * v.x1;
            gcb[0] += dz0; gcb[1] += dz1;
            const float dmp = ds0 + ds1;
            const bool firstmax = v.x0 >= v.x1;                       // max-pool grad: first element on ties
            const float dx0 = dz0 * cw[0] + dz1 * cw[1] + (firstmax ? dmp : 0.f);
            const float dx1 = dz0 * cw[2] + dz1 * cw[3] + (firstmax ? 0.f : dmp);
            const float dI0 = dx0 * act_grad_f(v.I0, act), dI1 = dx1 * act_grad_f(v.I1, act);
            atomicAdd(&myE[v.i * g.K + 2 * t], dI0 * v.ejx);
            atomicAdd(&myE[v.i * g.K + 2 * t + 1], dI1 * v.ejy);
            atomicAdd(&myE[v.j * g.K + 2 * t], dI0 * v.eix);
            atomicAdd(&myE[v.j * g.K + 2 * t + 1], dI1 * v.eiy);
        };
#pragma unroll
        for (int k = 0; k < UPRE; ++k) {
            const int u = threadIdx.x + k * blockDim.x;
            if (u < units) unit_step(u, w2p[k]);
        }
        for (int u = threadIdx.x + UPRE * blockDim.x; u < units; u += blockDim.x) unit_step(u, wd2[u]);
        if (threadIdx.x == 0) gdb += db;
        __syncthreads();
        for (int i = threadIdx.x; i < FK; i += blockDim.x)
            dEi[(int64_t)b * FK + i] = ((dE[i] + dE[FK + i]) + dE[2 * FK + i]) + dE[3 * FK + i];
        first = false;
    }
    if (first) {   // this slab saw no example: it must still read as zeros
        for (int64_t i = threadIdx.x; i < (int64_t)g.P * g.K; i += blockDim.x) slab_dw[i] = 0.f;
    }
    // the seven scalar gradients in ONE block reduction (wavefront partials through the idle dE planes, added in
    // wavefront order): two barriers instead of fourteen
    float r[7] = {gcw[0], gcw[1], gcw[2], gcw[3], gcb[0], gcb[1], gdb};
#pragma unroll
    for (int q = 0; q < 7; ++q) r[q] = wave_sum(r[q]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 7; ++q) dE[wave * 8 + q] = r[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            float t = 0.f;
            for (int w = 0; w < nw; ++w) t += dE[w * 8 + q];
            r[q] = t;
        }
        for (int q = 0; q < 4; ++q) slab_cw[q] = r[q];
        slab_cb[0] = r[4]; slab_cb[1] = r[5];
        slab_db[0] = r[6];
    }
}

// returns the slab count of the inner range
static inline int fill_inner_bwd_args(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, InnerBwdArgs* out) {
    cffm_theta_layout_t tl; cffm_ws_layout_t wl;
    cffm_theta_layout(s, &tl); cffm_ws_layout(s, B, &wl);
    char* w = (char*)ws;
    SlabPlan sp;
    make_slab_plan(s, B, tl, &sp);
    const SlabRange& sr = sp.r[sp.inner];
    float* base = (float*)(w + wl.gpart) + sr.base - sr.off;      // slab 0 of theta offset x lives at base + x
    InnerBwdArgs& a = *out;
    a.g = make_geo(s); a.B = B;
    a.Ei = (const float*)(w + wl.Ei); a.dout = (const float*)(w + wl.dout);
    a.out = (const float*)(w + wl.out); a.y = nullptr; a.loss = s->loss; a.invB = 1.f / (float)B;
    a.cw = theta + tl.inner_cw; a.cb = theta + tl.inner_cb; a.wd = theta + tl.inner_dw;
    a.dEi = (float*)(w + wl.dEi); a.idx = nullptr; a.idxM = 0; a.idxStride = 0;
    a.slab_cw = base + tl.inner_cw; a.slab_cb = base + tl.inner_cb; a.slab_dw = base + tl.inner_dw; a.slab_db = base + tl.inner_db;
    a.slab_stride = sr.len;
    return sr.nslab;
}
