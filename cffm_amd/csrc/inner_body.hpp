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
static inline size_t inner_fwd_lds(const Geo& g) { return (size_t)(g.F * g.K + g.Pp + 8) * 4; }

__device__ __forceinline__ void inner_fwd_body(const InnerFwdArgs& ia, int b, char* smem) {
    const Geo& g = ia.g;
    const float* Ei = ia.Ei; const float* cw_g = ia.cw; const float* cb_g = ia.cb; const float* wd = ia.wd;
    const float* bd = ia.bd; float* inner_out = ia.inner_out; const FusedGather& fg = ia.fg;
    float* E = reinterpret_cast<float*>(smem);                                   // [F*K]
    uint32_t* lut = reinterpret_cast<uint32_t*>(E + g.F * g.K);                 // [Pp]
    float* red = reinterpret_cast<float*>(lut + g.Pp);                          // [4]
    const int FK4 = g.F * g.K / 4;
    if (fg.ids != nullptr) {
        const int K4 = g.K / 4, D4 = fg.D / 4;
        const int32_t* idb = fg.ids + (int64_t)b * g.F;
        for (int i = threadIdx.x; i < g.F * (K4 + D4); i += blockDim.x) {
            const bool in = i < g.F * K4;
            const int j = in ? i : i - g.F * K4, per = in ? K4 : D4, f = j / per, c = j - f * per;
            int id = idb[f];
            id = id < 0 ? 0 : (id >= fg.M ? fg.M - 1 : id);
            const float4 v = reinterpret_cast<const float4*>(in ? fg.inner : fg.outer)[(int64_t)id * per + c];
            if (in) {
                reinterpret_cast<float4*>(E)[j] = v;
                reinterpret_cast<float4*>(fg.Ei + (int64_t)b * g.F * g.K)[j] = v;
            } else {
                reinterpret_cast<float4*>(fg.Eo + (int64_t)b * g.F * fg.D)[j] = v;
            }
        }
        if (threadIdx.x < g.F) {
            const int raw = idb[threadIdx.x];
            const int id = raw < 0 ? 0 : (raw >= fg.M ? fg.M - 1 : raw);
            const int64_t slot = (int64_t)b * g.F + threadIdx.x;
            fg.fb[slot] = fg.fbias[id];
            fg.keys[slot] = ((unsigned long long)(unsigned)raw << 32) | (unsigned long long)slot;
        }
    } else {
        const float4* src = reinterpret_cast<const float4*>(Ei + (int64_t)b * g.F * g.K);
        for (int i = threadIdx.x; i < FK4; i += blockDim.x) reinterpret_cast<float4*>(E)[i] = src[i];
    }
    build_pair_lut(lut, g.F, g.Pp);
    float cw[4] = {cw_g[0], cw_g[1], cw_g[2], cw_g[3]};
    float cb[2] = {cb_g[0], cb_g[1]};
    __syncthreads();
    const int K2 = g.K / 2, units = g.P * K2;
    const float invK2 = 1.f / (float)K2;
    float part = 0.f;
    for (int u = threadIdx.x; u < units; u += blockDim.x) {
        const int p = fast_div(u, invK2), t = u - p * K2;
        const InnerUnit v = inner_unit(E, lut, p, t, g.K, cw, cb, g.act);
        const float2 w2 = *reinterpret_cast<const float2*>(&wd[(int64_t)p * g.K + 2 * t]);   // flat index p*K + t*2 + ch (:333)
        part += v.s0 * w2.x + v.s1 * w2.y;
    }
    const float tot = block_sum(part, red);
    if (threadIdx.x == 0) inner_out[b] = tot + bd[0];                            // :339
}

