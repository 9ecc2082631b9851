// Shared device helpers for the CFFM gfx950 kernels (wave64, fp32 MFMA, LDS tiles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cffm_hip.h"

#define CFFM_WAVE 64

#ifdef CFFM_PHASE_TIMERS
// debug build only (make PHASE_TIMERS=1, tools/phase.py): 100 MHz timestamps of workgroup 7's phase boundaries inside the
// fused forward launch (slots 0-7), and of the sub-phases of its head phase (slots 8-15)
static __device__ unsigned long long cffm_phase_times[16];
static __device__ unsigned long long cffm_wg_times[2 * 1024];
static __device__ unsigned long long cffm_bwd_times[48];     // backward launches: workgroup `wg` of a role, see tools/phase.py
#define PHASE_MARKB(i, wg) do { if ((wg) == 7 && threadIdx.x == 0) cffm_bwd_times[i] = wall_clock64(); } while (0)
#define PHASE_MARK(i) do { if (blockIdx.x == 7 && threadIdx.x == 0) cffm_phase_times[i] = wall_clock64(); } while (0)
#define PHASE_MARK2(i) do {} while (0)
#define PHASE_MARK3(i) do { if (CL && blockIdx.x == 7 && threadIdx.x == 0) cffm_phase_times[8 + (i)] = wall_clock64(); } while (0)
#else
#define PHASE_MARK(i) do {} while (0)
#define PHASE_MARK2(i) do {} while (0)
#define PHASE_MARK3(i) do {} while (0)
#define PHASE_MARKB(i, wg) do {} while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
// LDS byte offset of the copy of C_l that the fused forward keeps for its own later phases: C_0 at c0, C_1, C_2, ... packed
// from c1 on ([S_l*S_l][Pp] floats each, S_l = D >> (l+1)).  Computed, not looked up: an array of offsets indexed by the
// runtime layer number lives in scratch memory (a global-memory round trip per phase).
__device__ __forceinline__ int fused_c_off(int l, int c0, int c1, int D, int Pp) {
    if (l == 0) return c0;
    int o = c1;
    for (int k = 1; k < l; ++k) { const int S = D >> (k + 1); o += S * S * Pp * 4; }
    return o;
}


#define CFFM_CHECK_LAUNCH()                          \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

static inline int ceil_div_i(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int ilog2_i(int v) { int r = 0; while ((1 << (r + 1)) <= v) ++r; return r; }

// Geometry derived from the shape; passed to kernels by value.
struct Geo {
    int M, F, K, D, P, Pp, Lc, live, act, linear_att;
    float lamda_att, beta_outer;
};

static inline Geo make_geo(const cffm_shape_t* s) {
    Geo g;
    g.M = s->M; g.F = s->F; g.K = s->K; g.D = s->D;
    g.P = s->F * (s->F - 1) / 2;
    g.Pp = (g.P + 15) / 16 * 16;
    g.Lc = ilog2_i(s->D);
    g.live = g.Lc - 1;
    g.act = s->act; g.linear_att = s->linear_att;
    g.lamda_att = s->lamda_att; g.beta_outer = s->beta_outer;
    return g;
}

static inline int check_shape(const cffm_shape_t* s) {
    if (!s) return CFFM_ERR_BAD_SHAPE;
    if (s->F < 2 || s->F > CFFM_MAX_FIELDS) return CFFM_ERR_BAD_SHAPE;
    if (s->D < 4 || (s->D & (s->D - 1))) return CFFM_ERR_BAD_SHAPE;
    if (s->K < 4 || (s->K & 3)) return CFFM_ERR_BAD_SHAPE;          // float4 rows
    if (s->M < 1) return CFFM_ERR_BAD_SHAPE;
    if (ilog2_i(s->D) - 1 > CFFM_MAX_LAYERS) return CFFM_ERR_BAD_SHAPE;
    if (s->act < 0 || s->act > CFFM_ACT_GELU) return CFFM_ERR_BAD_SHAPE;
    if (s->loss < 0 || s->loss > CFFM_LOSS_HYBRID) return CFFM_ERR_BAD_SHAPE;
    if (s->optimizer < 0 || s->optimizer > CFFM_OPT_ADAM) return CFFM_ERR_BAD_SHAPE;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Wide filters (Pp > 64): which kernel runs layer 0, how the conv GEMMs cut their columns, and the partial sum pools the conv
// epilogues leave for the head (shared by the workspace layout in api.hip and the kernels in conv.hip / head.hip).
// ---------------------------------------------------------------------------------------------
#define C0T_MAXKS 16      // k-steps of one step-1 unit of the tiled layer 0: 2(F-1)/4 <= 16 for F <= 32
static inline bool conv0_fact_tile_ok(const Geo& g) {
    const int S = g.D / 2;
    return g.Pp > 64 && S >= 16 && S % 16 == 0 && 2 * (g.F - 1) <= 4 * C0T_MAXKS;
}
// the tiled layer-0 input gradient that reads the filter as pre-packed MFMA fragments (ws.w0pack)
static inline bool conv0_tile_dgrad2_ok(const Geo& g) { return conv0_fact_tile_ok(g) && g.D / 2 <= 32 && g.F <= 32; }
static inline bool conv0_tile_fwd_ok(const Geo& g) { return conv0_fact_tile_ok(g) && 2 * ((g.F + 3) & ~3) <= 4 * C0T_MAXKS; }
// ws.relu0: relu masks of C_0 .. C_{live-2} of the wide shapes, one after the other ([rows of C_l][Pp/16] 16-bit words each)
static inline int64_t relu_mask_bytes(const Geo& g, int64_t B, int l) {
    const int64_t S = g.D >> (l + 1);
    return (B * S * S * (g.Pp / 16) * 2 + 255) / 256 * 256;
}
static inline int64_t relu_mask_off(const Geo& g, int64_t B, int l) {
    int64_t o = 0;
    for (int k = 0; k < l; ++k) o += relu_mask_bytes(g, B, k);
    return o;
}
// column tiles of 16 -> column blocks of NT tiles (NT in {1,2,3,4,6,8}) of the implicit-GEMM kernels
static inline void pick_nt(int tiles, int* nblk, int* NT) {
    int nb = (tiles + 7) / 8;
    int need = (tiles + nb - 1) / nb;
    static const int allowed[6] = {1, 2, 3, 4, 6, 8};
    int nt = 8;
    for (int i = 0; i < 6; ++i)
        if (allowed[i] >= need) { nt = allowed[i]; break; }
    *NT = nt;
    *nblk = (tiles + nt - 1) / nt;
}
// Sum pools of the wide shapes (CFFM.py:390-391): s_{l+1}[b][y] = sum_{x,q} act(C_l[b][y][x][q]).  Round 2's head swept the conv
// outputs again for them (21.7 GB at the stress shape: 3.7 ms, HBM-bound on values the conv epilogues just had in registers).
// Now every epilogue leaves, per (example, row y), ONE partial per column block (layers >= 1) or per (column tile, channel tile)
// (tiled layer 0), in a fixed place; the head adds the pool_partials(g, l) partials of a row in index order (bitwise reproducible).
// 0: this shape keeps the sweep.
static inline int pool_partials(const Geo& g, int l) {
    if (g.Pp <= 64 || g.D / 2 > 64) return 0;              // rows of one y must fit the smallest row tile (64 rows)
    if (l == 0 && conv0_tile_fwd_ok(g)) return (g.D / 2 / 16) * (g.Pp / 16);
    int nblk, NT;
    pick_nt(g.Pp / 16, &nblk, &NT);
    return nblk;
}

// ---------------------------------------------------------------------------------------------
// Where the looked-up rows of a [B,F] id batch are read from (tf.nn.embedding_lookup, CFFM.py:303, :354, :422).
//   idx == nullptr  the copy a gather left in the workspace: row `slot` (= b*F + f) of base [B*F][dim]
//   idx != nullptr  straight from the table: row clamp(idx[slot]) of base [M][dim].  The wide shapes (Pp > 64) run this
//                   way: the fused gather + inner-branch forward consumes the rows in the kernel that fetches them and
//                   nothing is materialised; the few later kernels that need a row again (the tiled layer 0, the
//                   inner-branch backward) fetch it from the table - 67 MB of reads against 27-64 ms kernels.
// ---------------------------------------------------------------------------------------------
struct RowSrc {
    const float* base;
    const int32_t* idx;
    int M;
    int stride = 0;      // floats between consecutive rows of base; 0 = the row length.  The row-sharded step reads its rows out of the
                         // packed records it received ([n_records][K+D+4]: base = records (+ K for the outer row), idx = slot -> record)
};
__device__ __forceinline__ const float* row_ptr(const float* base, const int32_t* idx, int M, int64_t slot, int dim, int stride = 0) {
    if (idx == nullptr) return base + slot * dim;
    int id = idx[slot];
    id = id < 0 ? 0 : (id >= M ? M - 1 : id);               // clamp: a bad id must not fault the GPU
    return base + (int64_t)id * (stride > 0 ? stride : dim);
}
// [F][D] rows of example b -> LDS tile Es[f * Dp + d] (the staging loop of the tiled layer-0 kernels)
__device__ __forceinline__ void stage_example_rows(float* Es, const float* base, const int32_t* idx, int M, int b, int F, int D,
                                                   int Dp, int tid, int nth, int stride = 0) {
    const float invD = 1.f / (float)D;
    if (idx == nullptr) {
        const float* e = base + (int64_t)b * F * D;
        for (int i = tid; i < F * D; i += nth) {
            const int f = (int)(((float)i + 0.5f) * invD), d = i - f * D;
            Es[f * Dp + d] = e[i];
        }
    } else {
        for (int i = tid; i < F * D; i += nth) {
            const int f = (int)(((float)i + 0.5f) * invD), d = i - f * D;
            Es[f * Dp + d] = row_ptr(base, idx, M, (int64_t)b * F + f, D, stride)[d];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Split-K gradient slabs.  Every dense-gradient producer writes per-workgroup partial sums ("slabs") that
// cffm_reduce_slabs adds up in slab order.  theta is cut into contiguous ranges, each with its own slab
// count: the small head / inner ranges get one slab per example (up to 256), a conv layer gets CFFM_NSLAB or,
// when it has many rows, 256.  Range r keeps its slabs at gpart + base: [nslab][len] floats.
// ---------------------------------------------------------------------------------------------
#define CFFM_NSLAB_SMALL 256
// slabs of the small ranges (head, inner branch): one per example between 256 and 1024, so that the kernels that walk
// them (head_bwd, inner_bwd) get one example per workgroup up to B = 1024
static inline int small_slabs(int32_t B) { return B <= 256 ? 256 : (B >= 1024 ? 1024 : (B + 255) / 256 * 256); }
struct SlabRange { int64_t off, len, base; int nslab; };
struct SlabPlan {
    SlabRange r[CFFM_MAX_LAYERS + 3];
    int n;
    int64_t total;                       // floats
    int head_front, inner, conv0, head_back;   // indices into r
};

// Fused top of the backward (bwd_top_kernel): head + the top two conv layers, one workgroup (= one slab) per example
static inline bool bwd_top_ok(const cffm_shape_t* s, int32_t B) {
    const int F = s->F, Pp = (F * (F - 1) / 2 + 15) / 16 * 16;
    const int live = ilog2_i(s->D) - 1;
    return Pp <= 64 && B <= 256 && s->inner_conv && s->outer_conv && live >= 2;
}
static inline int bwd_top_first_layer(const cffm_shape_t* s) {
    const int live = ilog2_i(s->D) - 1;
    return live - 2 >= 1 ? live - 2 : 1;
}
// Slabs of a conv layer below the fused top: 256 when the layer has at least 64 rows per slab that way, else CFFM_NSLAB
static inline int conv_slabs_plain(const cffm_shape_t* s, int32_t B, int l) {
    const int F = s->F, Pp = (F * (F - 1) / 2 + 15) / 16 * 16;
    const int64_t S = s->D >> (l + 1);
    return (Pp <= 64 && (int64_t)B * S * S >= 256 * 64) ? 256 : CFFM_NSLAB;
}
// The layer right below the fused top runs its input and weight gradient as two roles of one launch
// (conv_bwd_pair_kernel) when its slabs are shorter than 128 rows.
static inline bool conv_pair_ok(const cffm_shape_t* s, int32_t B, int l) {
    const int F = s->F, Pp = (F * (F - 1) / 2 + 15) / 16 * 16;
    const int64_t S = s->D >> (l + 1), rows = (int64_t)B * S * S;
    const int nslab = conv_slabs_plain(s, B, l);
    return l >= 1 && Pp <= 64 && (rows + nslab - 1) / nslab < 128;
}
// The weight gradients of the fused top's conv layers have few rows (B*4 and B*16 at D = 32): with one slab per example
// they cost 2 x 256 slabs of 4*Pp*Pp floats (19 MB written and read again at frappe) for 5120 rows of work.  When a pair
// launch follows the fused top, they run there instead, as two more roles over 64-row slabs (16 + 64 slabs at frappe).
// D = 32 (four conv layers) with Pp <= 48 and 64 <= B <= 256, i.e. the frappe command: everything below the fused top - the
// weight gradients of layers 3, 2, 1, the input gradient of layer 1 and the factorised backward of layer 0 - runs in ONE
// launch with one 16-wavefront workgroup per example (conv01_bwd_kernel); every conv layer then has one slab per example.
#define CFFM_TOP_SLAB_ROWS 32      // rows per slab of layers 2 and 3 in conv01_bwd_kernel (half of what the layer-1 group of a workgroup handles)
static inline bool bwd_fused01_ok(const cffm_shape_t* s, int32_t B) {
    const int F = s->F, Pp = (F * (F - 1) / 2 + 15) / 16 * 16;
    return bwd_top_ok(s, B) && s->D == 32 && Pp <= 48 && B >= 64;
}
static inline bool top_wgrad_deferred(const cffm_shape_t* s, int32_t B) {
    return bwd_top_ok(s, B) && !bwd_fused01_ok(s, B) && conv_pair_ok(s, B, bwd_top_first_layer(s) - 1);
}
static inline int conv_slabs(const cffm_shape_t* s, int32_t B, int l) {
    const int64_t S = s->D >> (l + 1);
    if (bwd_fused01_ok(s, B)) {                  // layers 0, 1: one slab per example; layers 2, 3: slabs of CFFM_TOP_SLAB_ROWS rows
        if (l <= 1) return 256;
        const int64_t n = ((int64_t)B * S * S + CFFM_TOP_SLAB_ROWS - 1) / CFFM_TOP_SLAB_ROWS;
        return n < 1 ? 1 : (int)n;
    }
    if (bwd_top_ok(s, B) && l >= bwd_top_first_layer(s)) {
        if (!top_wgrad_deferred(s, B)) return 256;
        const int64_t n = ((int64_t)B * S * S + 63) / 64;
        return n < 1 ? 1 : (n > 256 ? 256 : (int)n);
    }
    return conv_slabs_plain(s, B, l);
}

static inline void make_slab_plan(const cffm_shape_t* s, int32_t B, const cffm_theta_layout_t& tl, SlabPlan* p) {
    int n = 0;
    int64_t base = 0;
    auto add = [&](int64_t off, int64_t end, int nslab) {
        p->r[n].off = off; p->r[n].len = end - off; p->r[n].base = base; p->r[n].nslab = nslab;
        base += (end - off) * nslab;
        return n++;
    };
    const int64_t convs = tl.live > 0 ? tl.conv_w[0] : tl.d1_w;
    p->head_front = add(0, tl.inner_cw, small_slabs(B));
    p->inner = add(tl.inner_cw, convs, small_slabs(B));
    p->conv0 = n;
    for (int l = 0; l < tl.live; ++l)
        add(tl.conv_w[l], l + 1 < tl.live ? tl.conv_w[l + 1] : tl.d1_w, conv_slabs(s, B, l));
    p->head_back = add(tl.d1_w, tl.n, small_slabs(B));
    p->n = n;
    p->total = base;
}

// dL/dout of one example (CFFM.py:486-513); `out` is what head_fwd left in ws.out (sigmoid(logit) for log_loss)
__device__ __forceinline__ float head_dout(int loss, float out, float y, float invB, float L) {
    switch (loss) {
        case CFFM_LOSS_SQUARE_RMSE: return (out - y) * invB / L;
        case CFFM_LOSS_MSE: return 2.f * (out - y) * invB;
        case CFFM_LOSS_MAE: return (out > y ? 1.f : (out < y ? -1.f : 0.f)) * invB;
        case CFFM_LOSS_SQUARE_L2: return out - y;                 // d/dout of sum (y - out)^2 / 2
        case CFFM_LOSS_HYBRID:
            return 0.5f * (out - y) - 0.5f * invB * (y / (out + 1e-7f) - (1.f - y) / (1.f - out + 1e-7f));
        default: {
            const float s = out;
            return -(y / (s + 1e-7f) - (1.f - y) / (1.f - s + 1e-7f)) * invB * s * (1.f - s);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// activations, TF-1.14 functor semantics
// ---------------------------------------------------------------------------------------------
#define CFFM_SELU_SCALE 1.0507009873554804934193349852946f
#define CFFM_SELU_SCALE_ALPHA 1.7580993408473768599402175208123f

__device__ __forceinline__ float act_f(float x, int act) {
    switch (act) {
        case CFFM_ACT_RELU: return fmaxf(x, 0.f);
        case CFFM_ACT_PRELU: return fmaxf(x, 0.f) + 0.25f * (-fmaxf(-x, 0.f));
        case CFFM_ACT_ELU: return x < 0.f ? expf(x) - 1.f : x;
        case CFFM_ACT_SELU: return x < 0.f ? CFFM_SELU_SCALE_ALPHA * (expf(x) - 1.f) : CFFM_SELU_SCALE * x;
        default: return x * (0.5f * (1.f + erff(x * 0.70710678118654752440f)));
    }
}

// d act(x) / dx as TF autodiff yields it
__device__ __forceinline__ float act_grad_f(float x, int act) {
    switch (act) {
        case CFFM_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case CFFM_ACT_PRELU: return x > 0.f ? 1.f : (x < 0.f ? 0.25f : 0.f);
        case CFFM_ACT_ELU: return x < 0.f ? expf(x) : 1.f;                       // y + 1 for y < 0
        case CFFM_ACT_SELU: return x < 0.f ? CFFM_SELU_SCALE_ALPHA * expf(x) : CFFM_SELU_SCALE;  // y + scale_alpha
        default: {
            float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
            float pdf = expf(-0.5f * x * x) * 0.39894228040143267794f;
            return cdf + x * pdf;
        }
    }
}

// act on a value known to be >= 0 (output of the relu inside conv_layer, CFFM.py:478)
__device__ __forceinline__ float act_pos(float c, int act) {
    switch (act) {
        case CFFM_ACT_SELU: return CFFM_SELU_SCALE * c;
        case CFFM_ACT_GELU: return c * (0.5f * (1.f + erff(c * 0.70710678118654752440f)));
        default: return c;   // relu / prelu / elu are the identity on [0, inf)
    }
}

// d act(relu(z)) / dz expressed through c = relu(z)
__device__ __forceinline__ float act_relu_grad(float c, int act) {
    return c > 0.f ? act_grad_f(c, act) : 0.f;
}

// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier).  __syncthreads() also drains every
// outstanding global load and store (vmcnt(0)): a phase that ends with a burst of global stores then pays their full
// L2 round trip before the next phase may even issue its loads.  With this barrier the stores drain in the background
// and loads issued before it stay in flight.  Only for barriers that protect LDS data: anything one thread passes to
// another through GLOBAL memory still needs __syncthreads().
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---------------------------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------------------------
// 64-lane reductions on the DPP path (v_add_f32 ... row_shr / row_bcast): ~6 VALU ops, against six dependent
// ds_bpermute round trips (~60 cycles each) for the __shfl_xor butterfly.  Lane 63 ends up with the total,
// which v_readlane broadcasts as a wave-uniform value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v, float identity) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1, 0xf>(v, 0.f);     // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xf>(v, 0.f);     // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xf>(v, 0.f);    // row_half_mirror
    v += dpp_mov<0x140, 0xf>(v, 0.f);    // row_mirror          -> every lane holds its row's sum
    v += dpp_mov<0x142, 0xa>(v, 0.f);    // row_bcast:15 into rows 1 and 3
    v += dpp_mov<0x143, 0xc>(v, 0.f);    // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    const float ninf = -INFINITY;
    v = fmaxf(v, dpp_mov<0xB1, 0xf>(v, ninf));
    v = fmaxf(v, dpp_mov<0x4E, 0xf>(v, ninf));
    v = fmaxf(v, dpp_mov<0x141, 0xf>(v, ninf));
    v = fmaxf(v, dpp_mov<0x140, 0xf>(v, ninf));
    v = fmaxf(v, dpp_mov<0x142, 0xa>(v, ninf));
    v = fmaxf(v, dpp_mov<0x143, 0xc>(v, ninf));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// sum over the 8 (SIXTEEN = false) or 16 consecutive lanes of a DPP row; every lane of the group ends with it
template <bool SIXTEEN>
__device__ __forceinline__ float row_group_sum(float v) {
    v += dpp_mov<0xB1, 0xf>(v, 0.f);     // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xf>(v, 0.f);     // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xf>(v, 0.f);    // row_half_mirror: 8 lanes
    if (SIXTEEN) v += dpp_mov<0x140, 0xf>(v, 0.f);   // row_mirror: 16 lanes
    return v;
}

// Block-wide sum in a fixed order (bitwise reproducible); red must hold >= blockDim/64 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// pair p -> packed (i | j << 16) in the reference's row-major i<j order (CFFM.py:304-305)
__device__ __forceinline__ void build_pair_lut(uint32_t* lut, int F, int Pp) {
    for (int i = threadIdx.x; i < F; i += blockDim.x) {
        int base = i * (2 * F - i - 1) / 2;
        for (int j = i + 1; j < F; ++j) lut[base + j - i - 1] = (uint32_t)i | ((uint32_t)j << 16);
    }
    const int P = F * (F - 1) / 2;
    for (int p = P + threadIdx.x; p < Pp; p += blockDim.x) lut[p] = 0u;   // padded pairs -> (0,0), weights are 0
}

// n / d for 0 <= n < 2^19 and 1 <= d <= 2048 without the ~40-instruction integer division sequence:
// (n + 0.5) * (1/d) is at least 0.5/d away from every integer, far above the fp32 rounding error.
__device__ __forceinline__ int fast_div(int n, float inv_d) { return (int)(((float)n + 0.5f) * inv_d); }

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
