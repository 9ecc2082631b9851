// Optimiser step: tf.train.AdagradOptimizer(lr, initial_accumulator_value=1e-8).minimize (CFFM.py:523-524).
//
//   reduce_slabs    sum of the CFFM_NSLAB split-K partial gradients in slab order (bitwise reproducible)
//   dense_adagrad   acc += g*g; v -= lr*g/sqrt(acc)                        (no epsilon, TF semantics)
//   sparse_adagrad  IndexedSlices semantics: duplicate ids are summed FIRST, then one update per distinct
//                   row; rows not in the batch and their accumulators are untouched.  Implemented as a
//                   stable radix sort of (id, slot) followed by one wavefront per segment head that walks
//                   its segment in slot order - HBM-bound: per distinct row 5 x (K+D+1) x 4 bytes.
#include "internal.hpp"

#include <cmath>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

// One workgroup = 64 groups of four consecutive gradients (every member of theta and every slab range is 16-byte aligned,
// cffm_theta_layout) x 4 wavefronts, each adding a quarter of the range's slabs with 16-byte loads; the four partial sums
// meet in LDS and are added in wavefront order.  Bitwise reproducible, and four times the bytes in flight of the
// one-thread-per-gradient loop it replaces (that one kept 16 x 4-byte loads per lane in flight on 161 workgroups and
// reached 3 TB/s on the 26 MB of slabs of the frappe step: update_all 11.4 us; this form 9.8 us; an 8-way split over
// half-wavefronts, twice the workgroups, was slower again at 10.8 us).
#define REDUCE_GROUPS 64
static inline int reduce_slab_wgs(int64_t n) { return (int)((n / 4 + REDUCE_GROUPS - 1) / REDUCE_GROUPS); }
__device__ __forceinline__ void reduce_slabs_body(int bid, const float* __restrict__ gpart, int64_t n, const SlabPlan& sp,
                                                  float* __restrict__ grad, float* __restrict__ theta,
                                                  float* __restrict__ acc, float lr) {
    __shared__ float4 part[4][REDUCE_GROUPS];
    const int g = threadIdx.x & 63, q = threadIdx.x >> 6;          // group, quarter
    const int64_t i = ((int64_t)bid * REDUCE_GROUPS + g) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        int hit = 0;
        for (int r = 1; r < sp.n; ++r)
            if (i >= sp.r[r].off) hit = r;
        const SlabRange rg = sp.r[hit];
        const float4* src = reinterpret_cast<const float4*>(gpart + rg.base + (i - rg.off));
        const int64_t stride = rg.len / 4;
        const int per = (rg.nslab + 3) / 4, k0 = q * per, k1 = min(rg.nslab, k0 + per);
#pragma unroll 8
        for (int k = k0; k < k1; ++k) {                       // independent loads, fixed add order
            const float4 v = src[(int64_t)k * stride];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    part[q][g] = s;
    __syncthreads();
    if (q != 0 || i >= n) return;
#pragma unroll
    for (int e = 1; e < 4; ++e) {
        const float4 p = part[e][g];
        s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    *reinterpret_cast<float4*>(grad + i) = s;
    if (theta != nullptr) {                 // fused dense Adagrad (single-GPU step)
        float4 a = *reinterpret_cast<float4*>(acc + i), t = *reinterpret_cast<float4*>(theta + i);
        a.x += s.x * s.x; a.y += s.y * s.y; a.z += s.z * s.z; a.w += s.w * s.w;
        t.x -= lr * s.x / sqrtf(a.x); t.y -= lr * s.y / sqrtf(a.y); t.z -= lr * s.z / sqrtf(a.z); t.w -= lr * s.w / sqrtf(a.w);
        *reinterpret_cast<float4*>(acc + i) = a;
        *reinterpret_cast<float4*>(theta + i) = t;
    }
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ gpart, int64_t n, SlabPlan sp,
                                                           float* __restrict__ grad, float* __restrict__ theta,
                                                           float* __restrict__ acc, float lr) {
    reduce_slabs_body(blockIdx.x, gpart, n, sp, grad, theta, acc, lr);
}

// Late loss normalisation of the data-parallel step: the backward pass ran with dL/dout = (out - y) / Bg, i.e.
// WITHOUT the 1/L of the RMSE-style loss (CFFM.py:493), because L needs the loss-term sum over the GLOBAL batch,
// which only exists after the all-reduce that also carries the gradients.  Every gradient is linear in dL/dout,
// so the factor 1/L = rsqrt(sum / Bg + 1e-10) is applied here, on the summed gradient.
struct LateScale { const float* sum; float inv_Bg; int on; };
__device__ __forceinline__ float late_scale(const LateScale& ls) {
    return ls.on ? 1.f / sqrtf(ls.sum[0] * ls.inv_Bg + 1e-10f) : 1.f;
}

__device__ __forceinline__ void dense_adagrad_body(int bid, float* __restrict__ v, float* __restrict__ acc,
                                                   const float* __restrict__ grad, int64_t n, float lr, const LateScale& ls,
                                                   float* __restrict__ loss_out) {
    const int64_t i = (int64_t)bid * 256 + threadIdx.x;
    if (i == 0 && loss_out != nullptr) loss_out[0] = ls.on ? sqrtf(ls.sum[0] * ls.inv_Bg + 1e-10f) : ls.sum[0] * ls.inv_Bg;
    if (i >= n) return;
    const float g = grad[i] * late_scale(ls);
    const float a = acc[i] + g * g;
    acc[i] = a;
    v[i] -= lr * g / sqrtf(a);
}

__global__ __launch_bounds__(256) void dense_adagrad_kernel(float* __restrict__ v, float* __restrict__ acc,
                                                            const float* __restrict__ grad, int64_t n, float lr,
                                                            LateScale ls, float* __restrict__ loss_out) {
    dense_adagrad_body(blockIdx.x, v, acc, grad, n, lr, ls, loss_out);
}

// Sorted order of n <= 8192 unique keys (id << 32 | slot) without a sort: the place of a key is the number of keys below
// it.  Workgroup w of nwg places keys [w*kpw, (w+1)*kpw): its 256 threads split the n candidates, count per key in
// registers, wave/block-reduce, and the first kpw threads store their key at its rank.  ids are read with a stride
// (the id column of the packed rows of the data-parallel step, or a plain id array with stride 1).
#define RANK_KPW_MAX 32
__device__ __forceinline__ void rank_place_body(int wg, int nwg, const int32_t* __restrict__ ids, int64_t id_stride, int n,
                                                unsigned long long* __restrict__ out, float* cnt /* LDS [4][RANK_KPW_MAX] */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kpw = (n + nwg - 1) / nwg, k0 = wg * kpw;
    unsigned long long mine[RANK_KPW_MAX];
#pragma unroll
    for (int k = 0; k < RANK_KPW_MAX; ++k) {
        const int slot = k0 + k;
        mine[k] = (k < kpw && slot < n) ? (((unsigned long long)(unsigned)ids[(int64_t)slot * id_stride] << 32) | (unsigned)slot) : 0ull;
    }
    int c[RANK_KPW_MAX];
#pragma unroll
    for (int k = 0; k < RANK_KPW_MAX; ++k) c[k] = 0;
    for (int j = tid; j < n; j += 256) {
        const unsigned long long kj = ((unsigned long long)(unsigned)ids[(int64_t)j * id_stride] << 32) | (unsigned)j;
#pragma unroll
        for (int k = 0; k < RANK_KPW_MAX; ++k)
            if (k < kpw) c[k] += kj < mine[k] ? 1 : 0;
    }
#pragma unroll
    for (int k = 0; k < RANK_KPW_MAX; ++k) {
        if (k < kpw) {
            const float t = wave_sum((float)c[k]);               // counts <= 8192: exact in fp32
            if (lane == 0) cnt[wave * RANK_KPW_MAX + k] = t;
        }
    }
    __syncthreads();
    if (tid < kpw && k0 + tid < n) {
        const int rank = (int)(((cnt[tid] + cnt[RANK_KPW_MAX + tid]) + cnt[2 * RANK_KPW_MAX + tid]) + cnt[3 * RANK_KPW_MAX + tid]);
        const int slot = k0 + tid;
        out[rank] = ((unsigned long long)(unsigned)ids[(int64_t)slot * id_stride] << 32) | (unsigned)slot;
    }
}

// Global order of n_runs sorted runs of m keys each (every rank's own run, made by the single-launch forward) without
// sorting them again: the place of the key at (run r, position p) is p + sum over the other runs of the number of keys
// that precede it there - an upper bound on its id in the runs before r, a lower bound in the runs after r (the global
// slot r*m + local slot breaks id ties in run order).  Every workgroup stages all ids (4*n bytes) in LDS once and its
// 256 threads run n_runs - 1 binary searches each.
struct MergeArgs {
    const float* rows;            // n_runs blocks of [m*W | m keys (u64)]
    int64_t block_floats;         // m * (W + 2)
    int64_t keys_off;             // m * W
    int m, n_runs;
    unsigned long long* out;      // [n_runs * m] keys (id << 32 | global slot) in global order
};
__device__ __forceinline__ void merge_runs_body(int wg, const MergeArgs& a, unsigned* ids_l /* LDS [n] */) {
    const int n = a.m * a.n_runs;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int r = e / a.m, p = e - r * a.m;
        const unsigned long long k = reinterpret_cast<const unsigned long long*>(a.rows + (int64_t)r * a.block_floats + a.keys_off)[p];
        ids_l[e] = (unsigned)(k >> 32);
    }
    __syncthreads();
    const int e = wg * 256 + threadIdx.x;
    if (e >= n) return;
    const int r = e / a.m, p = e - r * a.m;
    const unsigned long long k = reinterpret_cast<const unsigned long long*>(a.rows + (int64_t)r * a.block_floats + a.keys_off)[p];
    const unsigned id = (unsigned)(k >> 32);
    int rank = p;
    for (int q = 0; q < a.n_runs; ++q) {
        if (q == r) continue;
        const unsigned* run = ids_l + q * a.m;
        int lo = 0, hi = a.m;                                  // first position whose id is > id (q < r) or >= id (q > r)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const unsigned v = run[mid];
            const bool before = q < r ? v <= id : v < id;
            if (before) lo = mid + 1; else hi = mid;
        }
        rank += lo;
    }
    a.out[rank] = ((unsigned long long)id << 32) | (unsigned long long)(unsigned)(r * a.m + (int)(k & 0xffffffffull));
}

__global__ __launch_bounds__(256) void dp_head_merge_kernel(float* __restrict__ v, float* __restrict__ acc, const float* __restrict__ grad,
                                                            int64_t n, float lr, LateScale ls, float* __restrict__ loss_out,
                                                            int n_dense, MergeArgs ma) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < n_dense) dense_adagrad_body(blockIdx.x, v, acc, grad, n, lr, ls, loss_out);
    else merge_runs_body(blockIdx.x - n_dense, ma, reinterpret_cast<unsigned*>(smem));
}

// first launch of cffm_dp_apply: dense Adagrad with the late 1/L ∥ placement of the gathered keys
__global__ __launch_bounds__(256) void dp_head_kernel(float* __restrict__ v, float* __restrict__ acc, const float* __restrict__ grad,
                                                      int64_t n, float lr, LateScale ls, float* __restrict__ loss_out, int n_dense,
                                                      const int32_t* __restrict__ ids, int64_t id_stride, int n_keys, int n_rank,
                                                      unsigned long long* __restrict__ keys_out) {
    __shared__ float cnt[4 * RANK_KPW_MAX];
    if ((int)blockIdx.x < n_dense) dense_adagrad_body(blockIdx.x, v, acc, grad, n, lr, ls, loss_out);
    else rank_place_body(blockIdx.x - n_dense, n_rank, ids, id_stride, n_keys, keys_out, cnt);
}

// last launch of cffm_dp_local: slab reduction (no update) ∥ packing of (id | dEi | dEo | dfb) rows + the local loss sum
struct PackArgs {
    const int32_t* ids;
    int64_t n_slots;
    int K, D, B;
    const float *dEi, *dEo, *dfb, *sqerr;
    float *sum_dst, *rows, *scalars;
    const unsigned long long* keys_sorted;   // non-NULL: appended after the rows as this rank's sorted run
};
__device__ __forceinline__ void pack_rows_body(int bid, int nblk, const PackArgs& a, float* red) {
    const int W = 1 + a.K + a.D + 1;
    const int64_t total = a.n_slots * W;
    if (bid == 0) {                                  // loss-term sum of this rank, fixed order
        float part = 0.f;
        for (int i = threadIdx.x; i < a.B; i += 256) part += a.sqerr[i];
        const float sum = block_sum(part, red);
        if (threadIdx.x == 0) { a.sum_dst[0] = sum; a.scalars[0] = sum; }
    }
    for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < total; i += (int64_t)nblk * 256) {
        const int64_t slot = i / W;
        const int c = (int)(i - slot * W);
        float v;
        if (c == 0) v = __int_as_float(a.ids[slot]);
        else if (c <= a.K) v = a.dEi ? a.dEi[slot * a.K + (c - 1)] : 0.f;           // disabled branch: its columns travel as zeros
        else if (c <= a.K + a.D) v = a.dEo ? a.dEo[slot * a.D + (c - 1 - a.K)] : 0.f;
        else v = a.dfb[slot];
        a.rows[i] = v;
    }
    if (a.keys_sorted != nullptr) {
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.rows + total);      // total * 4 bytes is 8-byte aligned: see cffm_dp_tail
        for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < a.n_slots; i += (int64_t)nblk * 256) dst[i] = a.keys_sorted[i];
    }
}
__global__ __launch_bounds__(256) void dp_tail_kernel(const float* __restrict__ gpart, int64_t n, SlabPlan sp, float* __restrict__ grad,
                                                      int n_reduce, PackArgs pa, int n_pack) {
    __shared__ float red[4];
    if ((int)blockIdx.x < n_reduce) reduce_slabs_body(blockIdx.x, gpart, n, sp, grad, nullptr, nullptr, 0.f);
    else pack_rows_body(blockIdx.x - n_reduce, n_pack, pa, red);
}

int cffm_dp_tail(const cffm_shape_t* s, const int32_t* ids, int32_t B, void* ws, float* grad, float* rows, bool with_run,
                 hipStream_t st) {
    cffm_theta_layout_t tl; cffm_ws_layout_t wl;
    cffm_theta_layout(s, &tl); cffm_ws_layout(s, B, &wl);
    char* w = (char*)ws;
    SlabPlan sp;
    make_slab_plan(s, B, tl, &sp);
    PackArgs pa;
    pa.ids = ids; pa.n_slots = (int64_t)B * s->F; pa.K = s->K; pa.D = s->D; pa.B = B;
    pa.dEi = s->inner_conv ? (const float*)(w + wl.dEi) : nullptr; pa.dEo = s->outer_conv ? (const float*)(w + wl.dEo) : nullptr;
    pa.dfb = (const float*)(w + wl.dfb);
    pa.sqerr = (const float*)(w + wl.sqerr); pa.sum_dst = grad + tl.n; pa.rows = rows; pa.scalars = (float*)(w + wl.scalars);
    // n_slots * W floats: W = K + D + 2 is even for the float4-aligned K, D this library accepts, so the run is 8-byte aligned
    pa.keys_sorted = with_run ? (const unsigned long long*)(w + wl.sort_vals) : nullptr;
    const int n_reduce = reduce_slab_wgs(tl.n);
    const int64_t total = pa.n_slots * (1 + s->K + s->D + 1);
    int n_pack = (int)((total + 1023) / 1024);
    if (n_pack > 2048) n_pack = 2048;
    hipLaunchKernelGGL(dp_tail_kernel, dim3(n_reduce + n_pack), dim3(256), 0, st, (const float*)(w + wl.gpart), (int64_t)tl.n, sp,
                       grad, n_reduce, pa, n_pack);
    CFFM_CHECK_LAUNCH();
    return 0;
}

// An id outside [0, M) is keyed as M: the radix sorts look at ceil(log2(M + 1)) id bits only, and a raw bad id whose low
// bits equal a valid id would land inside that id's run and split its segment in two (two wavefronts updating one row).
// All bad ids form ONE segment with id M, which every update kernel skips (id >= M).
__global__ __launch_bounds__(256) void pack_keys_kernel(const int32_t* __restrict__ ids, unsigned long long* keys, int64_t n,
                                                        int64_t id_stride, int M) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int raw = ids[i * id_stride];
        keys[i] = ((unsigned long long)(unsigned)((raw < 0 || raw >= M) ? M : raw) << 32) | (unsigned long long)i;
    }
}

#include "sort_body.hpp"

__global__ __launch_bounds__(256) void small_sort_kernel(const unsigned long long* __restrict__ in,
                                                         unsigned long long* __restrict__ out, int n, int id_bits) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    small_sort_body(in, nullptr, out, n, id_bits, smem);
}

// one wavefront per sorted position; only segment heads do work
struct SparseArgs {
    const unsigned long long* keys;
    int64_t n;
    int M, K, D;
    const float *dEi, *dEo, *dfb;
    float *inner, *outer, *fbias, *a_inner, *a_outer, *a_fbias;
    float lr;
    int64_t sEi, sEo, sfb;
    LateScale ls;
    int run_len;            // > 0: slot s lives in block s / run_len at local index s % run_len; blocks are run_stride floats apart
    int64_t run_stride;
    float inv_run_len;
};

__device__ __forceinline__ void sparse_adagrad_body(int bid, const SparseArgs& a) {
    const unsigned long long* __restrict__ keys = a.keys;
    const int64_t n = a.n;
    const int M = a.M, K = a.K, D = a.D;
    const float* __restrict__ dEi = a.dEi; const float* __restrict__ dEo = a.dEo; const float* __restrict__ dfb = a.dfb;
    const float gscale = late_scale(a.ls);
    const int64_t pos = (int64_t)bid * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pos >= n) return;
    const int id = (int)(keys[pos] >> 32);
    if (pos > 0 && (int)(keys[pos - 1] >> 32) == id) return;          // not a segment head
    if (id < 0 || id >= M) return;
    const int W = (dEi ? K : 0) + (dEo ? D : 0) + 1;     // columns: inner | outer | bias
    const int Ki = dEi ? K : 0;
    for (int c0 = 0; c0 < W; c0 += 64) {
        const int c = c0 + lane;
        float g = 0.f;
        if (c < W) {
            for (int64_t q = pos; q < n; ++q) {
                const unsigned long long kq = keys[q];
                if ((int)(kq >> 32) != id) break;
                int64_t sl = (int64_t)(kq & 0xffffffffull);
                int64_t boff = 0;
                if (a.run_len > 0) {
                    const int blk = fast_div((int)sl, a.inv_run_len);
                    boff = (int64_t)blk * a.run_stride;
                    sl -= (int64_t)blk * a.run_len;
                }
                g += c < Ki ? dEi[boff + sl * a.sEi + c] : (c < W - 1 ? dEo[boff + sl * a.sEo + (c - Ki)] : dfb[boff + sl * a.sfb]);
            }
            g *= gscale;
            float *vp, *ap;
            if (c < Ki) { vp = a.inner + (int64_t)id * K + c; ap = a.a_inner + (int64_t)id * K + c; }
            else if (c < W - 1) { vp = a.outer + (int64_t)id * D + (c - Ki); ap = a.a_outer + (int64_t)id * D + (c - Ki); }
            else { vp = a.fbias + id; ap = a.a_fbias + id; }
            const float acc = *ap + g * g;
            *ap = acc;
            *vp -= a.lr * g / sqrtf(acc);
        }
    }
}

__global__ __launch_bounds__(256) void sparse_adagrad_kernel(SparseArgs a) { sparse_adagrad_body(blockIdx.x, a); }

// The two halves of the update do not depend on each other (dense slabs vs table rows): one launch, two roles.
__global__ __launch_bounds__(256) void update_all_kernel(const float* __restrict__ gpart, int64_t n, SlabPlan sp,
                                                         float* __restrict__ grad, float* __restrict__ theta,
                                                         float* __restrict__ acc, float lr, int n_reduce, SparseArgs sa) {
    if ((int)blockIdx.x < n_reduce) {
        PHASE_MARKB(26, blockIdx.x);
        reduce_slabs_body(blockIdx.x, gpart, n, sp, grad, theta, acc, lr);
        PHASE_MARKB(27, blockIdx.x);
    } else {
        PHASE_MARKB(28, blockIdx.x - n_reduce);
        sparse_adagrad_body(blockIdx.x - n_reduce, sa);
        PHASE_MARKB(29, blockIdx.x - n_reduce);
    }
}
#ifdef CFFM_PHASE_TIMERS
extern "C" int cffm_debug_upd_times(unsigned long long* host32) {
    return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(cffm_bwd_times), sizeof(cffm_bwd_times));
}
#endif

extern "C" int cffm_reduce_slabs(const cffm_shape_t* s, void* ws, int32_t B, float* grad, void* stream) {
    return cffm_reduce_slabs_impl(s, ws, B, grad, nullptr, nullptr, 0.f, (hipStream_t)stream);
}

int cffm_reduce_slabs_impl(const cffm_shape_t* s, void* ws, int32_t B, float* grad, float* theta, float* acc, float lr,
                           hipStream_t stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    cffm_theta_layout_t tl; cffm_ws_layout_t wl;
    cffm_theta_layout(s, &tl); cffm_ws_layout(s, B, &wl);
    SlabPlan sp;
    make_slab_plan(s, B, tl, &sp);
    const float* gpart = (const float*)((char*)ws + wl.gpart);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)reduce_slab_wgs(tl.n)), dim3(256), 0, stream, gpart,
                       (int64_t)tl.n, sp, grad, theta, acc, lr);
    CFFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cffm_dense_adagrad(float* theta, float* acc, const float* grad, int64_t n, float lr, void* stream) {
    if (n <= 0) return 0;
    LateScale ls = {nullptr, 0.f, 0};
    hipLaunchKernelGGL(dense_adagrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       theta, acc, grad, n, lr, ls, (float*)nullptr);
    CFFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cffm_sparse_adagrad(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* acc,
                                   const int32_t* ids, int64_t n_rows, const float* dEi, const float* dEo,
                                   const float* dfb, void* ws, int32_t B_ws, void* stream) {
    return cffm_sparse_adagrad_impl(s, tab, acc, ids, n_rows, dEi, dEo, dfb, ws, B_ws, false, (hipStream_t)stream);
}

int cffm_sparse_adagrad_impl(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* acc,
                             const int32_t* ids, int64_t n_rows, const float* dEi, const float* dEo, const float* dfb,
                             void* ws, int32_t B_ws, bool prepacked, hipStream_t st) {
    int rc = cffm_sort_keys_impl(s, ids, n_rows, ws, B_ws, prepacked, st);
    if (rc) return rc;
    return cffm_sparse_apply_impl(s, tab, acc, n_rows, dEi, dEo, dfb, ws, B_ws, st);
}

int cffm_sort_keys_impl(const cffm_shape_t* s, const int32_t* ids, int64_t n_rows, void* ws, int32_t B_ws, bool prepacked,
                        hipStream_t st, int64_t id_stride) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (n_rows <= 0) return 0;
    if (n_rows > (int64_t)B_ws * s->F) return CFFM_ERR_BAD_SHAPE;
    cffm_ws_layout_t wl;
    cffm_ws_layout(s, B_ws, &wl);
    char* w = (char*)ws;
    unsigned long long* keys_in = (unsigned long long*)(w + wl.sort_keys);     // (id << 32) | slot
    unsigned long long* keys_out = (unsigned long long*)(w + wl.sort_vals);
    void* tmp = (void*)(w + wl.sort_tmp);
    size_t tmp_bytes = 0;
    int bits = 1;
    while ((1ll << bits) <= (long long)s->M && bits < 31) ++bits;      // ids 0 .. M (M = the key of every out-of-range id)
    // slots are unique, so sorting the packed keys IS the stable sort by id with slots ascending inside a segment
    if (!prepacked) {
        hipLaunchKernelGGL(pack_keys_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, ids, keys_in, n_rows, id_stride, s->M);
        CFFM_CHECK_LAUNCH();
    }
    if (n_rows <= 4096) {                       // one workgroup, one launch
        {
            hipError_t e0 = hipFuncSetAttribute((const void*)small_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMALL_SORT_LDS);
            if (e0 != hipSuccess) return (int)e0;
        }
        hipLaunchKernelGGL(small_sort_kernel, dim3(1), dim3(256), SMALL_SORT_LDS, st, keys_in, keys_out, (int)n_rows, bits);
        CFFM_CHECK_LAUNCH();
        return 0;
    }
    hipError_t e = rocprim::radix_sort_keys((void*)nullptr, tmp_bytes, keys_in, keys_out, (size_t)n_rows, 0u,
                                            (unsigned)(32 + bits), st);
    if (e != hipSuccess) return (int)e;
    if (tmp_bytes > (size_t)wl.sort_tmp_bytes) return CFFM_ERR_BAD_SHAPE;
    e = rocprim::radix_sort_keys(tmp, tmp_bytes, keys_in, keys_out, (size_t)n_rows, 0u, (unsigned)(32 + bits), st);
    return e == hipSuccess ? 0 : (int)e;
}

static void fill_sparse_args(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* acc, int64_t n_rows,
                             const float* dEi, int64_t sEi, const float* dEo, int64_t sEo, const float* dfb, int64_t sfb,
                             void* ws, int32_t B_ws, LateScale ls, SparseArgs* out) {
    cffm_ws_layout_t wl;
    cffm_ws_layout(s, B_ws, &wl);
    SparseArgs& a = *out;
    a.keys = (const unsigned long long*)((char*)ws + wl.sort_vals);
    a.n = n_rows; a.M = s->M; a.K = s->K; a.D = s->D;
    a.dEi = dEi; a.dEo = dEo; a.dfb = dfb;
    a.inner = tab->inner_emb; a.outer = tab->outer_emb; a.fbias = tab->feat_bias;
    a.a_inner = acc->inner_emb; a.a_outer = acc->outer_emb; a.a_fbias = acc->feat_bias;
    a.lr = s->lr; a.sEi = sEi; a.sEo = sEo; a.sfb = sfb; a.ls = ls;
    a.run_len = 0; a.run_stride = 0; a.inv_run_len = 0.f;
}

int cffm_sparse_apply_impl(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* acc, int64_t n_rows,
                           const float* dEi, const float* dEo, const float* dfb, void* ws, int32_t B_ws, hipStream_t st) {
    LateScale ls = {nullptr, 0.f, 0};
    return cffm_sparse_apply_strided(s, tab, acc, n_rows, dEi, s->K, dEo, s->D, dfb, 1, ws, B_ws, ls, st);
}

int cffm_sparse_apply_strided(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* acc, int64_t n_rows,
                              const float* dEi, int64_t sEi, const float* dEo, int64_t sEo, const float* dfb, int64_t sfb,
                              void* ws, int32_t B_ws, LateScale ls, hipStream_t st) {
    if (n_rows <= 0) return 0;
    cffm_ws_layout_t wl;
    cffm_ws_layout(s, B_ws, &wl);
    SparseArgs a;
    fill_sparse_args(s, tab, acc, n_rows, dEi, sEi, dEo, sEo, dfb, sfb, ws, B_ws, ls, &a);
    hipLaunchKernelGGL(sparse_adagrad_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, a);
    CFFM_CHECK_LAUNCH();
    return 0;
}

// fused single-GPU update: slab reduction + dense Adagrad and the sparse table update in one launch
int cffm_update_all(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* tab_acc, float* theta,
                    float* theta_acc, float* grad, void* ws, int32_t B, hipStream_t st) {
    cffm_theta_layout_t tl; cffm_ws_layout_t wl;
    cffm_theta_layout(s, &tl); cffm_ws_layout(s, B, &wl);
    char* w = (char*)ws;
    SlabPlan sp;
    make_slab_plan(s, B, tl, &sp);
    const int64_t n_rows = (int64_t)B * s->F;
    LateScale ls = {nullptr, 0.f, 0};
    SparseArgs a;
    fill_sparse_args(s, tab, tab_acc, n_rows, (const float*)(w + wl.dEi), s->K, (const float*)(w + wl.dEo), s->D,
                     (const float*)(w + wl.dfb), 1, ws, B, ls, &a);
    const int n_reduce = reduce_slab_wgs(tl.n);
    hipLaunchKernelGGL(update_all_kernel, dim3((unsigned)(n_reduce + (n_rows + 3) / 4)), dim3(256), 0, st,
                       (const float*)(w + wl.gpart), (int64_t)tl.n, sp, grad, theta, theta_acc, s->lr, n_reduce, a);
    CFFM_CHECK_LAUNCH();
    return 0;
}

// Data-parallel apply (cffm_amd/dist.py): grad_sum = all-reduced [theta.n gradients | pad | loss-term sum at index
// theta.n], rows = all-gathered packed rows [n_rows][1 + K + D + 1] = (id bits | dEi | dEo | dfb).
extern "C" int cffm_dp_apply(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* acc, float* theta,
                             float* theta_acc, const float* grad_sum, int64_t B_global, const float* rows,
                             int64_t n_rows, void* ws, int32_t B_ws, float* loss_out, int32_t n_runs, void* stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    cffm_theta_layout_t tl;
    cffm_theta_layout(s, &tl);
    LateScale ls = {grad_sum + tl.n, 1.f / (float)B_global, s->loss == CFFM_LOSS_SQUARE_RMSE ? 1 : 0};
    const int64_t W = 1 + s->K + s->D + 1;
    // a disabled branch (CFFM.py:301, :348) has no table: its columns of the rows are zeros and are not applied
    const float* rEi = s->inner_conv ? rows + 1 : nullptr;
    const float* rEo = s->outer_conv ? rows + 1 + s->K : nullptr;
    const int n_dense = (int)((tl.n + 255) / 256);
    if (n_runs > 0) {
        // sorted runs (one per rank, from cffm_dp_local): merge by rank, rows addressed block-wise
        if (n_rows <= 0 || n_rows % n_runs || n_rows > (int64_t)B_ws * s->F || n_rows * 4 > 150 * 1024) return CFFM_ERR_BAD_SHAPE;
        const int m = (int)(n_rows / n_runs);
        if (m % s->F || !cffm_fwd_all_ok(s, m / s->F)) return CFFM_ERR_UNSUPPORTED;      // the runs only exist on that path
        cffm_ws_layout_t wl;
        cffm_ws_layout(s, B_ws, &wl);
        MergeArgs ma;
        ma.rows = rows; ma.block_floats = (int64_t)m * (W + 2); ma.keys_off = (int64_t)m * W; ma.m = m; ma.n_runs = n_runs;
        ma.out = (unsigned long long*)((char*)ws + wl.sort_vals);
        const size_t lds = (size_t)n_rows * 4;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)dp_head_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(dp_head_merge_kernel, dim3(n_dense + (unsigned)((n_rows + 255) / 256)), dim3(256), lds, st, theta, theta_acc,
                           grad_sum, (int64_t)tl.n, s->lr, ls, loss_out, n_dense, ma);
        CFFM_CHECK_LAUNCH();
        SparseArgs a;
        fill_sparse_args(s, tab, acc, n_rows, rEi, W, rEo, W, rows + 1 + s->K + s->D, W, ws, B_ws, ls, &a);
        a.run_len = m; a.run_stride = ma.block_floats; a.inv_run_len = 1.f / (float)m;
        hipLaunchKernelGGL(sparse_adagrad_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, a);
        CFFM_CHECK_LAUNCH();
        return 0;
    }
    if (n_rows > 0 && n_rows <= 8192 && n_rows <= (int64_t)B_ws * s->F) {
        // dense update and key placement are independent: one launch, two roles
        cffm_ws_layout_t wl;
        cffm_ws_layout(s, B_ws, &wl);
        const int n_rank = 256;                              // kpw = ceil(n_rows / 256) <= 32 keys per workgroup
        hipLaunchKernelGGL(dp_head_kernel, dim3(n_dense + n_rank), dim3(256), 0, st, theta, theta_acc, grad_sum, (int64_t)tl.n,
                           s->lr, ls, loss_out, n_dense, (const int32_t*)rows, W, (int)n_rows, n_rank,
                           (unsigned long long*)((char*)ws + wl.sort_vals));
        CFFM_CHECK_LAUNCH();
    } else {
        hipLaunchKernelGGL(dense_adagrad_kernel, dim3((unsigned)n_dense), dim3(256), 0, st, theta, theta_acc, grad_sum,
                           (int64_t)tl.n, s->lr, ls, loss_out);
        CFFM_CHECK_LAUNCH();
        rc = cffm_sort_keys_impl(s, (const int32_t*)rows, n_rows, ws, B_ws, false, st, W);
        if (rc) return rc;
    }
    return cffm_sparse_apply_strided(s, tab, acc, n_rows, rEi, W, rEo, W, rows + 1 + s->K + s->D, W, ws, B_ws, ls, st);
}

// ---- data-parallel step for SMALL vocabularies: the tables' gradients travel as one dense buffer --------------------
// When M*(K+D+1) floats are fewer than what all ranks' row gradients add up to (frappe: 350 K floats against
// N * 174 K), each rank scatters its duplicates-summed row gradients into a zeroed dense [M][K | D | 1] image that
// rides behind the dense-parameter gradient in ONE all-reduce; afterwards every rank sweeps the whole tables.  Rows
// nobody looked up carry an exact 0: acc + 0*0 and w - lr*0/sqrt(acc) leave them bit-identical, which is TF's sparse
// semantics without a mask.  flat = [theta.n gradients | loss sum | pad to n4 | Gi M*K | Go M*D | Gfb M].
static inline int64_t dp_dense_table_off(const cffm_theta_layout_t& tl) { return ((int64_t)tl.n + 4 + 3) / 4 * 4; }

struct ScatterArgs {
    const unsigned long long* keys;   // this rank's sorted keys
    int64_t n;
    int M, K, D, B;
    const float *dEi, *dEo, *dfb, *sqerr;
    float *Gi, *Go, *Gfb, *sum_dst, *scalars;
};
__device__ __forceinline__ void scatter_rows_body(int bid, const ScatterArgs& a, float* red) {
    if (bid == 0) {                                  // role 0 of this range: loss-term sum of this rank, fixed order
        float part = 0.f;
        for (int i = threadIdx.x; i < a.B; i += 256) part += a.sqerr[i];
        const float sum = block_sum(part, red);
        if (threadIdx.x == 0) { a.sum_dst[0] = sum; a.scalars[0] = sum; }
        return;
    }
    const int64_t pos = (int64_t)(bid - 1) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pos >= a.n) return;
    const int id = (int)(a.keys[pos] >> 32);
    if (pos > 0 && (int)(a.keys[pos - 1] >> 32) == id) return;
    if (id < 0 || id >= a.M) return;
    const int K = a.K, D = a.D, W = K + D + 1;
    for (int c0 = 0; c0 < W; c0 += 64) {
        const int c = c0 + lane;
        if (c >= W) continue;
        float g = 0.f;
        for (int64_t q = pos; q < a.n; ++q) {
            const unsigned long long kq = a.keys[q];
            if ((int)(kq >> 32) != id) break;
            const int64_t sl = (int64_t)(kq & 0xffffffffull);
            g += c < K ? a.dEi[sl * K + c] : (c < K + D ? a.dEo[sl * D + (c - K)] : a.dfb[sl]);
        }
        if (c < K) a.Gi[(int64_t)id * K + c] = g;
        else if (c < K + D) a.Go[(int64_t)id * D + (c - K)] = g;
        else a.Gfb[id] = g;
    }
}
__global__ __launch_bounds__(256) void dp_tail_dense_kernel(const float* __restrict__ gpart, int64_t n, SlabPlan sp,
                                                            float* __restrict__ grad, int n_reduce, ScatterArgs sa) {
    __shared__ float red[4];
    if ((int)blockIdx.x < n_reduce) reduce_slabs_body(blockIdx.x, gpart, n, sp, grad, nullptr, nullptr, 0.f);
    else scatter_rows_body(blockIdx.x - n_reduce, sa, red);
}

int cffm_dp_tail_dense(const cffm_shape_t* s, int32_t B, void* ws, float* flat, hipStream_t st) {
    cffm_theta_layout_t tl; cffm_ws_layout_t wl;
    cffm_theta_layout(s, &tl); cffm_ws_layout(s, B, &wl);
    char* w = (char*)ws;
    SlabPlan sp;
    make_slab_plan(s, B, tl, &sp);
    const int64_t toff = dp_dense_table_off(tl);
    ScatterArgs sa;
    sa.keys = (const unsigned long long*)(w + wl.sort_vals); sa.n = (int64_t)B * s->F;
    sa.M = s->M; sa.K = s->K; sa.D = s->D; sa.B = B;
    sa.dEi = (const float*)(w + wl.dEi); sa.dEo = (const float*)(w + wl.dEo); sa.dfb = (const float*)(w + wl.dfb);
    sa.sqerr = (const float*)(w + wl.sqerr);
    sa.Gi = flat + toff; sa.Go = sa.Gi + (int64_t)s->M * s->K; sa.Gfb = sa.Go + (int64_t)s->M * s->D;
    sa.sum_dst = flat + tl.n; sa.scalars = (float*)(w + wl.scalars);
    const int n_reduce = reduce_slab_wgs(tl.n);
    const int n_scatter = 1 + (int)((sa.n + 3) / 4);
    hipLaunchKernelGGL(dp_tail_dense_kernel, dim3(n_reduce + n_scatter), dim3(256), 0, st, (const float*)(w + wl.gpart),
                       (int64_t)tl.n, sp, flat, n_reduce, sa);
    CFFM_CHECK_LAUNCH();
    return 0;
}

struct DenseTableArgs { float *w[3], *a[3]; float* g[3]; int64_t n[3]; };
__global__ __launch_bounds__(256) void dp_apply_dense_kernel(float* __restrict__ v, float* __restrict__ acc, const float* __restrict__ grad,
                                                             int64_t n, float lr, LateScale ls, float* __restrict__ loss_out, int n_dense,
                                                             DenseTableArgs t) {
    if ((int)blockIdx.x < n_dense) { dense_adagrad_body(blockIdx.x, v, acc, grad, n, lr, ls, loss_out); return; }
    int64_t i = (int64_t)(blockIdx.x - n_dense) * 256 + threadIdx.x;
    const float gs = late_scale(ls);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (i < t.n[k]) {
            const float g = t.g[k][i] * gs;
            t.g[k][i] = 0.f;                                 // zero on exit: the next step's scatter finds a clean image (no memset)
            const float a = t.a[k][i] + g * g;               // g == 0 (row not looked up by any rank): a and w unchanged
            t.a[k][i] = a;
            t.w[k][i] -= lr * g / sqrtf(a);
            return;
        }
        i -= t.n[k];
    }
}

extern "C" int64_t cffm_dp_dense_floats(const cffm_shape_t* s) {
    if (check_shape(s)) return -1;
    cffm_theta_layout_t tl;
    cffm_theta_layout(s, &tl);
    return dp_dense_table_off(tl) + (int64_t)s->M * (s->K + s->D + 1);
}

extern "C" int cffm_dp_apply_dense(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* acc, float* theta,
                                   float* theta_acc, float* flat_sum, int64_t B_global, float* loss_out, void* stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (!s->inner_conv || !s->outer_conv) return CFFM_ERR_UNSUPPORTED;
    cffm_theta_layout_t tl;
    cffm_theta_layout(s, &tl);
    const int64_t toff = dp_dense_table_off(tl);
    LateScale ls = {flat_sum + tl.n, 1.f / (float)B_global, s->loss == CFFM_LOSS_SQUARE_RMSE ? 1 : 0};
    DenseTableArgs t;
    t.w[0] = tab->inner_emb; t.w[1] = tab->outer_emb; t.w[2] = tab->feat_bias;
    t.a[0] = acc->inner_emb; t.a[1] = acc->outer_emb; t.a[2] = acc->feat_bias;
    t.n[0] = (int64_t)s->M * s->K; t.n[1] = (int64_t)s->M * s->D; t.n[2] = s->M;
    t.g[0] = flat_sum + toff; t.g[1] = t.g[0] + t.n[0]; t.g[2] = t.g[1] + t.n[1];
    const int n_dense = (int)((tl.n + 255) / 256);
    const int64_t nt = t.n[0] + t.n[1] + t.n[2];
    hipLaunchKernelGGL(dp_apply_dense_kernel, dim3((unsigned)(n_dense + (nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, theta,
                       theta_acc, flat_sum, (int64_t)tl.n, s->lr, ls, loss_out, n_dense, t);
    CFFM_CHECK_LAUNCH();
    return 0;
}

// ---- regularised square loss (CFFM.py:489-491): the l2 terms make the table gradients dense ------------------------
// 1) segment heads of the sorted keys write the duplicates-summed row gradients into zeroed dense buffers Gi/Go and
//    apply the (still sparse) feature_bias update; 2) a dense sweep applies Adagrad with g = G + lamda * w to every row.
__global__ __launch_bounds__(256) void scatter_rows_l2_kernel(const unsigned long long* __restrict__ keys, int64_t n, int M,
                                                              int K, int D, const float* __restrict__ dEi,
                                                              const float* __restrict__ dEo, const float* __restrict__ dfb,
                                                              float* __restrict__ Gi, float* __restrict__ Go,
                                                              float* __restrict__ fbias, float* __restrict__ a_fbias, float lr) {
    const int64_t pos = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pos >= n) return;
    const int id = (int)(keys[pos] >> 32);
    if (pos > 0 && (int)(keys[pos - 1] >> 32) == id) return;
    if (id < 0 || id >= M) return;
    const int W = K + D + 1;
    for (int c0 = 0; c0 < W; c0 += 64) {
        const int c = c0 + lane;
        if (c >= W) continue;
        float g = 0.f;
        for (int64_t q = pos; q < n; ++q) {
            const unsigned long long kq = keys[q];
            if ((int)(kq >> 32) != id) break;
            const int64_t sl = (int64_t)(kq & 0xffffffffull);
            g += c < K ? dEi[sl * K + c] : (c < K + D ? dEo[sl * D + (c - K)] : dfb[sl]);
        }
        if (c < K) Gi[(int64_t)id * K + c] = g;
        else if (c < K + D) Go[(int64_t)id * D + (c - K)] = g;
        else {
            const float a = a_fbias[id] + g * g;
            a_fbias[id] = a;
            fbias[id] -= lr * g / sqrtf(a);
        }
    }
}

__global__ __launch_bounds__(256) void table_adagrad_l2_kernel(float* __restrict__ w, float* __restrict__ acc,
                                                               const float* __restrict__ G, int64_t n, float lam, float lr) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = G[i] + lam * w[i];
    const float a = acc[i] + g * g;
    acc[i] = a;
    w[i] -= lr * g / sqrtf(a);
}

int cffm_tables_adagrad_l2(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* acc, const int32_t* ids,
                           int64_t n_rows, void* ws, int32_t B_ws, hipStream_t st) {
    cffm_ws_layout_t wl;
    cffm_ws_layout(s, B_ws, &wl);
    char* w = (char*)ws;
    float* Gi = (float*)(w + wl.Gi);
    float* Go = (float*)(w + wl.Go);
    const int64_t ni = (int64_t)s->M * s->K, no = (int64_t)s->M * s->D;
    hipError_t e = hipMemsetAsync(Gi, 0, (size_t)ni * 4, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(Go, 0, (size_t)no * 4, st);
    if (e != hipSuccess) return (int)e;
    int rc = cffm_sort_keys_impl(s, ids, n_rows, ws, B_ws, true, st);
    if (rc) return rc;
    const unsigned long long* keys = (const unsigned long long*)(w + wl.sort_vals);
    hipLaunchKernelGGL(scatter_rows_l2_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, keys, n_rows, s->M, s->K,
                       s->D, (const float*)(w + wl.dEi), (const float*)(w + wl.dEo), (const float*)(w + wl.dfb), Gi, Go,
                       tab->feat_bias, acc->feat_bias, s->lr);
    CFFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(table_adagrad_l2_kernel, dim3((unsigned)((ni + 255) / 256)), dim3(256), 0, st, tab->inner_emb,
                       acc->inner_emb, Gi, ni, s->lamda, s->lr);
    CFFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(table_adagrad_l2_kernel, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, st, tab->outer_emb,
                       acc->outer_emb, Go, no, s->lamda_att, s->lr);       // quirk Q13: lamda_att scales the outer table
    CFFM_CHECK_LAUNCH();
    return 0;
}

// ---- the other optimizers of CFFM.py:519-529 (TF-1.14 semantics) ------------------------------------------------------
struct OptConst { int opt; float lr, lr_t, b1, b2, omb1, omb2, eps, mom; };   // omb* = 1 - beta, rounded once from double

__device__ __forceinline__ void opt_update(float& w, float* s1, float* s2, float g, const OptConst& c) {
    if (c.opt == CFFM_OPT_SGD) {
        w -= c.lr * g;
    } else if (c.opt == CFFM_OPT_MOMENTUM) {
        const float a = c.mom * (*s1) + g;
        *s1 = a;
        w -= c.lr * a;
    } else {                                      // Adam
        const float m = c.b1 * (*s1) + c.omb1 * g;
        const float v = c.b2 * (*s2) + c.omb2 * g * g;
        *s1 = m; *s2 = v;
        w -= c.lr_t * m / (sqrtf(v) + c.eps);
    }
}

// lam != 0: the l2_regularizer term of the regularised square loss (CFFM.py:489-491), g = grad + lam * w
__global__ __launch_bounds__(256) void dense_opt_kernel(float* __restrict__ w, float* __restrict__ s1, float* __restrict__ s2,
                                                        const float* __restrict__ grad, int64_t n, OptConst c, float lam) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float wv = w[i];
    opt_update(wv, s1 ? s1 + i : nullptr, s2 ? s2 + i : nullptr, (grad ? grad[i] : 0.f) + lam * wv, c);
    w[i] = wv;
}

// SGD / Momentum on the touched rows (duplicates summed first, in slot order); Adam: the summed rows go to the dense
// buffers G* and dense_opt_kernel then sweeps every row (TF's non-lazy sparse Adam moves all of them)
__global__ __launch_bounds__(256) void sparse_opt_kernel(const unsigned long long* __restrict__ keys, int64_t n, int M, int K, int D,
                                                         const float* __restrict__ dEi, const float* __restrict__ dEo,
                                                         const float* __restrict__ dfb, cffm_tables_t tab, cffm_tables_t st1,
                                                         float* __restrict__ Gi, float* __restrict__ Go, float* __restrict__ Gfb,
                                                         OptConst c, int dense_tables) {
    const int64_t pos = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pos >= n) return;
    const int id = (int)(keys[pos] >> 32);
    if (pos > 0 && (int)(keys[pos - 1] >> 32) == id) return;
    if (id < 0 || id >= M) return;
    const int W = K + D + 1;
    for (int c0 = 0; c0 < W; c0 += 64) {
        const int col = c0 + lane;
        if (col >= W) continue;
        // a disabled branch (CFFM.py:301, :348) has no table variable: nothing to update
        if ((col < K && dEi == nullptr) || (col >= K && col < K + D && dEo == nullptr)) continue;
        float g = 0.f;
        for (int64_t q = pos; q < n; ++q) {
            const unsigned long long kq = keys[q];
            if ((int)(kq >> 32) != id) break;
            const int64_t sl = (int64_t)(kq & 0xffffffffull);
            g += col < K ? dEi[sl * K + col] : (col < K + D ? dEo[sl * D + (col - K)] : dfb[sl]);
        }
        const int64_t off = col < K ? (int64_t)id * K + col : (col < K + D ? (int64_t)id * D + (col - K) : (int64_t)id);
        float* wt = col < K ? tab.inner_emb : (col < K + D ? tab.outer_emb : tab.feat_bias);
        // dense gradient of a table (Adam's non-lazy sparse apply; the regularised loss): the summed rows go to G*
        if (c.opt == CFFM_OPT_ADAM || (dense_tables && col < K + D)) {
            (col < K ? Gi : (col < K + D ? Go : Gfb))[off] = g;
        } else {
            float* s1 = c.opt == CFFM_OPT_MOMENTUM ? (col < K ? st1.inner_emb : (col < K + D ? st1.outer_emb : st1.feat_bias)) + off : nullptr;
            float wv = wt[off];
            opt_update(wv, s1, nullptr, g, c);
            wt[off] = wv;
        }
    }
}

int cffm_apply_opt(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* st1, const cffm_tables_t* st2,
                   float* theta, float* th1, float* th2, const float* grad, const int32_t* ids, int64_t n_rows, void* ws,
                   int32_t B_ws, int64_t step, hipStream_t st) {
    cffm_ws_layout_t wl; cffm_theta_layout_t tl;
    cffm_ws_layout(s, B_ws, &wl); cffm_theta_layout(s, &tl);
    char* w = (char*)ws;
    OptConst c;
    c.opt = s->optimizer; c.lr = s->lr; c.b1 = 0.9f; c.b2 = 0.999f; c.omb1 = (float)(1.0 - 0.9); c.omb2 = (float)(1.0 - 0.999); c.eps = 1e-8f; c.mom = 0.95f;
    c.lr_t = (float)((double)s->lr * sqrt(1.0 - pow(0.999, (double)step)) / (1.0 - pow(0.9, (double)step)));
    hipLaunchKernelGGL(dense_opt_kernel, dim3((unsigned)((tl.n + 255) / 256)), dim3(256), 0, st, theta, th1, th2, grad,
                       (int64_t)tl.n, c, 0.f);
    CFFM_CHECK_LAUNCH();
    // regularised square loss (CFFM.py:489-491): the l2 terms make the gradients of the two embedding tables dense for
    // every optimizer (IndexedSlices + dense aggregates to dense); feature_bias stays sparse (Adam: non-lazy, all rows)
    const bool l2 = s->loss == CFFM_LOSS_SQUARE_L2;
    const bool adam = c.opt == CFFM_OPT_ADAM;
    float *Gi = nullptr, *Go = nullptr, *Gfb = nullptr;
    const int64_t ni = (int64_t)s->M * s->K, no = (int64_t)s->M * s->D, nf = s->M;
    if (adam || l2) {
        Gi = (float*)(w + wl.Gi); Go = (float*)(w + wl.Go); Gfb = (float*)(w + wl.Gfb);
        hipError_t e = hipMemsetAsync(Gi, 0, (size_t)(wl.Gfb + nf * 4 - wl.Gi), st);     // the three buffers are contiguous
        if (e != hipSuccess) return (int)e;
    }
    int rc = cffm_sort_keys_impl(s, ids, n_rows, ws, B_ws, true, st);
    if (rc) return rc;
    cffm_tables_t t1 = st1 ? *st1 : cffm_tables_t{nullptr, nullptr, nullptr};
    cffm_tables_t t2 = st2 ? *st2 : cffm_tables_t{nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(sparse_opt_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st,
                       (const unsigned long long*)(w + wl.sort_vals), n_rows, s->M, s->K, s->D,
                       s->inner_conv ? (const float*)(w + wl.dEi) : nullptr, s->outer_conv ? (const float*)(w + wl.dEo) : nullptr,
                       (const float*)(w + wl.dfb), *tab, t1, Gi, Go, Gfb, c, l2 ? 1 : 0);
    CFFM_CHECK_LAUNCH();
    if (adam || l2) {      // dense sweeps; a disabled branch has no table variable and is left alone
        if (s->inner_conv)
            hipLaunchKernelGGL(dense_opt_kernel, dim3((unsigned)((ni + 255) / 256)), dim3(256), 0, st, tab->inner_emb, t1.inner_emb,
                               t2.inner_emb, (const float*)Gi, ni, c, l2 ? s->lamda : 0.f);
        if (s->outer_conv)
            hipLaunchKernelGGL(dense_opt_kernel, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, st, tab->outer_emb, t1.outer_emb,
                               t2.outer_emb, (const float*)Go, no, c, l2 ? s->lamda_att : 0.f);    // quirk Q13: lamda_att scales the outer table
        if (adam)
            hipLaunchKernelGGL(dense_opt_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, tab->feat_bias, t1.feat_bias,
                               t2.feat_bias, (const float*)Gfb, nf, c, 0.f);
        CFFM_CHECK_LAUNCH();
    }
    return 0;
}
