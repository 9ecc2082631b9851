// Embedding row gather: tf.nn.embedding_lookup x3 (CFFM.py:303, :354, :422).
//
// HBM-bound.  A row is K*4 (inner) or D*4 (outer) contiguous bytes; one lane moves 16 B
// (global_load_dwordx4), so a row of 32 floats is 8 consecutive lanes and a wavefront moves
// 8 whole rows per instruction (4 rows at 64 floats).  The id of a slot is read by all lanes of that
// slot (same address -> one broadcast fetch).  The 4-byte feature_bias rows are gathered by a
// separate slot-per-lane grid tail so that they do not put a divergent scalar load in the row path.
#include "internal.hpp"

template <int UNROLL, bool NT_ST, bool NT_LD>
__global__ __launch_bounds__(256) void gather_rows_kernel(
    const float* __restrict__ inner, const float* __restrict__ outer, const float* __restrict__ fbias,
    const int32_t* __restrict__ ids, int64_t n_slots, int K4, int D4,
    float* __restrict__ Ei, float* __restrict__ Eo, float* __restrict__ fb, int M,
    unsigned long long* __restrict__ keys) {
    const int CH = K4 + D4;                       // 16-byte chunks per slot across both tables
    const int64_t total = n_slots * CH;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t g0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // UNROLL independent row pieces in flight per lane before the first store
    for (; g0 < total; g0 += stride * UNROLL) {
        f32x4 v[UNROLL];
        int64_t dst[UNROLL];
        int which[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t g = g0 + u * stride;
            which[u] = -1;
            if (g < total) {
                const int64_t slot = g / CH;
                const int ch = (int)(g - slot * CH);
                int id = ids[slot];
                id = id < 0 ? 0 : (id >= M ? M - 1 : id);   // clamp: a bad id must not fault the GPU
                if (ch < K4) {
                    which[u] = 0;
                    dst[u] = slot * K4 + ch;
                    const f32x4* sp = reinterpret_cast<const f32x4*>(inner) + (int64_t)id * K4 + ch;
                    v[u] = NT_LD ? __builtin_nontemporal_load(sp) : *sp;
                } else {
                    which[u] = 1;
                    dst[u] = slot * D4 + (ch - K4);
                    const f32x4* sp = reinterpret_cast<const f32x4*>(outer) + (int64_t)id * D4 + (ch - K4);
                    v[u] = NT_LD ? __builtin_nontemporal_load(sp) : *sp;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (which[u] >= 0) {
                f32x4* dp = reinterpret_cast<f32x4*>(which[u] == 0 ? Ei : Eo) + dst[u];
                if (NT_ST) __builtin_nontemporal_store(v[u], dp); else *dp = v[u];
            }
        }
    }
    if (fb != nullptr || keys != nullptr) {
        for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += stride) {
            const int raw = ids[s];
            const int id = raw < 0 ? 0 : (raw >= M ? M - 1 : raw);
            if (fb != nullptr) fb[s] = fbias[id];
            if (keys != nullptr) keys[s] = ((unsigned long long)(unsigned)((raw < 0 || raw >= M) ? M : raw) << 32) | (unsigned long long)s;   // bad id -> key M
        }
    }
}

extern "C" int cffm_gather(const cffm_shape_t* s, const cffm_tables_t* t, const int32_t* ids, int32_t B,
                           float* Ei, float* Eo, float* fb, void* stream) {
    return cffm_gather_impl(s, t, ids, B, Ei, Eo, fb, nullptr, (hipStream_t)stream);
}

int cffm_gather_impl(const cffm_shape_t* s, const cffm_tables_t* t, const int32_t* ids, int32_t B, float* Ei, float* Eo,
                     float* fb, unsigned long long* keys, hipStream_t stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (B <= 0) return 0;
    if ((s->D & 3) || (s->K & 3)) return CFFM_ERR_BAD_SHAPE;
    const int64_t n_slots = (int64_t)B * s->F;
    const int K4 = Ei ? s->K / 4 : 0, D4 = Eo ? s->D / 4 : 0;
    const int64_t total = n_slots * (K4 + D4);
    int64_t work = total > n_slots ? total : n_slots;
    int blocks = (int)((work + 256 * 4 - 1) / (256 * 4));
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 32) blocks = 256 * 32;
    // measured on MI355X (1M-row tables, 262144 lookups): plain loads/stores, 4 pieces in flight per lane and a
    // grid of up to 8192 workgroups is the fastest of {nt loads, nt stores, 8 pieces per lane, 2048 workgroups}
    hipLaunchKernelGGL((gather_rows_kernel<4, false, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       t->inner_emb, t->outer_emb, t->feat_bias, ids, n_slots, K4, D4, Ei, Eo, fb, s->M, keys);
    CFFM_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(256) void pack_rows_kernel(const int32_t* __restrict__ ids, int64_t n_slots, int K, int D,
                                                        const float* __restrict__ dEi, const float* __restrict__ dEo,
                                                        const float* __restrict__ dfb, const float* __restrict__ scalars,
                                                        float* __restrict__ sum_dst, float* __restrict__ rows) {
    const int W = 1 + K + D + 1;
    const int64_t total = n_slots * W;
    if (blockIdx.x == 0 && threadIdx.x == 0) sum_dst[0] = scalars[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t slot = i / W;
        const int c = (int)(i - slot * W);
        float v;
        if (c == 0) v = __int_as_float(ids[slot]);
        else if (c <= K) v = dEi ? dEi[slot * K + (c - 1)] : 0.f;       // disabled branch: zeros
        else if (c <= K + D) v = dEo ? dEo[slot * D + (c - 1 - K)] : 0.f;
        else v = dfb[slot];
        rows[i] = v;
    }
}

int cffm_pack_rows(const cffm_shape_t* s, const int32_t* ids, int32_t B, const float* dEi, const float* dEo, const float* dfb,
                   const float* scalars, float* sum_dst, float* rows, hipStream_t st) {
    const int64_t n_slots = (int64_t)B * s->F;
    const int64_t total = n_slots * (1 + s->K + s->D + 1);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_rows_kernel, dim3(blocks), dim3(256), 0, st, ids, n_slots, s->K, s->D, dEi, dEo, dfb, scalars,
                       sum_dst, rows);
    CFFM_CHECK_LAUNCH();
    return 0;
}

// ---- row-sharded tables: the owner side of a lookup, the requester side of its answer, and the duplicate-summed
// ---- gradient message (cffm_amd/dist.py ShardedStep; SURVEY 8e collectives 1-3) -----------------------------------
// A looked-up row travels as ONE packed record of cffm_packed_row_floats() = K + D + 4 floats: (inner row | outer row |
// feature_bias, 0, 0, 0) - 16-byte aligned, so both ends move it in 16-byte pieces and the 4-byte feature_bias row rides
// in the same lanes instead of a scalar gather of its own.
extern "C" int32_t cffm_packed_row_floats(const cffm_shape_t* s) { return s ? s->K + s->D + 4 : 0; }

// g -> (g / CH, g % CH) without the 64-bit integer division sequence, for 0 <= g < 2^31 and 1 <= CH <= 2^12.  inv_ch must be
// the DOUBLE reciprocal 1.0 / CH (relative error 2^-53): the product (g + 0.5) * inv_ch is then within 2^-21 of the true
// quotient, which is itself at least 0.5 / CH >= 2^-13 away from every integer, so the truncation is exact; the +-1 correction
// is a belt on top.  (A float reciprocal widened to double has a relative error of 6e-8 and is wrong by more than 1 near 2^30:
// tests/test_abi.py::test_div_chunks_guard_boundary replays this arithmetic on the host.)
__host__ __device__ __forceinline__ void div_chunks(int64_t g, int CH, double inv_ch, int64_t* slot, int* ch) {
    int64_t q = (int64_t)(((double)g + 0.5) * inv_ch);
    int r = (int)(g - q * CH);
    if (r < 0) { --q; r += CH; } else if (r >= CH) { ++q; r -= CH; }
    *slot = q; *ch = r;
}

template <int UNROLL>
__global__ __launch_bounds__(256) void gather_packed_kernel(const float* __restrict__ inner, const float* __restrict__ outer,
                                                            const float* __restrict__ fbias, const int32_t* __restrict__ rows,
                                                            int64_t n, int K4, int D4, int M, float* __restrict__ out) {
    const int CH = K4 + D4 + 1;
    const double inv_ch = 1.0 / (double)CH;                      // a TRUE double reciprocal: see div_chunks()
    const int64_t total = n * CH, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g0 < total; g0 += stride * UNROLL) {
        f32x4 v[UNROLL];
        bool ok[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {                       // UNROLL independent 16-byte pieces in flight per lane
            const int64_t g = g0 + u * stride;
            ok[u] = g < total;
            if (ok[u]) {
                int64_t slot; int ch;
                div_chunks(g, CH, inv_ch, &slot, &ch);
                int id = rows[slot];
                id = id < 0 ? 0 : (id >= M ? M - 1 : id);        // clamp: a bad id must not fault the GPU
                if (ch < K4) v[u] = *(reinterpret_cast<const f32x4*>(inner) + (int64_t)id * K4 + ch);
                else if (ch < K4 + D4) v[u] = *(reinterpret_cast<const f32x4*>(outer) + (int64_t)id * D4 + (ch - K4));
                else v[u] = (f32x4){fbias[id], 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            if (ok[u]) *(reinterpret_cast<f32x4*>(out) + g0 + u * stride) = v[u];
    }
}

extern "C" int cffm_gather_packed(const cffm_shape_t* s, const cffm_tables_t* t, const int32_t* rows, int64_t n, float* out,
                                  void* stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (n <= 0) return 0;
    if (!t || !rows || !out || (s->D & 3) || (s->K & 3)) return CFFM_ERR_BAD_SHAPE;
    const int K4 = s->K / 4, D4 = s->D / 4;
    const int64_t total = n * (K4 + D4 + 1);
    if (total >= (1ll << 31)) return CFFM_ERR_BAD_SHAPE;
    int blocks = (int)((total + 256 * 4 - 1) / (256 * 4));
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL((gather_packed_kernel<4>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, t->inner_emb, t->outer_emb,
                       t->feat_bias, rows, n, K4, D4, s->M, out);
    CFFM_CHECK_LAUNCH();
    return 0;
}

// requester side: slot i of the batch takes packed record pos[i] (duplicates of one id share a record) -> ws.Ei / ws.Eo / ws.fb
__global__ __launch_bounds__(256) void stage_packed_kernel(const float* __restrict__ packed, const int32_t* __restrict__ pos,
                                                           int64_t n_slots, int64_t n_records, int K4, int D4,
                                                           float* __restrict__ Ei, float* __restrict__ Eo, float* __restrict__ fb) {
    const int CH = K4 + D4 + 1;
    const double inv_ch = 1.0 / (double)CH;
    const int64_t total = n_slots * CH, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        int64_t slot; int ch;
        div_chunks(g, CH, inv_ch, &slot, &ch);
        int64_t r = pos ? (int64_t)pos[slot] : slot;
        r = r < 0 ? 0 : (r >= n_records ? n_records - 1 : r);
        const f32x4 v = *(reinterpret_cast<const f32x4*>(packed) + r * CH + ch);
        if (ch < K4) { if (Ei) *(reinterpret_cast<f32x4*>(Ei) + slot * K4 + ch) = v; }
        else if (ch < K4 + D4) { if (Eo) *(reinterpret_cast<f32x4*>(Eo) + slot * D4 + (ch - K4)) = v; }
        else fb[slot] = v[0];
    }
}

extern "C" int cffm_stage_packed(const cffm_shape_t* s, const float* packed, const int32_t* pos, int64_t n_records, int32_t B,
                                 void* ws, void* stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (B <= 0) return 0;
    if (!packed || !ws || n_records <= 0 || (s->D & 3) || (s->K & 3)) return CFFM_ERR_BAD_SHAPE;
    cffm_ws_layout_t wl;
    cffm_ws_layout(s, B, &wl);
    char* w = (char*)ws;
    const int K4 = s->K / 4, D4 = s->D / 4;
    const int64_t n_slots = (int64_t)B * s->F, total = n_slots * (K4 + D4 + 1);
    if (total >= (1ll << 31)) return CFFM_ERR_BAD_SHAPE;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(stage_packed_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, packed, pos, n_slots, n_records, K4,
                       D4, s->inner_conv ? (float*)(w + wl.Ei) : nullptr, s->outer_conv ? (float*)(w + wl.Eo) : nullptr,
                       (float*)(w + wl.fb));
    CFFM_CHECK_LAUNCH();
    return 0;
}

// Gradient message of a row-sharded step with the duplicates of one id summed BEFORE the exchange ("all-to-all
// of deduplicated ids").  order[q] = slot at sorted position q (sorted by (owner, local row), stable: slots ascend inside
// a segment), uniq[q] = index of that position's distinct id; a wavefront per segment head sums its segment in slot order
// (bitwise reproducible) into record uniq[q] of out: (local row bits | dEi | dEo | dfb), the row format cffm_dp_apply takes.
__global__ __launch_bounds__(256) void pack_rows_dedup_kernel(const int32_t* __restrict__ local_ids, const int32_t* __restrict__ order,
                                                              const int32_t* __restrict__ uniq, int64_t n, int K, int D,
                                                              const float* __restrict__ dEi, const float* __restrict__ dEo,
                                                              const float* __restrict__ dfb, float* __restrict__ out) {
    const int64_t q0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q0 >= n) return;
    const int u = uniq[q0];
    if (q0 > 0 && uniq[q0 - 1] == u) return;                     // not a segment head
    const int W = 1 + K + D + 1;
    float* o = out + (int64_t)u * W;
    for (int c0 = 0; c0 < W; c0 += 64) {
        const int c = c0 + lane;
        if (c >= W) continue;
        if (c == 0) { o[0] = __int_as_float(local_ids[order[q0]]); continue; }
        float g = 0.f;
        for (int64_t q = q0; q < n && uniq[q] == u; ++q) {
            const int64_t sl = order[q];
            g += c <= K ? (dEi ? dEi[sl * K + (c - 1)] : 0.f) : (c <= K + D ? (dEo ? dEo[sl * D + (c - 1 - K)] : 0.f) : dfb[sl]);
        }
        o[c] = g;
    }
}

extern "C" int cffm_pack_rows_dedup(const cffm_shape_t* s, const int32_t* local_ids, const int32_t* order, const int32_t* uniq,
                                    int32_t B, void* ws, float* out, void* stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (B <= 0) return 0;
    if (!local_ids || !order || !uniq || !ws || !out) return CFFM_ERR_BAD_SHAPE;
    cffm_ws_layout_t wl;
    cffm_ws_layout(s, B, &wl);
    char* w = (char*)ws;
    const int64_t n = (int64_t)B * s->F;
    hipLaunchKernelGGL(pack_rows_dedup_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, local_ids, order,
                       uniq, n, s->K, s->D, s->inner_conv ? (const float*)(w + wl.dEi) : nullptr,
                       s->outer_conv ? (const float*)(w + wl.dEo) : nullptr, (const float*)(w + wl.dfb), out);
    CFFM_CHECK_LAUNCH();
    return 0;
}
