// Inner (element-wise product) branch of the CFFM graph, CFFM.py:301-343, and its gradient.
//
// Per example: P pairwise products of K-vectors -> act -> 1x2/stride-2 conv (1 -> 2 channels) + bias
// -> relu -> act -> + max-pool(2) of the activated products -> flatten (p, t, ch) -> dense(1).
// Everything is element-wise followed by ONE dot product, so the whole branch is a single pass over
// the gathered rows held in LDS with a wavefront/block reduction at the end: nothing of the
// [B,P,K/2,2] intermediate (1 GB at F=32,K=64,B=8192) is ever written.
#include "internal.hpp"

#include "inner_body.hpp"

#include <stdlib.h>

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
    return cffm_inner_bwd_rows(s, theta, ws, B, nullptr, (hipStream_t)stream);
}

int cffm_inner_bwd_wide(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, const RowSrc* rs, hipStream_t stream);

int cffm_inner_bwd_rows(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, const RowSrc* rs, hipStream_t stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (B <= 0 || !s->inner_conv) return 0;
    if (cffm_wide_regather_ok(s) && !getenv("CFFM_INNER_BWD_V1")) return cffm_inner_bwd_wide(s, theta, ws, B, rs, stream);
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
    if (rs) { a.Ei = rs->base; a.idx = rs->idx; a.idxM = rs->M; a.idxStride = rs->stride; }
    hipLaunchKernelGGL(inner_bwd_kernel, dim3(nslab), dim3(256), lds, (hipStream_t)stream, a);
    CFFM_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Wide shapes (Pp > 64; BASELINE configs[3]/[4]: F32 K64 D64): tf.nn.embedding_lookup x3 (CFFM.py:303, :354, :422) fused
// with everything that consumes a whole looked-up example in one pass - the inner branch (:304-343), the s0 sum pool of the
// outer-product map (:381, in closed form) and the first-order inputs (:422) - so that the rows go HBM -> LDS -> registers
// and are never written back (the materialising gather of rounds 1-2 wrote as many bytes as it read).
//
// One workgroup of 1024 threads per CU.  Examples are taken E = 4 at a time ("phase"); the rows of the next phase (E x F x
// (K + D) floats = 64 KB at F32 K64 D64) are in flight - global_load_dwordx4 into registers, 16 lanes = one 256-byte row,
// ids fetched a phase earlier - while the current phase is computed out of LDS, so 64 KB per CU stay in flight behind the
// VALU work (tools/probe_gather.hip: this access shape alone reads at 0.73 of 8 TB/s on this box).
//
// Inner branch: thread (g, t) = (tid / K2, tid % K2) owns the units (p, t) of pairs p = g*UPT .. g*UPT+UPT-1 for EVERY
// example, so its 2 x UPT dense(1) weights and the LDS offsets of its rows live in registers for the whole kernel (the old
// kernel re-read the 127 KB weight vector from L2 for every example).  Per unit: ds_read_b64 x2, then 10 VALU instructions,
// 4 of them packed (v_pk_mul / v_pk_fma / v_pk_add on the (k, k+1) / (ch0, ch1) pairs).
// The kernel is VALU-bound at this shape, not HBM-bound: 8192 x 496 x 32 units x 10 instructions = 1.3 G lane-instructions
// against the ~46 T/s this box issues (tools/probe_gather.hip, v_fma_f32 on every SIMD) = 28 us, the 136 MB at the
// measured gather rate 23 us.
//
// LDS image of one example: [F][K] inner rows, [F][D] outer rows = the pieces in fetch order (piece q at byte 16*q).
// ---------------------------------------------------------------------------------------------------------------------
#define GIW_E 4            // examples per phase
#define GIW_T 1024         // threads per workgroup

struct GatherInnerWideArgs {
    const float *inner, *outer, *fbias;     // tables [M][K], [M][D], [M]
    const int32_t* ids;                     // [B][F]
    const float *cw, *cb, *wd, *bd;         // inner-branch parameters (theta)
    float *inner_out, *t1, *fb;             // [B], [B][t1w] (columns 0..D-1 = s0), [B][F]
    unsigned long long* keys;               // [B*F] packed (id << 32 | slot) for the sparse update (may be NULL)
    int B, M, F, K, D, P, t1w, act;
    int row4_in, row4_out, fb_stride;       // 16-byte pieces between rows of inner / outer, floats between rows of fbias (natural: K/4, D/4, 1;
                                            // the row-sharded step passes its packed records: (K+D+4)/4 twice and K+D+4)
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// x + |x| = 2 relu(x) in ONE v_add_f32 (abs source modifier).  As inline assembly: written in C the SLP vectoriser pairs two of them
// into v_and_b32 x2 + v_pk_add_f32, three instructions for what two do.
__device__ __forceinline__ float relu2(float x) {
    float r;
    asm("v_add_f32_e64 %0, %1, |%1|" : "=v"(r) : "v"(x));
    return r;
}

// v_max_f32 as is: fmaxf() on the outputs of the assembly above gets two canonicalising v_max_f32 x, x in front (IEEE mode)
__device__ __forceinline__ float max_raw(float a, float b) {
    float r;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// four wavefront sums as interleaved DPP chains (no hazard s_nops between the steps); every lane ends with the four totals
__device__ __forceinline__ void wave_sum4(float (&v)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += dpp_mov<0xB1, 0xf>(v[e], 0.f);     // quad_perm [1,0,3,2]
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += dpp_mov<0x4E, 0xf>(v[e], 0.f);     // quad_perm [2,3,0,1]
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += dpp_mov<0x141, 0xf>(v[e], 0.f);    // row_half_mirror
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += dpp_mov<0x140, 0xf>(v[e], 0.f);    // row_mirror
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += dpp_mov<0x142, 0xa>(v[e], 0.f);    // row_bcast:15
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += dpp_mov<0x143, 0xc>(v[e], 0.f);    // row_bcast:31 -> lane 63 holds the total
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[e]), 63));
}

// FS > 0: the field count as a compile-time constant (with K == D == 2*K2 this fixes the LDS image: every per-example offset of
// the unit loop becomes an immediate of its ds_read_b64 instead of a v_add per read)
//
// CIRC (round 4; the F = 32, K = D = 64 instance of BASELINE configs[3] / [4]): the pairs are dealt to the threads as a
// CIRCULANT - half-wavefront (wave w, half h) owns row i = w + 16 h and the pairs {i, (i + d) & 31}, d = 1 .. 16 (d = 16 only
// for h = 0: the 16 diameters) - instead of 16 consecutive pairs of the row-major list.  What that buys per unit:
//   * the i-row piece is read from LDS once per example, not once per unit: 17 ds_read_b64 per 16 units instead of 32;
//   * the j-row address is S(d) ^ V with S(d) = ((w + d) & 31) << 8 | buffer << 16 a SCALAR and V = h << 12 | 8 t one VGPR:
//     one v_xor per (unit, FOUR examples) and no offset table in registers (16 VGPRs back);
// NOT taken (measured, tools/probe_mfma_unit.hip, profiles/r04_probe_mfma_unit.txt): the 1x2 conv [x0, x1, 1] -> [z0, z1] (CFFM.py:327)
// on the MFMA pipe.  v_mfma_f32_4x4x1_16b_f32 has exactly the right layout (16 blocks of D[m][n] = C[m][n] + A[m] * B[n], B[n] from
// lane 4b+n = the lane's OWN x, D[.][n] back in that lane: confirmed on the box), but an MFMA holds the SIMD's vector issue port
// for its whole 8 cycles: 8 VALU + 2 MFMA run at 29.4 ns per unit-wave per SIMD against 23.0 for the 10 VALU instructions (17.2
// for the 8 alone) at this kernel's 4 waves per SIMD.  The pipe is idle, the issue port is not.
template <int K2, int UPT, int ACTC, int FS, bool CIRC = false>
__global__ __launch_bounds__(GIW_T) void gather_inner_fwd_wide_kernel(GatherInnerWideArgs a) {
    constexpr int E = GIW_E, T = GIW_T, K = 2 * K2, NG = T / K2;
    static_assert(!CIRC || (K2 == 32 && UPT == 16 && FS == 32), "the circulant instance is F = 32, K = D = 64");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int F = FS > 0 ? FS : a.F, D = FS > 0 ? K : a.D, P = FS > 0 ? FS * (FS - 1) / 2 : a.P, act = ACTC >= 0 ? ACTC : a.act;
    const int K4 = K >> 2, D4 = D >> 2;
    const int npiece = F * (K4 + D4);                        // 16-byte pieces of one example (<= T)
    const int slot_bytes = npiece * 16;                      // LDS image of one example (<= 16 KB)
    constexpr int buf_bytes = 65536;                         // the two row buffers sit at LDS addresses 0 and 64 KB (see the unit loop)
    char* rows = smem;                                       // [2 buffers][E][slot]
    float* RS = reinterpret_cast<float*>(smem + 2 * buf_bytes);   // [2][E][32] row sums of the outer rows
    float* red = RS + 2 * E * 32;                            // [2][16 waves][E] inner_out partials
    uint32_t* lut = reinterpret_cast<uint32_t*>(red + 2 * 16 * E);   // [NG * UPT] pair -> (i | j << 16)
    float* fbL = reinterpret_cast<float*>(lut + NG * UPT);           // [2][128] feature_bias of the slots of a phase (landing zone of the DMA)
    int* idsL = reinterpret_cast<int*>(fbL + 2 * 128);               // [2][128] raw ids of the slots of a phase
    f32x2* S0P = reinterpret_cast<f32x2*>(idsL + 2 * 128);           // CIRC: [2][4 quarters][E][64] (local s0 partial, column sum)
    float* S0Q = reinterpret_cast<float*>(S0P + 2 * 4 * E * 64);     // CIRC: [2][4][E] row-sum totals of the quarters
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: LDS bases (M0 of the row DMA) and the quarter / example of a wave cost no VALU
    const int nphase = (a.B + E - 1) / E;

    // ---- fetch role of this thread: piece q = tid of each of the E examples of a phase ------------------------------
    const bool fetch = tid < npiece;
    const bool is_in = tid < F * K4;
    const int fq = is_in ? tid / K4 : (tid - F * K4) / D4;                // field of the piece
    const int cq = is_in ? tid - fq * K4 : (tid - F * K4) - fq * D4;      // 16-byte piece inside the row
    const uint32_t rowlen4 = (uint32_t)(is_in ? a.row4_in : a.row4_out);
    const f32x4* tbl = reinterpret_cast<const f32x4*>(is_in ? a.inner : a.outer) + cq;   // this thread's piece of row 0
    // The ids of a phase travel through LDS as well: thread tid < E*F owns slot (e, f) = (tid / F, tid % F) and fetches its id by a
    // 4-byte global_load_lds one phase before the rows that need it are requested; the fetch threads read the ids of their four
    // rows from there.  No id, no feature_bias value and no pointer to them is held in a register across the unit loop: with
    // them the kernel spilled, and a spill reload is a vector-memory operation - its vmcnt wait also waits for every row DMA
    // issued before it, i.e. the waves that reloaded at the top of a phase sat out the whole HBM latency of their rows there and
    // the rest of the workgroup waited for them at the phase barrier.
    const bool fo_on = tid < E * F;                                        // first-order role (:422) + id fetch: waves 0 .. E*F/64 - 1
    const int fo_e = fo_on ? tid / F : 0, fo_f = fo_on ? tid - fo_e * F : 0;
    const bool wave_has_outer = (wave + 1) * 64 > F * K4 && wave * 64 < npiece;      // some lane of this wave fetches an outer piece
    auto clampid = [&](int id) { return (uint32_t)max(0, min(id, a.M - 1)); };          // v_med3_i32: a bad id must not fault the GPU
    auto load_ids = [&](int ph, int ib) {                                  // ids of phase ph -> idsL[ib]
        if (fo_on) {
            int b = ph * E + fo_e;
            b = b < a.B ? b : a.B - 1;                                     // tail: a harmless duplicate
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(a.ids + (int64_t)b * F + fo_f),
                                             (void __attribute__((address_space(3)))*)(idsL + ib * 128 + wave * 64), 4, 0, 0);
        }
    };
    // rows go HBM -> LDS directly (global_load_lds_dwordx4: the wave's 64 pieces land as 1 KB at a wave-uniform LDS base +
    // 16 * lane, which IS the image layout), so no staging registers are held across the compute phase
    auto load_rows = [&](int buf, int ib) {                                // rows (and feature_bias) of the phase whose ids are in idsL[ib]
        if (fetch) {
            int idv[E];
#pragma unroll
            for (int e = 0; e < E; ++e) idv[e] = idsL[ib * 128 + e * F + fq];
#pragma unroll
            for (int e = 0; e < E; ++e)
                __builtin_amdgcn_global_load_lds(
                    (const void __attribute__((address_space(1)))*)(tbl + (uint64_t)clampid(idv[e]) * rowlen4),
                    (void __attribute__((address_space(3)))*)(rows + buf * buf_bytes + e * slot_bytes + wave * 1024), 16, 0, 0);
        }
        if (fo_on)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(a.fbias + (uint64_t)clampid(idsL[ib * 128 + tid]) * (uint32_t)a.fb_stride),
                                             (void __attribute__((address_space(3)))*)(fbL + buf * 128 + wave * 64), 4, 0, 0);
    };
    // once the wave's own loads have landed (vmcnt(0)): row sums of the outer rows for the s0 pool - D4 lanes hold one row
    auto row_sums = [&](int buf, int phn, int ib) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (fo_on) {                                                       // first-order inputs + sort keys of phase phn
            const int b = phn * E + fo_e;
            if (b < a.B) {
                const int64_t slot = (int64_t)b * F + fo_f;
                const int raw = idsL[ib * 128 + tid];
                a.fb[slot] = fbL[buf * 128 + tid];                         // this wave's own DMA: visible behind the vmcnt(0) above
                if (a.keys) a.keys[slot] = ((unsigned long long)(unsigned)((raw < 0 || raw >= a.M) ? a.M : raw) << 32) | (unsigned long long)slot;   // bad id -> key M
            }
        }
        if (wave_has_outer) {                                              // wave-uniform: the waves that hold only inner pieces skip
#pragma unroll
            for (int e = 0; e < E; ++e) {
                f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (fetch && !is_in) v = *reinterpret_cast<const f32x4*>(rows + buf * buf_bytes + e * slot_bytes + tid * 16);
                float s = (v.x + v.y) + (v.z + v.w);
                s = D4 == 16 ? row_group_sum<true>(s) : row_group_sum<false>(s);
                if (fetch && !is_in && cq == 0) RS[(buf * E + e) * 32 + fq] = s;
            }
        }
    };

    // the ids of this workgroup's first phase are requested before anything else (the pair table below is built in their shadow)
    if ((int)blockIdx.x < nphase) load_ids(blockIdx.x, 0);

    // ---- per-thread constants: its units' weights and LDS offsets ----------------------------------------------------
    if (!CIRC) {
        for (int i = tid; i < F; i += T) {
            const int base = i * (2 * F - i - 1) / 2;
            for (int j = i + 1; j < F; ++j) lut[base + j - i - 1] = (uint32_t)i | ((uint32_t)j << 16);
        }
        for (int p = P + tid; p < NG * UPT; p += T) lut[p] = 0u;
    }
    const int g = tid / K2, t = tid - g * K2;
    f32x2 w[UPT];
    uint32_t off[CIRC ? 1 : UPT];                            // LDS byte offsets of the unit's two rows inside an example image: i-row | j-row << 16
    const f32x2* wd2 = reinterpret_cast<const f32x2*>(a.wd);
    const int wave_s = wave;                                  // scalar: the j-row addresses of the circulant are S(d) ^ V
    const int half = lane >> 5, irow = wave + 16 * half;      // CIRC: this thread's row i
#pragma unroll
    for (int k = 0; k < UPT; ++k) {
        if (CIRC) {
            const int d = k + 1, j = (irow + d) & 31;
            const int lo = irow < j ? irow : j, hi = irow < j ? j : irow;
            const int p = lo * (2 * 32 - lo - 1) / 2 + (hi - lo - 1);
            w[k] = (d < 16 || half == 0) ? wd2[(int64_t)p * K2 + t] : (f32x2){0.f, 0.f};   // the 16 diameters belong to the lower row
            if (ACTC == CFFM_ACT_RELU) w[k] = w[k] * 0.5f;                                  // the relu build works on 2 (c + mp), see compute()
        } else {
            const int p = g * UPT + k;
            w[k] = p < P ? wd2[(int64_t)p * K2 + t] : (f32x2){0.f, 0.f};      // flat index p*K + 2t + ch (:333)
        }
    }
    const f32x2 w0 = (f32x2){a.cw[0], a.cw[1]}, w1 = (f32x2){a.cw[2], a.cw[3]}, cb2 = (f32x2){a.cb[0], a.cb[1]};
    const f32x2 w0h = w0 * 0.5f, w1h = w1 * 0.5f;                                 // CIRC relu build: taps of the doubled activations
    const uint32_t Vx = ((uint32_t)half << 12) | (uint32_t)(8 * (lane & 31));     // CIRC: per-lane part of a j-row address
    const uint32_t ai0 = ((uint32_t)irow << 8) | (uint32_t)(8 * (lane & 31));      // CIRC: the i-row piece inside an example image
    const float bd = a.bd[0];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the first phase's ids (and the weights above) have landed
    __syncthreads();
    if ((int)blockIdx.x < nphase) {                                       // rows of the first phase, ids of the second
        load_rows(0, 0);
        if ((int)(blockIdx.x + gridDim.x) < nphase) load_ids(blockIdx.x + gridDim.x, 1);
    }
    if (!CIRC) {
#pragma unroll
        for (int k = 0; k < UPT; ++k) {
            const uint32_t ij = lut[g * UPT + k];                             // padded pairs: (0, 0) with zero weights
            off[k] = (uint32_t)(((int)(ij & 0xffff) * K + 2 * t) * 4) | ((uint32_t)(((int)(ij >> 16) * K + 2 * t) * 4) << 16);
        }
    }
    // the unit loop forms LDS addresses as (offset | buffer << 16): dynamic LDS starts at address 0 in a kernel without static
    // __shared__ variables.  The HOST checks that (giw_lds_base_ok: hipFuncGetAttributes().sharedSizeBytes == 0 for every instance,
    // else cffm_wide_regather_ok() is false and the materialising path runs); the device-side trap is a debug build's.
#ifdef CFFM_TILE_DBG
    if ((uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
#endif

    int ph = blockIdx.x;
    if (ph < nphase) row_sums(0, ph, 0);
    __syncthreads();
    int par = 0;                                               // phase ph: rows in rows[par], ids in idsL[par]
    for (; ph < nphase; ph += gridDim.x, par ^= 1) {
        const int nxt = ph + gridDim.x;
        const bool more = nxt < nphase;
        if (more) {
            load_rows(par ^ 1, par ^ 1);                                   // next phase's rows: in flight across the compute below
            if (nxt + (int)gridDim.x < nphase) load_ids(nxt + gridDim.x, par);   // idsL[par] (this phase's ids) is free: its keys are out
        }
        // ---- inner branch of the E examples of this phase ------------------------------------------------------------
        const char* buf = rows + par * buf_bytes;
        f32x2 acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = (f32x2){0.f, 0.f};
        // The unit loop as a rolling pipeline over half steps (two examples each): while one half is computed, the LDS reads
        // of the next one are in flight, so a wave's own reads hide behind its own arithmetic (with the four reads of a step
        // issued and awaited together, 40 % of the SIMD cycles had no VALU instruction to issue: SQ_ACTIVE_INST_VALU,
        // profiles/r03_gather_pmc.md).  The accumulators are pinned after every half step and nothing is scheduled across
        // the pins: left alone, the compiler hoists the reads of ALL units to the top of the phase and sinks the arithmetic
        // below them (267 spilled registers under the 128-register budget of a 1024-thread workgroup).
        typedef const f32x2 __attribute__((address_space(3))) * lds2_t;
        static_assert(E == 4, "two half steps of two examples");
        if constexpr (CIRC) {
            // Steps of UB units x ONE example, examples innermost: step s = (batch s / E, example s % E).  The reads of step
            // s + 1 (UB j-row pieces + the i-row piece, imm offset = example) are issued before step s is computed, into the
            // other half of a register double buffer; the UB addresses of a batch are built once for its four examples.
            constexpr int UB = 4, NB = UPT / UB, NS = NB * E;
            f32x2 ejv[2][UB], eiv[2];
            uint32_t ajv[2][UB];
            const uint32_t pbit = (uint32_t)par << 16;
            const uint32_t ai = ai0 | pbit;
            auto mkaddr = [&](int kb, uint32_t* a4) {
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const uint32_t S = ((uint32_t)((wave_s + kb * UB + u + 1) & 31) << 8) | pbit;    // scalar
                    a4[u] = S ^ Vx;
                }
            };
            auto issue = [&](int s_) {
                const int kb = s_ / E, e = s_ % E;
#pragma unroll
                for (int u = 0; u < UB; ++u) ejv[s_ & 1][u] = *(lds2_t)(size_t)(ajv[kb & 1][u] + e * slot_bytes);
                eiv[s_ & 1] = *(lds2_t)(size_t)(ai + e * slot_bytes);
            };
            auto compute = [&](int s_) {
                const int kb = s_ / E, e = s_ % E;
                f32x2 x[UB], c[UB], zz[UB];
                float mp[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) x[u] = eiv[s_ & 1] * ejv[s_ & 1][u];                      // :310
                if constexpr (ACTC == CFFM_ACT_RELU) {
                    // relu twice over as x + |x| = 2 relu(x): v_add_f32 with the abs source modifier issues at 1.6 ns per wave
                    // instruction per SIMD, v_max_f32 at 1.95 (profiles/r03_probe_valu.txt).  The factors of two are exact and are
                    // taken back by the halved conv and dense weights (w0h, w1h, wh): (2a)(b/2) rounds exactly like ab, 2a + 2b
                    // like 2(a + b) - the result is bit-identical to the max form.
#pragma unroll
                    for (int u = 0; u < UB; ++u) { x[u].x = relu2(x[u].x); x[u].y = relu2(x[u].y); }   // 2 relu(x) :319
#pragma unroll
                    for (int u = 0; u < UB; ++u) zz[u] = __builtin_elementwise_fma((f32x2){x[u].x, x[u].x}, w0h, cb2);      // :327 (exactly z)
#pragma unroll
                    for (int u = 0; u < UB; ++u) zz[u] = __builtin_elementwise_fma((f32x2){x[u].y, x[u].y}, w1h, zz[u]);
#pragma unroll
                    for (int u = 0; u < UB; ++u) mp[u] = max_raw(x[u].x, x[u].y);                       // 2 maxpool :331
#pragma unroll
                    for (int u = 0; u < UB; ++u) { c[u].x = relu2(zz[u].x); c[u].y = relu2(zz[u].y); }   // 2 relu(z) :478, :330
#pragma unroll
                    for (int u = 0; u < UB; ++u) c[u] = c[u] + (f32x2){mp[u], mp[u]};                   // 2 (c + mp) :332
#pragma unroll
                    for (int u = 0; u < UB; ++u) acc[e] = __builtin_elementwise_fma(c[u], w[kb * UB + u], acc[e]);   // w holds wd / 2 :339
                } else {
#pragma unroll
                    for (int u = 0; u < UB; ++u) { x[u].x = act_f(x[u].x, act); x[u].y = act_f(x[u].y, act); }   // :319
#pragma unroll
                    for (int u = 0; u < UB; ++u) zz[u] = __builtin_elementwise_fma((f32x2){x[u].x, x[u].x}, w0, cb2);       // :327  cw[tap*2+ch]
#pragma unroll
                    for (int u = 0; u < UB; ++u) zz[u] = __builtin_elementwise_fma((f32x2){x[u].y, x[u].y}, w1, zz[u]);
#pragma unroll
                    for (int u = 0; u < UB; ++u) mp[u] = fmaxf(x[u].x, x[u].y);                             // :331
#pragma unroll
                    for (int u = 0; u < UB; ++u) { c[u].x = act_pos(fmaxf(zz[u].x, 0.f), act); c[u].y = act_pos(fmaxf(zz[u].y, 0.f), act); }   // :478, :330
#pragma unroll
                    for (int u = 0; u < UB; ++u) c[u] = c[u] + (f32x2){mp[u], mp[u]};                       // :332
#pragma unroll
                    for (int u = 0; u < UB; ++u) acc[e] = __builtin_elementwise_fma(c[u], w[kb * UB + u], acc[e]);   // :339
                }
            };
            mkaddr(0, ajv[0]);
            issue(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) {
                if (s_ + 1 < NS) {
                    if ((s_ + 1) % E == 0) mkaddr((s_ + 1) / E, ajv[((s_ + 1) / E) & 1]);
                    issue(s_ + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                compute(s_);
                asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            f32x2 eiA[2], ejA[2], eiB[2], ejB[2];
            auto issue = [&](int k, int half, f32x2* ei, f32x2* ej) {
                // two instructions per address pair: v_and_or_b32 and v_alignbit_b32 put the buffer bit on top of the 16-bit offsets
                const uint32_t ai = (off[k] & 0xffffu) | ((uint32_t)par << 16);
                const uint32_t aj = __builtin_amdgcn_alignbit((uint32_t)par, off[k], 16);
    #pragma unroll
                for (int e = 0; e < 2; ++e) {
                    ei[e] = *(lds2_t)(size_t)(ai + (2 * half + e) * slot_bytes);
                    ej[e] = *(lds2_t)(size_t)(aj + (2 * half + e) * slot_bytes);
                }
            };
            auto compute = [&](int k, int half, const f32x2* ei, const f32x2* ej) {
                f32x2 x[2], z[2], c[2];
                float mp[2];
    #pragma unroll
                for (int e = 0; e < 2; ++e) x[e] = ei[e] * ej[e];                                   // :310
    #pragma unroll
                for (int e = 0; e < 2; ++e) { x[e].x = act_f(x[e].x, act); x[e].y = act_f(x[e].y, act); }   // :319
    #pragma unroll
                for (int e = 0; e < 2; ++e) z[e] = __builtin_elementwise_fma((f32x2){x[e].x, x[e].x}, w0, cb2);   // :327  cw[tap*2+ch]
    #pragma unroll
                for (int e = 0; e < 2; ++e) z[e] = __builtin_elementwise_fma((f32x2){x[e].y, x[e].y}, w1, z[e]);
    #pragma unroll
                for (int e = 0; e < 2; ++e) mp[e] = fmaxf(x[e].x, x[e].y);                          // :331
    #pragma unroll
                for (int e = 0; e < 2; ++e) { c[e].x = act_pos(fmaxf(z[e].x, 0.f), act); c[e].y = act_pos(fmaxf(z[e].y, 0.f), act); }   // :478, :330
    #pragma unroll
                for (int e = 0; e < 2; ++e) c[e] = c[e] + (f32x2){mp[e], mp[e]};                    // :332
    #pragma unroll
                for (int e = 0; e < 2; ++e) acc[2 * half + e] = __builtin_elementwise_fma(c[e], w[k], acc[2 * half + e]);   // :339
            };
            issue(0, 0, eiA, ejA);
            issue(0, 1, eiB, ejB);
            __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
            for (int k = 0; k < UPT; ++k) {
                compute(k, 0, eiA, ejA);
                asm volatile("" : "+v"(acc[0]), "+v"(acc[1]) : : "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (k + 1 < UPT) issue(k + 1, 0, eiA, ejA);
                __builtin_amdgcn_sched_barrier(0);
                compute(k, 1, eiB, ejB);
                asm volatile("" : "+v"(acc[2]), "+v"(acc[3]) : : "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (k + 1 < UPT) issue(k + 1, 1, eiB, ejB);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (CIRC) {
            // the four wavefront sums as four INTERLEAVED DPP chains (one after the other every step waits out the DPP hazard in
            // s_nops: 80 issue slots for the four, 34 interleaved), one 16-byte store of the four partials
            float sv[E];
#pragma unroll
            for (int e = 0; e < E; ++e) sv[e] = acc[e].x + acc[e].y;
            wave_sum4(sv);
            if (lane == 0) *reinterpret_cast<f32x4*>(red + (par * 16 + wave) * E) = (f32x4){sv[0], sv[1], sv[2], sv[3]};
            // ---- s0 pool (:381) over ALL sixteen wavefronts: s0[h] = sum_i Eo[i][h] * R_i, R_i = sum_{j>i} rowsum(Eo[j]).  Wave ->
            // (example e = wave & 3, quarter q = wave >> 2) sweeps the rows i = 8q+7 .. 8q with R counted from the top of ITS quarter
            // and keeps the column sum of its rows: s0 = sum_q (local_q + colsum_q * Rtop_q), Rtop_q = the row sums of the quarters
            // above, put together behind the phase barrier.  The serial sweep on wavefronts 0 .. 3 alone was ~130 issue slots and
            // four dependent LDS round trips at the END of their phase, run by one wave per SIMD while the other three waited.
            {
                const int e = wave_s & 3, q = wave_s >> 2, h = lane;
                const float* Eo = reinterpret_cast<const float*>(buf + e * slot_bytes) + 32 * K + (8 * q) * 64 + h;
                const float* rs = RS + (par * E + e) * 32 + 8 * q;
                float ev[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) ev[u] = Eo[(7 - u) * 64];
                const f32x4 rlo = *reinterpret_cast<const f32x4*>(rs), rhi = *reinterpret_cast<const f32x4*>(rs + 4);
                const float rv[8] = {rhi.w, rhi.z, rhi.y, rhi.x, rlo.w, rlo.z, rlo.y, rlo.x};      // rows 8q+7 .. 8q
                float R = 0.f, sl = 0.f, col = 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) { sl += ev[u] * R; col += ev[u]; R += rv[u]; }
                S0P[((par * 4 + q) * E + e) * 64 + h] = (f32x2){sl, col};
                if (lane == 0) S0Q[(par * 4 + q) * E + e] = R;                  // the quarter's row-sum total
            }
        } else {
    #pragma unroll
            for (int e = 0; e < E; ++e) {
                const float v = wave_sum(acc[e].x + acc[e].y);
                if (lane == 0) red[(par * 16 + wave) * E + e] = v;
            }
            // ---- s0 pool (:381): s0[h] = sum_i Eo[i][h] * R_i, R_i = sum_{j>i} rowsum(Eo[j]): wavefront e takes example e, lane = h
            // (the same order of operations as head_fwd_body).  Waves 0..3 sit on four different SIMDs.
            if (wave < E) {
                const int e = wave, h = lane, b = ph * E + e;
                const float* Eo = reinterpret_cast<const float*>(buf + e * slot_bytes) + F * K;
                const float* rs = RS + (par * E + e) * 32;
                const int hh = h < D ? h : 0;
                float s = 0.f, R = 0.f;
                int i = F - 2;
                for (; i >= 7; i -= 8) {                                       // eight rows per step: their sixteen LDS reads are in flight together
                    float ev[8], rv[8];
    #pragma unroll
                    for (int u = 0; u < 8; ++u) { ev[u] = Eo[(i - u) * D + hh]; rv[u] = rs[i - u + 1]; }
    #pragma unroll
                    for (int u = 0; u < 8; ++u) { R += rv[u]; s += ev[u] * R; }
                }
                for (; i >= 0; --i) { R += rs[i + 1]; s += Eo[i * D + hh] * R; }
                if (h < D && b < a.B) a.t1[(int64_t)b * a.t1w + h] = s;
            }
        }
        if (more) row_sums(par ^ 1, nxt, par ^ 1);                                  // waits for the rows that were in flight
        __syncthreads();
        if (CIRC && wave < E) {                                            // s0 of this phase: the four quarters, top rows first
            const int b = ph * E + wave;
            const f32x2* sp = S0P + (par * 4 * E + wave) * 64 + lane;
            const float* qp = S0Q + par * 4 * E + wave;
            float s0v = sp[3 * E * 64].x, Rt = qp[3 * E];
#pragma unroll
            for (int q = 2; q >= 0; --q) {
                const f32x2 v = sp[q * E * 64];
                s0v += v.x + v.y * Rt;
                Rt += qp[q * E];
            }
            if (b < a.B) a.t1[(int64_t)b * a.t1w + lane] = s0v;
        }
        if (tid < E) {                                                     // inner_out of this phase: wavefront partials in wave order
            const int b = ph * E + tid;
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < 16; ++wv) s += red[(par * 16 + wv) * E + tid];
            if (b < a.B) a.inner_out[b] = s + bd;                          // :339
        }
    }
}

template <int K2, int UPT, int ACTC, int FS, bool CIRC = false>
static int launch_giw1(const GatherInnerWideArgs& a, int grid, size_t lds, hipStream_t st) {
    hipError_t e = hipFuncSetAttribute((const void*)gather_inner_fwd_wide_kernel<K2, UPT, ACTC, FS, CIRC>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((gather_inner_fwd_wide_kernel<K2, UPT, ACTC, FS, CIRC>), dim3(grid), dim3(GIW_T), lds, st, a);
    CFFM_CHECK_LAUNCH();
    return 0;
}
template <int K2, int UPT>
static int launch_giw(const GatherInnerWideArgs& a, int grid, size_t lds, hipStream_t st) {
    if (K2 == 32 && UPT == 16 && a.F == 32 && a.D == 64) {          // BASELINE configs[3] / [4]: F32 K64 D64
        static const bool v3 = getenv("CFFM_GIW_V3") != nullptr;     // A/B runs: round 3's row-major pair assignment, 10 VALU per unit
        if (v3) {
            if (a.act == CFFM_ACT_RELU) return launch_giw1<32, 16, CFFM_ACT_RELU, 32>(a, grid, lds, st);
            return launch_giw1<32, 16, -1, 32>(a, grid, lds, st);
        }
        if (a.act == CFFM_ACT_RELU) return launch_giw1<32, 16, CFFM_ACT_RELU, 32, true>(a, grid, lds, st);
        return launch_giw1<32, 16, -1, 32, true>(a, grid, lds, st);
    }
    if (a.act == CFFM_ACT_RELU) return launch_giw1<K2, UPT, CFFM_ACT_RELU, 0>(a, grid, lds, st);
    return launch_giw1<K2, UPT, -1, 0>(a, grid, lds, st);
}

// Every instance forms LDS addresses as (offset | buffer << 16), i.e. assumes that its dynamic LDS starts at LDS address 0 - true
// while the kernel has no static __shared__ variable.  Checked HERE, on the host, once: if a future edit or an instrumented build
// gives any instance static LDS, cffm_wide_regather_ok() turns false and every composite takes the materialising path (cffm_gather
// + the staged kernels) instead of aborting the GPU (the device-side trap is left to the CFFM_TILE_DBG build).
bool cffm_giw_lds_ok() {
    static int cached = -1;
    if (cached >= 0) return cached != 0;
    bool ok = true, asked = false;
    auto chk = [&](const void* f) {
        hipFuncAttributes at;
        if (hipFuncGetAttributes(&at, f) != hipSuccess) { (void)hipGetLastError(); return; }     // no device (layout queries on a CPU box)
        asked = true;
        if (at.sharedSizeBytes != 0) ok = false;
    };
#define GIW_CHK2(K2, UPT) chk((const void*)gather_inner_fwd_wide_kernel<K2, UPT, CFFM_ACT_RELU, 0>); chk((const void*)gather_inner_fwd_wide_kernel<K2, UPT, -1, 0>)
    GIW_CHK2(32, 4); GIW_CHK2(32, 8); GIW_CHK2(32, 16); GIW_CHK2(16, 4); GIW_CHK2(16, 8); GIW_CHK2(16, 16);
#undef GIW_CHK2
    chk((const void*)gather_inner_fwd_wide_kernel<32, 16, CFFM_ACT_RELU, 32>); chk((const void*)gather_inner_fwd_wide_kernel<32, 16, -1, 32>);
    chk((const void*)gather_inner_fwd_wide_kernel<32, 16, CFFM_ACT_RELU, 32, true>); chk((const void*)gather_inner_fwd_wide_kernel<32, 16, -1, 32, true>);
    if (asked) cached = ok ? 1 : 0;
    return ok;
}

int cffm_gather_inner_fwd_wide(const cffm_shape_t* s, const cffm_tables_t* tab, const float* theta, const int32_t* ids, int32_t B,
                               void* ws, hipStream_t stream, int tab_stride, int tab_rows) {
    if (!cffm_wide_regather_ok(s) || !tab || !ids) return CFFM_ERR_UNSUPPORTED;
    if (B <= 0) return 0;
    cffm_theta_layout_t tl; cffm_ws_layout_t wl;
    cffm_theta_layout(s, &tl); cffm_ws_layout(s, B, &wl);
    const Geo g = make_geo(s);
    char* w = (char*)ws;
    GatherInnerWideArgs a;
    a.inner = tab->inner_emb; a.outer = tab->outer_emb; a.fbias = tab->feat_bias; a.ids = ids;
    a.cw = theta + tl.inner_cw; a.cb = theta + tl.inner_cb; a.wd = theta + tl.inner_dw; a.bd = theta + tl.inner_db;
    a.inner_out = (float*)(w + wl.inner_out); a.t1 = (float*)(w + wl.t1); a.fb = (float*)(w + wl.fb);
    a.keys = tab_stride > 0 ? nullptr : (unsigned long long*)(w + wl.sort_keys);   // record indices are not update keys
    a.B = B; a.M = tab_rows > 0 ? tab_rows : s->M; a.F = g.F; a.K = g.K; a.D = g.D; a.P = g.P; a.t1w = 2 * g.D - 2; a.act = g.act;
    if (tab_stride > 0 && (tab_stride & 3)) return CFFM_ERR_BAD_SHAPE;            // rows are fetched in 16-byte pieces
    a.row4_in = tab_stride > 0 ? tab_stride / 4 : g.K / 4; a.row4_out = tab_stride > 0 ? tab_stride / 4 : g.D / 4;
    a.fb_stride = tab_stride > 0 ? tab_stride : 1;
    const int K2 = g.K / 2, NG = GIW_T / K2;
    const int upt = (g.P + NG - 1) / NG;                       // pairs per thread group
    const int npiece = g.F * (g.K / 4 + g.D / 4);
    if (npiece > GIW_T || g.D > 64 || g.F > 32 || GIW_E * 256 != GIW_T || GIW_E * g.F > 128) return CFFM_ERR_UNSUPPORTED;
    const int nphase = (B + GIW_E - 1) / GIW_E;
    int grid = nphase < 256 ? nphase : 256;                    // one workgroup per CU, phases dealt round-robin
    int uptT = upt <= 4 ? 4 : (upt <= 8 ? 8 : 16);
    if (upt > 16) return CFFM_ERR_UNSUPPORTED;
    const size_t lds = (size_t)2 * 65536 + (size_t)(2 * GIW_E * 32 + 2 * 16 * GIW_E + NG * uptT + 4 * 128 + 2 * 4 * GIW_E * 64 * 2 + 2 * 4 * GIW_E) * 4;   // + S0P, S0Q (circulant instance)
    if (K2 == 32) {
        if (uptT == 4) return launch_giw<32, 4>(a, grid, lds, stream);
        if (uptT == 8) return launch_giw<32, 8>(a, grid, lds, stream);
        return launch_giw<32, 16>(a, grid, lds, stream);
    }
    if (K2 == 16) {
        if (uptT == 4) return launch_giw<16, 4>(a, grid, lds, stream);
        if (uptT == 8) return launch_giw<16, 8>(a, grid, lds, stream);
        return launch_giw<16, 16>(a, grid, lds, stream);
    }
    return CFFM_ERR_UNSUPPORTED;
}

extern "C" int cffm_gather_inner_fwd_ok(const cffm_shape_t* s) { return cffm_wide_regather_ok(s) ? 1 : 0; }
extern "C" int cffm_gather_inner_fwd(const cffm_shape_t* s, const cffm_tables_t* t, const float* theta, const int32_t* ids,
                                     int32_t B, void* ws, void* stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (!t || !theta || !ids || !ws) return CFFM_ERR_BAD_SHAPE;
    return cffm_gather_inner_fwd_wide(s, t, theta, ids, B, ws, (hipStream_t)stream, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the inner branch for the wide shapes (same applicability as cffm_gather_inner_fwd_wide).  inner_bwd_body was
// written for the README shapes: one 256-thread workgroup per example, the dense(1) kernel gradient read-modify-written in the
// global slab once per unit and example (2 GB of L2 traffic at the stress shape), dE in four LDS planes fed by ds_add_f32:
// 2.96 ms at F32 K64 B8192.  Here thread (f, t) owns the two floats dEi[f][2t], dEi[f][2t+1] of every example of its slab and
// walks the F-1 partners j of row f: the unit (pair {f, j}, t) is evaluated from BOTH of its rows (twice the arithmetic, no atomics,
// no cross-thread sums: dEi accumulates in two registers in a fixed partner order), the side with f < j also owns the unit's
// dense-kernel gradient - 2 x (F-1-f) register accumulators that are written to the slab ONCE - and the conv-filter / bias
// gradients.  Rows come HBM/L2 -> LDS by global_load_lds (next example in flight behind the current one).
// ---------------------------------------------------------------------------------------------------------------------
struct InnerBwdWideArgs {
    RowSrc rows;                            // inner rows: ws.Ei (idx == NULL) or the inner table + the batch's ids
    const float *dout, *out, *y;            // dout == NULL: dL/dout = head_dout(loss, out[b], y[b], invB, L) on the fly (as inner_bwd_body)
    const float *cw, *cb, *wd;
    float* dEi;                             // [B][F][K]
    float *slab_cw, *slab_cb, *slab_dw, *slab_db;     // slab 0
    int64_t slab_stride;
    int B, F, K, P, act, loss;
    float invB, L;
};

template <int K2, int ACTC>
__global__ __launch_bounds__(1024) void inner_bwd_wide_kernel(InnerBwdWideArgs a) {
    constexpr int T = 1024, K = 2 * K2, K4 = K / 4, NJ = 31;   // NJ: partners of a row at F = 32
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* E0 = reinterpret_cast<float*>(smem);                // [2][F][K] rows of the current / next example
    float* red = E0 + 2 * 32 * K;                               // [16 waves][8]
    float* dWl = red + 16 * 8;                                  // [P][K] dense(1) kernel gradient of this slab: every (pair, t) element
                                                                // belongs to ONE thread, which adds to it once per example (64 registers
                                                                // per lane if kept there: 245 spilled under the 128-register budget)
    const int F = a.F, act = ACTC >= 0 ? ACTC : a.act;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = tid / K2, t = tid - f * K2;
    const bool on = f < F;
    const int slab = blockIdx.x, nslab = gridDim.x;
    float* slab_cw = a.slab_cw + slab * a.slab_stride; float* slab_cb = a.slab_cb + slab * a.slab_stride;
    float* slab_dw = a.slab_dw + slab * a.slab_stride; float* slab_db = a.slab_db + slab * a.slab_stride;
    if (tid < 8) slab_cw[tid] = 0.f;                            // inner_cw (4) + inner_cb (2 + 2 pad): gaps read as zeros
    if (tid < 4) slab_db[tid] = 0.f;
    const float cw0 = a.cw[0], cw1 = a.cw[1], cw2 = a.cw[2], cw3 = a.cw[3], cb0 = a.cb[0], cb1 = a.cb[1];
    const f32x2* wd2 = reinterpret_cast<const f32x2*>(a.wd);
    const int basef = on ? f * (2 * F - f - 1) / 2 : 0;        // first pair of row f on its i side
    for (int e = tid; e < a.P * K2; e += T) reinterpret_cast<f32x2*>(dWl)[e] = (f32x2){0.f, 0.f};
    float gcw[4] = {0.f, 0.f, 0.f, 0.f}, gcb[2] = {0.f, 0.f}, gdb = 0.f;
    // fetch of the rows of example b into buffer buf: piece q = tid (16 bytes), 16 / 8 lanes per row
    auto fetch = [&](int b, int buf) {
        if (tid < F * K4) {
            const int fr = tid / K4, c = tid - fr * K4;
            const float* row = row_ptr(a.rows.base, a.rows.idx, a.rows.M, (int64_t)b * F + fr, K, a.rows.stride);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(row + 4 * c),
                                             (void __attribute__((address_space(3)))*)(E0 + buf * 32 * K + wave * 256), 16, 0, 0);
        }
    };
    int it = 0;
    if (slab < a.B) fetch(slab, 0);
    for (int b = slab; b < a.B; b += nslab, ++it) {
        const float db = a.dout ? a.dout[b] : head_dout(a.loss, a.out[b], a.y[b], a.invB, a.L);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of example b have landed
        __syncthreads();                                       // ... and everybody's; the other buffer is free again
        if (b + nslab < a.B) fetch(b + nslab, (it + 1) & 1);
        const float* E = E0 + (it & 1) * 32 * K;
        if (on) {
            const f32x2 ef = *reinterpret_cast<const f32x2*>(E + f * K + 2 * t);
            f32x2 accF = (f32x2){0.f, 0.f};
            // the pair index and row offset of every partner are invariants of the example loop; hoisted out of it they are ~100
            // registers (374 spilled): an opaque zero keeps their handful of integer instructions inside the loop
            int fo;
            asm volatile("v_mov_b32 %0, %1" : "=v"(fo) : "v"(f));
            // partners in chunks of four: the four (row piece, weight pair) loads of a chunk are in flight together, and nothing
            // moves across a chunk (left alone, hipcc hoists the 62 loads of ALL partners to the top: 114 spilled registers)
#pragma clang loop unroll(full)
            for (int c4 = 0; c4 < (NJ + 3) / 4; ++c4) {
                f32x2 ejv[4], w2v[4];
                int pv[4];
#pragma clang loop unroll(full)
                for (int u = 0; u < 4; ++u) {
                    const int jj = 4 * c4 + u;
                    int j = fo + 1 + jj;
                    const bool iside = j < F;                   // row f is the first row of the pair: this thread owns the unit's dense gradients
                    j = iside ? j : j - F;
                    const bool live = jj < F - 1;
                    j = live ? j : 0;
                    pv[u] = live ? (iside ? fo * (2 * F - fo - 1) / 2 + jj : j * (2 * F - j - 1) / 2 + (fo - j - 1)) : 0;
                    ejv[u] = *reinterpret_cast<const f32x2*>(E + j * K + 2 * t);
                    w2v[u] = wd2[(int64_t)pv[u] * K2 + t];      // 127 KB vector, L2-resident
                }
#pragma clang loop unroll(full)
                for (int u = 0; u < 4; ++u) {
                    const int jj = 4 * c4 + u;
                    if (jj >= NJ) continue;
                    const bool iside = fo + 1 + jj < F, live = jj < F - 1;
                    const f32x2 ej = ejv[u], w2 = w2v[u];
                    // forward of the unit (CFFM.py:310-332), as inner_unit()
                    const float I0 = ef.x * ej.x, I1 = ef.y * ej.y;
                    const float x0 = act_f(I0, act), x1 = act_f(I1, act);
                    const float z0 = x0 * cw0 + x1 * cw2 + cb0, z1 = x0 * cw1 + x1 * cw3 + cb1;
                    const float r0 = fmaxf(z0, 0.f), r1 = fmaxf(z1, 0.f);
                    // backward
                    const float dbl = live ? db : 0.f;          // F < 32: the surplus partners contribute exact zeros
                    const float ds0 = dbl * w2.x, ds1 = dbl * w2.y;
                    const float dz0 = ds0 * act_relu_grad(r0, act), dz1 = ds1 * act_relu_grad(r1, act);
                    const float dmp = ds0 + ds1;
                    const bool firstmax = x0 >= x1;             // max-pool grad: first element on ties
                    const float dx0 = dz0 * cw0 + dz1 * cw1 + (firstmax ? dmp : 0.f);
                    const float dx1 = dz0 * cw2 + dz1 * cw3 + (firstmax ? 0.f : dmp);
                    const float dI0 = dx0 * act_grad_f(I0, act), dI1 = dx1 * act_grad_f(I1, act);
                    accF.x += dI0 * ej.x; accF.y += dI1 * ej.y; // dEi[f] += dI (.) e_j, partners in a fixed order
                    const float di = iside ? dbl : 0.f;         // the i side owns the unit's parameter gradients
                    const float mp = fmaxf(x0, x1);
                    if (iside && live) {                        // dense(1) kernel gradient: flat * dout, into this thread's own LDS element
                        f32x2* dwp = reinterpret_cast<f32x2*>(dWl) + pv[u] * K2 + t;
                        f32x2 dv = *dwp;
                        dv.x += (act_pos(r0, act) + mp) * di;
                        dv.y += (act_pos(r1, act) + mp) * di;
                        *dwp = dv;
                    }
                    const float e0 = iside ? dz0 : 0.f, e1 = iside ? dz1 : 0.f;
                    gcw[0] += e0 * x0; gcw[1] += e1 * x0; gcw[2] += e0 * x1; gcw[3] += e1 * x1;
                    gcb[0] += e0; gcb[1] += e1;
                }
                // (all seven accumulators are pinned: pinning accF alone made the compiler sink the 6 filter / bias gradient
                // updates of every partner to the end of the example and keep their operands in scratch until then)
                asm volatile("" : "+v"(accF), "+v"(gcw[0]), "+v"(gcw[1]), "+v"(gcw[2]), "+v"(gcw[3]), "+v"(gcb[0]), "+v"(gcb[1]) : : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            *reinterpret_cast<f32x2*>(a.dEi + ((int64_t)b * F + f) * K + 2 * t) = accF;
        }
        if (tid == 0) gdb += db;
    }
    // the slab: this thread's units of the dense(1) kernel gradient (zeros when the slab saw no example)
    __syncthreads();
    for (int e = tid; e < a.P * K2; e += T) reinterpret_cast<f32x2*>(slab_dw)[e] = reinterpret_cast<const f32x2*>(dWl)[e];
    float r[7] = {gcw[0], gcw[1], gcw[2], gcw[3], gcb[0], gcb[1], gdb};
#pragma unroll
    for (int q = 0; q < 7; ++q) r[q] = wave_sum(r[q]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 7; ++q) red[wave * 8 + q] = r[q];
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            float v = 0.f;
            for (int w = 0; w < 16; ++w) v += red[w * 8 + q];
            r[q] = v;
        }
        for (int q = 0; q < 4; ++q) slab_cw[q] = r[q];
        slab_cb[0] = r[4]; slab_cb[1] = r[5];
        slab_db[0] = r[6];
    }
}

template <int K2>
static int launch_ibw(const InnerBwdWideArgs& a, int nslab, hipStream_t st) {
    const size_t lds = (size_t)(2 * 32 * 2 * K2 + 16 * 8 + a.P * 2 * K2) * 4;
    hipError_t e1 = hipFuncSetAttribute((const void*)inner_bwd_wide_kernel<K2, CFFM_ACT_RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e2 = hipFuncSetAttribute((const void*)inner_bwd_wide_kernel<K2, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e1 != hipSuccess || e2 != hipSuccess) return (int)(e1 != hipSuccess ? e1 : e2);
    if (a.act == CFFM_ACT_RELU) hipLaunchKernelGGL((inner_bwd_wide_kernel<K2, CFFM_ACT_RELU>), dim3(nslab), dim3(1024), lds, st, a);
    else hipLaunchKernelGGL((inner_bwd_wide_kernel<K2, -1>), dim3(nslab), dim3(1024), lds, st, a);
    CFFM_CHECK_LAUNCH();
    return 0;
}

// rs == NULL: rows from ws.Ei.  Returns CFFM_ERR_UNSUPPORTED for shapes outside cffm_wide_regather_ok().
int cffm_inner_bwd_wide(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, const RowSrc* rs, hipStream_t stream) {
    if (!cffm_wide_regather_ok(s) || s->F > 32) return CFFM_ERR_UNSUPPORTED;
    if (B <= 0) return 0;
    InnerBwdArgs o;
    const int nslab = fill_inner_bwd_args(s, theta, ws, B, &o);
    InnerBwdWideArgs a;
    a.rows.base = rs ? rs->base : o.Ei; a.rows.idx = rs ? rs->idx : nullptr; a.rows.M = rs ? rs->M : 0; a.rows.stride = rs ? rs->stride : 0;
    a.dout = o.dout; a.out = o.out; a.y = o.y; a.cw = o.cw; a.cb = o.cb; a.wd = o.wd; a.dEi = o.dEi;
    a.slab_cw = o.slab_cw; a.slab_cb = o.slab_cb; a.slab_dw = o.slab_dw; a.slab_db = o.slab_db; a.slab_stride = o.slab_stride;
    a.B = B; a.F = o.g.F; a.K = o.g.K; a.P = o.g.P; a.act = o.g.act; a.loss = o.loss; a.invB = o.invB; a.L = 1.f;
    if (o.g.K == 64) return launch_ibw<32>(a, nslab, stream);
    if (o.g.K == 32) return launch_ibw<16>(a, nslab, stream);
    return CFFM_ERR_UNSUPPORTED;
}
