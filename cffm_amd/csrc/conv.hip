// Outer-product / conv-stack contractions of the CFFM graph on fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
//   conv_fwd<GEN=true>   CFFM.py:355-367 + :385-386 (i = 0): the F x F outer-product map is never
//                        materialised; the A operand of the implicit GEMM is generated per lane from
//                        the example's embedding tile held in LDS (two ds_read + one v_mul per element).
//   conv_fwd<GEN=false>  CFFM.py:385-387 (i >= 1): 2x2 / stride-2 VALID conv as an implicit GEMM
//                        M = B*So*So, N = P, K = 4P; the A operand is act(C_{l-1}) read straight from
//                        HBM/L2 as 16-byte pieces of the four patch rows (no im2col buffer).
//   dgrad<L0=false>      gradient wrt the layer input; patches never overlap, so it is a scatter-store
//                        fused with the broadcast sum-pool gradient and the relu/act mask.
//   dgrad<L0=true>       layer 0: the dA tile is contracted with the embedding tile straight into dEo
//                        (per-wavefront private LDS accumulators, merged in wave order).
//   wgrad<GEN>           weight/bias gradients: split-K over CFFM_NSLAB slabs, both operands via LDS.
//
// Tiling: 256-thread workgroups = 4 wavefronts; a wavefront owns RM x NT tiles of 16x16 (rows = output
// positions, cols = output channels).  The weight tile of a 32-deep K step is shared by the four
// wavefronts through a double-buffered LDS tile, one barrier per K step.  Inside a K step a lane holds
// FOUR consecutive k of its row (one global_load_dwordx4) and feeds them to four MFMAs; the B fragment
// uses the same k permutation, which the contraction does not care about.
// Channels are padded to Pp = ceil16(P) in every activation tensor so that 16-byte pieces never
// straddle a row and tiles never straddle a filter tap; padded channels hold zeros.
#include <cstdlib>
#include <cstring>

#include "internal.hpp"

#include <type_traits>

#include <stdlib.h>
#include "inner_body.hpp"
#include "head_body.hpp"
#include "sort_body.hpp"

#define KSTEP 32
#define WG_KM 64       // rows of the M (reduction) dimension staged per wgrad step

// XCD-aware blockIdx -> tile mapping.  Workgroups are dealt round-robin over the 8 XCDs (block b and b + 8 share one), and
// every XCD has an L2 of its own: tiles that read the SAME operand rows should therefore be consecutive blocks OF ONE XCD,
// not consecutive block ids.  A launch of 8 * per_xcd blocks: XCD x = id & 7 works through the contiguous range
// [x * per_xcd, (x + 1) * per_xcd) of the logical tile order (the operand-sharing index fastest) in the order w = id >> 3.
// Placement is only a speed assumption (MI355X_MICROARCH.md: dispatch order is not a contract): any block -> XCD map
// gives the same results.
__device__ __forceinline__ int64_t xcd_tile(int64_t per_xcd) {
    const int64_t id = blockIdx.x;
    return (id & 7) * per_xcd + (id >> 3);
}
__host__ __device__ static inline int64_t xcd_per(int64_t tiles) { return (tiles + 7) / 8; }

struct RowPos { int b, y, x; };
__device__ __forceinline__ RowPos row_pos(int64_t m, int lgSo) {
    RowPos p;
    const int So = 1 << lgSo;
    p.x = (int)(m & (So - 1));
    p.y = (int)((m >> lgSo) & (So - 1));
    p.b = (int)(m >> (2 * lgSo));
    return p;
}

// -------------------------------------------------------------------------------------------------
// Generic K loop: acc[RM][NT] += A(rows of this wave, K) * Wtile(K, BN).
//   loadA(ks, areg): fills areg[RM][2] (float4 = 4 consecutive k of the lane's row, halves h = 0,1 of the 32-deep step)
//   w: where the weight tile comes from.  TRANS = false: element (k, n) = w.W[k * w.ld + n] (forward: HWIO filter);
//      TRANS = true: element (k, n) = w.W[n * w.ld + k] (input gradient: the same filter read transposed).  Elements with
//      k >= w.k_lim or n >= w.n_lim are zeros (k_lim is a multiple of 16: a group of 4 consecutive k is in or out as one).
//   khalves: number of valid 16-deep halves of the K dimension
// The weight tile of a step lives in LDS as [8 k-quads][BN columns][4 k]: a thread stages whole (k-quad, column) records
// with ONE 16-byte store at 16 * its index (conflict-free), and a lane's B fragments of the four MFMAs that consume one
// float4 of A come back with ONE ds_read_b128 per column tile (lanes of a k group read 256 contiguous bytes).  Every
// tile of the wave is computed (columns beyond n_lim are zeros; the callers' epilogues skip them): no branch sits between
// the MFMAs of a step, so the compiler keeps the global loads of the next step in flight behind them.
// -------------------------------------------------------------------------------------------------
// act on four values known to be >= 0 (see act_pos): ONE wave-uniform branch for the whole group instead of a switch per
// element - relu / prelu / elu are the identity there
__device__ __forceinline__ void act_pos4(float4& c, int act) {
    if (act == CFFM_ACT_SELU) {
        c.x *= CFFM_SELU_SCALE; c.y *= CFFM_SELU_SCALE; c.z *= CFFM_SELU_SCALE; c.w *= CFFM_SELU_SCALE;
    } else if (act == CFFM_ACT_GELU) {
        c.x = act_pos(c.x, CFFM_ACT_GELU); c.y = act_pos(c.y, CFFM_ACT_GELU);
        c.z = act_pos(c.z, CFFM_ACT_GELU); c.w = act_pos(c.w, CFFM_ACT_GELU);
    }
}

struct WSpec { const float* W; int ld, k_lim, n_lim, n0; const void* pre = nullptr; int npad = 0; };   // pre / npad: gemm_tile_b3's pre-split image

template <int NT, int RM, bool TRANS, class LoadA>
__device__ __forceinline__ void gemm_tile(f32x4 (&acc)[RM][NT], int khalves, float* Ws, const WSpec w, LoadA loadA, int actA = CFFM_ACT_RELU) {
    constexpr int BN = NT * 16;
    constexpr int NREC = 8 * BN;                       // (k-quad, column) records of one step
    constexpr int NP = (NREC + 255) / 256;             // records per thread
    constexpr int WBUF = NREC * 4;                     // floats per LDS buffer
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, kk = lane >> 4;
    const int nks = (khalves + 1) >> 1;
    float4 areg[RM][2], anext[RM][2];
    float4 w4[NP];
    auto fetchW = [&](int ks) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int rec = tid + 256 * i, kq = rec / BN, c = rec - kq * BN;
            const int k = ks * KSTEP + 4 * kq, n = w.n0 + c;
            const bool ok = (NREC % 256 == 0 || rec < NREC) && k < w.k_lim && n < w.n_lim;
            // unconditional loads from a clamped address, zeroed by a select: no exec-mask branch around a load
            const int kc = k < w.k_lim ? k : w.k_lim - 4, nc = n < w.n_lim ? n : w.n_lim - 1;
            float4 v;
            if (TRANS) {
                v = *reinterpret_cast<const float4*>(w.W + (int64_t)nc * w.ld + kc);
            } else {
                const float* src = w.W + (int64_t)kc * w.ld + nc;
                v = make_float4(src[0], src[w.ld], src[2 * w.ld], src[3 * w.ld]);
            }
            w4[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto storeW = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int rec = tid + 256 * i;
            if (NREC % 256 == 0 || rec < NREC) *reinterpret_cast<float4*>(buf + 4 * rec) = w4[i];
        }
    };
    loadA(0, areg);
    fetchW(0);
    __syncthreads();                          // previous users of Ws are done
    storeW(Ws);
    __syncthreads();
    for (int ks = 0; ks < nks; ++ks) {
        const bool more = ks + 1 < nks;
        if (more) {
            loadA(ks + 1, anext);
            fetchW(ks + 1);
        }
        // the activation of the A operand (forward: act(C_{l-1}), C >= 0) is applied HERE, to the operands loaded one
        // step ago - not at load time, where it would put a wait for the prefetch right behind its issue
        if (actA != CFFM_ACT_RELU && actA != CFFM_ACT_PRELU && actA != CFFM_ACT_ELU) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) act_pos4(areg[rm][h], actA);
        }
        const float* Wb = Ws + (ks & 1) * WBUF;
        const int nh = (ks == nks - 1 && (khalves & 1)) ? 1 : 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h < nh) {
                float4 bf[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bf[nt] = *reinterpret_cast<const float4*>(Wb + ((h * 4 + kk) * BN + nt * 16 + r) * 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int rm = 0; rm < RM; ++rm) {
                        const float av = t == 0 ? areg[rm][h].x : t == 1 ? areg[rm][h].y : t == 2 ? areg[rm][h].z : areg[rm][h].w;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const float bv = t == 0 ? bf[nt].x : t == 1 ? bf[nt].y : t == 2 ? bf[nt].z : bf[nt].w;
                            acc[rm][nt] = mfma16(av, bv, acc[rm][nt]);
                        }
                    }
                }
            }
        }
        if (more) {
            storeW(Ws + ((ks + 1) & 1) * WBUF);
#pragma unroll
            for (int rm = 0; rm < RM; ++rm) { areg[rm][0] = anext[rm][0]; areg[rm][1] = anext[rm][1]; }
        }
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------------------
// The same K loop on the bf16 MFMA pipe at fp32 accuracy ("bf16x3"): every fp32 operand is split WITHOUT ERROR into three bf16
// pieces, x = x1 + x2 + x3 (x1 = the top 8 significand bits of x, x2 those of the exact remainder x - x1, x3 those of x - x1 -
// x2: 24 bits in all, same exponent range as fp32), and a product a*b is taken as the six cross terms of weight >= 2^-16,
//     a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1),
// each term an EXACT fp32 value (8 x 8 significand bits) added into the fp32 accumulator of v_mfma_f32_16x16x32_bf16.  What is
// dropped - a2 b3 + a3 b2 + a3 b3 - is below 2^-23 |a b|: the rounding error of ONE fp32 multiply, against an accumulation
// over K = 496 .. 1984 terms that both forms round term by term.  Six bf16 MFMAs of 16 cycles do the work of eight fp32 MFMAs of
// 32 (16x16x4, same output layout): 2.7x on the pipe that bounds the wide shapes (CFFM.py:384-391 at F = 32: 98 % of the step).
// CFFM_CONV_FP32=1 runs the fp32 MFMA loop above instead (A/B, and the reference the parity of this one was first checked on).
//
// Operands: a lane's A operand of a 32-deep step is the two float4 the fp32 loop already loads (k = 4kk .. +3 and 16 + 4kk ..
// +3 of its row - any assignment of k to lanes is valid as long as B uses the same one); the weight tile lives in LDS as
// [piece][kk][column][8 bf16]: one ds_read_b128 per (piece, column tile), conflict-free like the fp32 image.
// -------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// two floats -> their three bf16 pieces, packed (low half = first float).  Truncation splits: every remainder is exact.
__device__ __forceinline__ void split_bf16x3(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    p1 = __builtin_amdgcn_perm(u1, u0, 0x07060302u);                              // (hi16(x0), hi16(x1))
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    p2 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    p3 = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}
__device__ __forceinline__ void split8_bf16x3(const float4& lo, const float4& hi, u32x4_t (&p)[3]) {
    unsigned q[4][3];
    split_bf16x3(lo.x, lo.y, q[0][0], q[0][1], q[0][2]);
    split_bf16x3(lo.z, lo.w, q[1][0], q[1][1], q[1][2]);
    split_bf16x3(hi.x, hi.y, q[2][0], q[2][1], q[2][2]);
    split_bf16x3(hi.z, hi.w, q[3][0], q[3][1], q[3][2]);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) p[pc] = (u32x4_t){q[0][pc], q[1][pc], q[2][pc], q[3][pc]};
}
__device__ __forceinline__ f32x4 mfma_bf16(const u32x4_t& a, const u32x4_t& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

template <int NT>
constexpr int gemm_b3_lds_bytes() { return 2 * 3 * 4 * NT * 16 * 16; }          // [2 buffers][3 pieces][4 kk][BN columns][16 B]

// ACTA: the A operand goes through act_pos4(actA) first (selu / gelu on the stored relu output); false: relu / prelu / elu, the
// identity there - a compile-time flag, so that the main loop of the common case is branch-free
template <int NT, int RM, bool TRANS, bool ACTA, class LoadA>
__device__ __forceinline__ void gemm_tile_b3(f32x4 (&acc)[RM][NT], int khalves, float* Ws, const WSpec w, LoadA loadA, int actA = CFFM_ACT_RELU) {
    constexpr int BN = NT * 16;
    constexpr int NREC = 4 * BN;                       // (kk, column) double records of one 32-deep step: 8 k each
    constexpr int PBUF = 3 * NREC;                     // 16-byte records per LDS buffer
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, kk = lane >> 4;
    const int nks = (khalves + 1) >> 1;
    u32x4_t* Wl = reinterpret_cast<u32x4_t*>(Ws);
    // The weight tile is split ONCE per launch (pack_w_b3_kernel, w.pre) into the very records this loop reads,
    // [k-step][piece][kk][padded column]: a step's tile is 12 rows of BN records = 24 wave-sized pieces that go global -> LDS by
    // DMA (global_load_lds_dwordx4, six per wave) - no staging registers, no split instructions, no ds_write in the loop.
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto dmaW = [&](int ks, u32x4_t* buf) {
        static_assert(BN == 128, "six 1 KB pieces per wave");
        const u32x4_t* src = reinterpret_cast<const u32x4_t*>(w.pre) + (int64_t)ks * 12 * w.npad + w.n0 + lane;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int j = wave * 6 + i, row = j >> 1, half = j & 1;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (int64_t)row * w.npad + half * 64),
                                             (void __attribute__((address_space(3)))*)(buf + row * BN + half * 64), 16, 0, 0);
        }
    };
    // A operands are requested TWO steps ahead: a step of this loop is 1536 MFMA cycles per wave (the fp32 loop: 4096), too short
    // to cover an HBM round trip under load with one step of lookahead.  It costs no register: the raw float4s of a step are dead
    // once they are split into bf16 pieces at the top of the step, so the loads of step ks + 2 go into the registers step ks just
    // gave up (two register sets, the loop runs in pairs of steps).  The filter tile comes out of L2 / MALL and stays one step ahead;
    // it is requested BEFORE the A loads, so that "at most RM * 2 loads still in flight" (vmcnt) means "the DMA has landed".
    float4 aA[RM][2], aB[RM][2];
    // FULL = true: a step of the main loop, both lookaheads exist - no branch in it, so that hipcc's own wait for `cur` at the top
    // of the next step counts the RM * 2 younger loads exactly (vmcnt(RM * 2)) instead of falling back to vmcnt(0) where control
    // flow merges
    auto body = [&](int ks, float4 (&cur)[RM][2], auto full) {
        constexpr bool FULL = decltype(full)::value;
        const bool more = FULL || ks + 1 < nks, more2 = FULL || ks + 2 < nks;
        if constexpr (ACTA) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) act_pos4(cur[rm][h], actA);
        }
        u32x4_t ap[RM][3];
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) split8_bf16x3(cur[rm][0], cur[rm][1], ap[rm]);
        __builtin_amdgcn_sched_barrier(0);                         // the splits read `cur` before the loads below overwrite it
        if (more) dmaW(ks + 1, Wl + ((ks + 1) & 1) * PBUF);        // that buffer was last read in step ks - 1, behind a barrier
        __builtin_amdgcn_sched_barrier(0);                         // DMA first, A loads second: see the vmcnt below
        if (more2) loadA(ks + 2, cur);
        __builtin_amdgcn_sched_barrier(0);
        // (an odd last step: the B records of its second half are zeros - the image is zero beyond k_lim - and the A values
        //  there are finite numbers read from a clamped address)
        const u32x4_t* Wb = Wl + (ks & 1) * PBUF + kk * BN + r;
        // two column tiles at a time: 2 * RM independent accumulation chains, so that consecutive MFMAs never wait on each other
        static_assert(NT % 2 == 0, "column tiles are taken in pairs");
#pragma unroll
        for (int nt = 0; nt < NT; nt += 2) {
            u32x4_t b[2][3];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) b[u][pc] = Wb[pc * NREC + (nt + u) * 16];
            // (A piece, B piece) of the six terms, the small ones first
            constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int rm = 0; rm < RM; ++rm) acc[rm][nt + u] = mfma_bf16(ap[rm][TA[t]], b[u][TB[t]], acc[rm][nt + u]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) {                                                // this wave's pieces of the next tile have landed; the A loads of
            if (more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RM * 2) : "memory");   // step ks + 2 (issued after them) may stay in flight
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // A bare s_barrier: __syncthreads() and lds_barrier() both make hipcc wait for EVERY vector-memory operation in flight
        // (vmcnt(0): it counts the LDS DMA as a write to the fenced address space), which would take the second step of lookahead
        // away again.  What the barrier has to order is covered explicitly: this wave's DMA pieces by the vmcnt above, its LDS reads
        // of the current tile by the MFMAs that consumed them.
        asm volatile("s_barrier" ::: "memory");
    };
    loadA(0, aA);
    __syncthreads();                          // previous users of Ws are done
    dmaW(0, Wl);
    if (nks > 1) loadA(1, aB);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int ks = 0;
    for (; ks + 3 < nks; ks += 2) {
        body(ks, aA, std::true_type());
        body(ks + 1, aB, std::true_type());
    }
    for (; ks < nks; ks += 2) {               // the last two to three steps
        body(ks, aA, std::false_type());
        if (ks + 1 < nks) body(ks + 1, aB, std::false_type());
    }
}

// The weight tile of gemm_tile_b3 split once per launch: image [k-step][piece][kk][npad columns] of 8-bf16 records, record
// (ks, kk, n) = the three pieces of element (k, n) for k = 32 ks + 16 h + 4 kk + t (h = 0, 1; t = 0 .. 3) in the order the loop's A
// operand uses; zeros beyond k_lim / n_lim.  TRANS as in WSpec.  One thread per record: ~6 MB per layer, a few microseconds.
template <bool TRANS>
__global__ __launch_bounds__(256) void pack_w_b3_kernel(const float* __restrict__ W, int ld, int k_lim, int n_lim, int npad, int nks,
                                                        u32x4_t* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)nks * 4 * npad;
    if (idx >= total) return;
    const int n = (int)(idx % npad), kk = (int)((idx / npad) & 3), ks = (int)(idx / ((int64_t)4 * npad));
    float v[8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = ks * KSTEP + 16 * h + 4 * kk + t;
            const bool ok = k < k_lim && n < n_lim;
            const int kc = k < k_lim ? k : k_lim - 1, nc = n < n_lim ? n : n_lim - 1;
            const float x = TRANS ? W[(int64_t)nc * ld + kc] : W[(int64_t)kc * ld + nc];
            v[4 * h + t] = ok ? x : 0.f;
        }
    u32x4_t p[3];
    split8_bf16x3(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), p);
    const int64_t o = ((int64_t)ks * 12 + kk) * npad + n;
    out[o] = p[0]; out[o + (int64_t)4 * npad] = p[1]; out[o + (int64_t)8 * npad] = p[2];
}
static inline int64_t wb3_image_bytes(int K, int N) {        // K = reduction length, N = columns of the weight operand
    return (int64_t)((K + KSTEP - 1) / KSTEP) * 12 * ((N + 127) / 128 * 128) * 16;
}
int64_t cffm_wb3_bytes(int Pp) {
    const int64_t f = wb3_image_bytes(4 * Pp, Pp), d = wb3_image_bytes(Pp, 4 * Pp);
    return f > d ? f : d;
}
template <bool TRANS>
static int pack_w_b3(const float* W, int ld, int K, int N, void* out, hipStream_t st) {
    const int nks = (K + KSTEP - 1) / KSTEP, npad = (N + 127) / 128 * 128;
    const int64_t total = (int64_t)nks * 4 * npad;
    hipLaunchKernelGGL((pack_w_b3_kernel<TRANS>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, ld, K, N, npad, nks,
                       (u32x4_t*)out);
    CFFM_CHECK_LAUNCH();
    return 0;
}

// stage the embedding rows of examples [b0, b0 + n_ex) into LDS with row pitch Dp = D + 1
__device__ __forceinline__ void stage_examples(float* Es, const float* __restrict__ Eo, int b0, int n_ex, int B,
                                               int F, int D, int Dp) {
    const int per = F * D;
    const float invD = 1.f / (float)D;
    for (int e = threadIdx.x; e < n_ex * per; e += blockDim.x) {
        const int row = fast_div(e, invD), d = e - row * D;          // row = ex * F + f
        const int64_t src = (int64_t)b0 * per + e;
        Es[row * Dp + d] = src < (int64_t)B * per ? Eo[src] : 0.f;
    }
}

struct ConvArgs {
    const float* in;     // GEN: Eo [B,F,D]; else C_{l-1} [B,Sin,Sin,Pp]
    const float* W;      // [4][Pp][Pp] (HWIO padded)
    const float* bias;   // [Pp]
    float* out;          // relu(conv + bias) [B,So,So,Pp]
    int64_t Mtot;        // B*So*So
    int B, lgSo, P, Pp, F, D, act;
    int nblk;            // conv_fwd_kernel: column blocks of a row tile (the grid is 1-D, see xcd_tile)
    int dbg;             // debug build only (make TILE_DBG=1): phase-skipping bits for tools/dbg_tile.py, 0 otherwise
    const int32_t* idx = nullptr;   // tiled layer 0 only: non-NULL = `in` is the outer TABLE [M][D] and row (b, f) is idx[b*F+f] (RowSrc)
    int idxM = 0;
    int idxStride = 0;              // floats between rows of the table (0 = D): see RowSrc
    float* pool = nullptr;          // wide shapes: partial sum pools of act(out), [B][So][pool_np] (pool_partials(), common.hpp)
    int pool_np = 0;
    uint16_t* relu = nullptr;       // wide shapes (tiled layer-0 forward, conv_fwd_kernel): bit mask of out > 0, [B*So*So][Pp/16] 16-bit words (ws.relu0)
    void* wb3 = nullptr;            // bf16x3 instance: scratch for the pre-split filter image (ws.wb3; NULL: split while staging)
    int wb3_npad = 0;
};

// The 128 x 128 instance of the wide shapes is held to 3 wavefronts per SIMD (166 VGPRs, nothing spilled; it took 106 + 96
// accumulation registers = 2 per SIMD without the bound): 12.93 -> 12.33 ms per launch at the stress shape.
// B3: the K loop on the bf16 pipe (gemm_tile_b3); everything around it - operand addressing, epilogue, pools, masks - is the same
template <int NT, int RM, bool GEN, bool B3 = false>
__global__ __launch_bounds__(256, (NT == 8 && RM == 2 && !GEN) ? 3 : 1) void conv_fwd_kernel(ConvArgs a) {
    constexpr int BN = NT * 16, LDW = BN + 4, BM = 64 * RM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ws = reinterpret_cast<float*>(smem);                    // [2][32][LDW]  (B3: [2][3][4][BN] 16-byte records)
    static_assert(!B3 || !GEN, "the bf16x3 loop serves the direct layers");
    uint32_t* lut = reinterpret_cast<uint32_t*>(Ws + (B3 ? gemm_b3_lds_bytes<NT>() / 4 : 2 * KSTEP * LDW));   // [Pp]   (GEN)
    float* Es = reinterpret_cast<float*>(lut + a.Pp);             // [n_ex][F][Dp] (GEN)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kk = lane >> 4;
    const int So = 1 << a.lgSo, Sin = 2 * So, Pp = a.Pp, P = a.P, Dp = a.D + 1;
    const float invPp = 1.f / (float)Pp;
    // the nblk column blocks of one row tile read the same A rows: consecutive blocks of one XCD
    const int64_t mtiles = (a.Mtot + BM - 1) / BM;
    const int64_t tile = xcd_tile(xcd_per(mtiles) * a.nblk);
    if (tile >= mtiles * a.nblk) return;
    const int64_t m0 = (tile / a.nblk) * BM;
    const int n0 = (int)(tile % a.nblk) * BN;
    const int nvalid = min(NT, (Pp - n0) / 16);

    int64_t abase[RM];   // !GEN: float offset of the (dh=0,dw=0) patch row piece of this lane
    int iy[RM], jx[RM];  // GEN: LDS offsets of Eo[ex][0][2y] and Eo[ex][0][2x]
    const int b0 = (int)(m0 >> (2 * a.lgSo));
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) {
        int64_t m = m0 + wave * (16 * RM) + rm * 16 + r;
        if (m >= a.Mtot) m = a.Mtot - 1;
        const RowPos rp = row_pos(m, a.lgSo);
        if (GEN) {
            const int eoff = (rp.b - b0) * a.F * Dp;
            iy[rm] = eoff + 2 * rp.y;
            jx[rm] = eoff + 2 * rp.x;
        } else {
            abase[rm] = (((int64_t)rp.b * Sin + 2 * rp.y) * Sin + 2 * rp.x) * Pp + 4 * kk;
        }
    }
    if (GEN) {
        const int S2 = So * So;
        const int n_ex = BM > S2 ? BM / S2 : 1;
        build_pair_lut(lut, a.F, Pp);
        stage_examples(Es, a.in, b0, n_ex, a.B, a.F, a.D, Dp);
        __syncthreads();
    }

    auto loadA = [&](int ks, float4 (&reg)[RM][2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = ks * KSTEP + 16 * h;
            const int tap = fast_div(k, invPp), pb = k - tap * Pp, dh = tap >> 1, dw = tap & 1;
            if (GEN) {
                const uint4 l4 = *reinterpret_cast<const uint4*>(&lut[pb + 4 * kk]);
                const uint32_t ij[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) {
                    float v[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int i = ij[t] & 0xffff, j = ij[t] >> 16;
                        v[t] = Es[iy[rm] + i * Dp + dh] * Es[jx[rm] + j * Dp + dw];
                    }
                    reg[rm][h] = make_float4(v[0], v[1], v[2], v[3]);
                }
            } else {
                const int64_t toff = (int64_t)(dh * Sin + dw) * Pp + pb;
#pragma unroll
                for (int rm = 0; rm < RM; ++rm)
                    reg[rm][h] = *reinterpret_cast<const float4*>(a.in + abase[rm] + toff);
            }
        }
    };
    const WSpec wspec = {a.W, Pp, 4 * Pp, Pp, n0, B3 ? a.wb3 : nullptr, a.wb3_npad};

    f32x4 acc[RM][NT];
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[rm][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (B3) {
        if (a.act != CFFM_ACT_RELU && a.act != CFFM_ACT_PRELU && a.act != CFFM_ACT_ELU) gemm_tile_b3<NT, RM, false, true>(acc, 4 * Pp / 16, Ws, wspec, loadA, a.act);
        else gemm_tile_b3<NT, RM, false, false>(acc, 4 * Pp / 16, Ws, wspec, loadA);
    }
    else gemm_tile<NT, RM, false>(acc, 4 * Pp / 16, Ws, wspec, loadA, GEN ? CFFM_ACT_RELU : a.act);   // act(C_{l-1}) on the A operand

    float psum[RM][4];                                         // pool partials: this lane's column of every row it holds
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int j = 0; j < 4; ++j) psum[rm][j] = 0.f;
    const int pact = GEN ? CFFM_ACT_RELU : a.act;              // (the pools apply self.activation to the stored relu output, :387)
    // The bias of every column tile is requested BEFORE the first store: a.bias and a.out may alias as far as hipcc knows, so a load
    // that follows a store is kept behind it and waits with vmcnt(0) - for itself and for every store in front of it (16 such round
    // trips per tile in the loop below as it was first written).
    float bvv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bvv[nt] = a.bias[n0 + (nt < nvalid ? nt : 0) * 16 + r];
    // INTERIOR: every column tile live, every row below Mtot (all tiles but those of the last row / column block) - no predicate on a
    // store.  PLAIN: the pool activation is the identity on [0, inf) (relu / prelu / elu) - no activation switch per element.
    // An element is (wave-uniform tile base)[32-bit offset]: no 64-bit arithmetic per element.
    int64_t mw = m0 + wave * (16 * RM);
    float* otile = a.out + ((int64_t)__builtin_amdgcn_readfirstlane((int)((mw * Pp) >> 32)) << 32 |
                            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(mw * Pp)));
    const uint32_t olane = (uint32_t)(kk * 4 * Pp + n0 + r);
    auto epilogue = [&](auto interior, auto plain) {
        constexpr bool IN = decltype(interior)::value, PL = decltype(plain)::value;
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) {
            unsigned long long mine = 0;                       // relu mask (a.relu): lane (kk, r = 4*(nt & 3) + j) keeps the ballot of (nt, j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (IN || nt < nvalid) {
                    const float bv = bvv[nt];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float c = fmaxf(acc[rm][nt][j] + bv, 0.f);                   // CFFM.py:478
                        if (IN || mw + rm * 16 + kk * 4 + j < a.Mtot) otile[olane + (uint32_t)((rm * 16 + j) * Pp + nt * 16)] = c;
                        psum[rm][j] += PL ? c : act_pos(c, a.act);     // padded channels: zero filter and bias -> act(0) = 0
                        if (!GEN) {
                            const unsigned long long bal = __ballot(c > 0.f);   // bits 16kk..16kk+15: the 16 channels of row (kk, j)
                            if (r == 4 * (nt & 3) + j) mine = bal;
                        }
                    }
                }
                if (!GEN && a.relu != nullptr && ((nt & 3) == 3 || nt == NT - 1)) {   // four column tiles collected: one 2-byte store per lane
                    const int ntw = (nt & ~3) + (r >> 2);
                    const int64_t m = mw + rm * 16 + kk * 4 + (r & 3);
                    if (ntw <= nt && (IN || (ntw < nvalid && m < a.Mtot))) a.relu[m * (Pp >> 4) + (n0 >> 4) + ntw] = (uint16_t)(mine >> (16 * kk));
                }
            }
        }
    };
    const bool plain_act = GEN || (a.act != CFFM_ACT_SELU && a.act != CFFM_ACT_GELU);
    if (nvalid == NT && m0 + BM <= a.Mtot && plain_act) epilogue(std::true_type(), std::true_type());
    else epilogue(std::false_type(), std::false_type());
    (void)pact;
    if (a.pool != nullptr) {
        // s_{l+1}[b][y] partial of this column block: columns over the 16 lanes of a DPP row, then the So rows (x) of one y in
        // row order out of LDS (the filter tiles are dead; So <= 64 <= BM and both are powers of two: a y never straddles tiles)
        float* rowsum = Ws;                                    // [BM]
        __syncthreads();                                       // every wave is done with the filter tiles
#pragma unroll
        for (int rm = 0; rm < RM; ++rm)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = row_group_sum<true>(psum[rm][j]);
                if (r == 0) rowsum[wave * (16 * RM) + rm * 16 + kk * 4 + j] = v;
            }
        __syncthreads();
        const int ny = BM >> a.lgSo;                           // (b, y) groups of this row tile
        if (tid < ny) {
            const int64_t m = m0 + ((int64_t)tid << a.lgSo);
            if (m < a.Mtot) {
                float v = 0.f;
                for (int x = 0; x < So; ++x) v += rowsum[(tid << a.lgSo) + x];
                a.pool[(m >> a.lgSo) * a.pool_np + (n0 / BN)] = v;          // m >> lgSo = b * So + y
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// dgrad
// -------------------------------------------------------------------------------------------------
struct DgradArgs {
    const float* dC;     // grad wrt C_l [B,So,So,Pp]
    const float* W;      // [4P][P]
    const float* Cprev;  // L0=false: C_{l-1} [B,Sin,Sin,Pp];  L0=true: Eo [B,F,D]
    const float* dt1;    // [B, t1w]
    float* dprev;        // L0=false: dC_{l-1};  L0=true: dEo [B,F,D]
    int64_t Mtot;
    int B, lgSo, P, Pp, F, D, act, t1w, t1off;   // t1off: offset of this layer's input pool inside t1
    int nblk;            // dgrad_kernel, L0 = false: column blocks of a row tile (1-D grid, see xcd_tile)
    const int32_t* idx = nullptr;   // tiled layer 0 only: non-NULL = Cprev is the outer TABLE [M][D], row (b, f) = idx[b*F+f] (RowSrc)
    int idxM = 0;
    int idxStride = 0;              // floats between rows of the table (0 = D): see RowSrc
    const uint16_t* relu = nullptr; // dgrad_kernel, L0 = false: bit mask of C_{l-1} > 0 ([rows of C_{l-1}][Pp/16] words) read INSTEAD of C_{l-1}
    void* wb3 = nullptr;            // bf16x3 instance: scratch for the pre-split (transposed) filter image (ws.wb3)
    int wb3_npad = 0;
};

// 3 wavefronts per SIMD for the 128 x 128 instance (166 VGPRs instead of 200, nothing spilled): 16.36 -> 14.80 ms per launch at
// the stress shape.  (wgrad2_kernel<8> at the same bound spills 189 registers and stays at 2.)
template <int NT, int RM, bool L0, bool B3 = false>
__global__ __launch_bounds__(256, (NT == 8 && RM == 2 && !L0) ? 3 : 1) void dgrad_kernel(DgradArgs a) {
    constexpr int BN = NT * 16, LDW = BN + 4, BM = 64 * RM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(!B3 || !L0, "the bf16x3 loop serves the direct layers");
    float* Ws = reinterpret_cast<float*>(smem);
    uint32_t* lut = reinterpret_cast<uint32_t*>(Ws + (B3 ? gemm_b3_lds_bytes<NT>() / 4 : 2 * KSTEP * LDW));   // L0 only from here on
    float* Es = reinterpret_cast<float*>(lut + a.Pp);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kk = lane >> 4;
    const int So = 1 << a.lgSo, Sin = 2 * So, Pp = a.Pp, P = a.P, Dp = a.D + 1, S2 = So * So;
    const float invPp = 1.f / (float)Pp;
    const int Ntot = 4 * Pp;
    const int rows_per_wg = L0 ? (S2 > BM ? S2 : BM) : BM;
    const int n_ex = L0 ? rows_per_wg / S2 : 0;
    const int mtiles = rows_per_wg / BM;
    const int nblocks = L0 ? (Ntot + BN - 1) / BN : 1;
    // L0 = false: the nblk column blocks of one row tile read the same dC rows: consecutive blocks of one XCD
    const int nby = L0 ? 1 : a.nblk;
    const int64_t mtiles_wg = (a.Mtot + rows_per_wg - 1) / rows_per_wg;
    const int64_t tile = xcd_tile(xcd_per(mtiles_wg) * nby);
    if (tile >= mtiles_wg * nby) return;
    const int64_t wg_m0 = (tile / nby) * rows_per_wg;
    const int by = (int)(tile % nby);
    const int b0 = (int)(wg_m0 >> (2 * a.lgSo));
    const int exsz = a.F * Dp;
    float* dEw = Es + n_ex * exsz + wave * (n_ex * exsz);     // this wave's private accumulators
    float* rs = Es + 5 * n_ex * exsz;                          // [n_ex][F] row sums, then [n_ex][F] dots
    const bool fast = L0 && a.lgSo >= 4;   // a 16-row tile is 16 consecutive x of ONE (b, y)
    if (L0) {
        build_pair_lut(lut, a.F, Pp);
        stage_examples(Es, a.Cprev, b0, n_ex, a.B, a.F, a.D, Dp);
        for (int e = tid; e < 4 * n_ex * exsz; e += 256) Es[n_ex * exsz + e] = 0.f;
        __syncthreads();
    }

    for (int nb = 0; nb < nblocks; ++nb) {
        const int n0 = (L0 ? nb : by) * BN;
        const int nvalid = min(NT, (Ntot - n0) / 16);
        // fast path, j-side: dEo[j_p][2x+dw] sums over y and dh -> keep it in registers across the m tiles
        float accj[RM][NT][4];
#pragma unroll
        for (int rm = 0; rm < RM; ++rm)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) accj[rm][nt][j] = 0.f;

        for (int mt = 0; mt < mtiles; ++mt) {
            const int64_t m0 = wg_m0 + (int64_t)mt * BM;
            int64_t arow[RM];
#pragma unroll
            for (int rm = 0; rm < RM; ++rm) {
                int64_t m = m0 + wave * (16 * RM) + rm * 16 + r;
                if (m >= a.Mtot) m = a.Mtot - 1;
                arow[rm] = m * Pp + 4 * kk;
            }
            auto loadA = [&](int ks, float4 (&reg)[RM][2]) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = ks * KSTEP + 16 * h;
#pragma unroll
                    for (int rm = 0; rm < RM; ++rm)       // k >= Pp only in the unused second half of an odd last step
                        reg[rm][h] = *reinterpret_cast<const float4*>(a.dC + arow[rm] + (k < Pp ? k : Pp - 16));
                }
            };
            const WSpec wspec = {a.W, Pp, Pp, Ntot, n0, B3 ? a.wb3 : nullptr, a.wb3_npad};        // W^T tile: (k = q, n) = W[n][q], 16-byte reads along q
            f32x4 acc[RM][NT];
#pragma unroll
            for (int rm = 0; rm < RM; ++rm)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[rm][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (B3) gemm_tile_b3<NT, RM, true, false>(acc, Pp / 16, Ws, wspec, loadA);
            else gemm_tile<NT, RM, true>(acc, Pp / 16, Ws, wspec, loadA);

            // ---- epilogue ------------------------------------------------------------------------
            if (!L0) {
                // scatter-store fused with the broadcast sum-pool gradient and the relu/act mask (patches never overlap).
                // Everything that depends only on the column (tap, p) or only on the row (b, y, x) is computed once:
                // the per-element work is one add, one load of C_{l-1}, one mask and one store.
                int noff[NT], ndh[NT];
                // d act(relu(z)) / dz through c = relu(z) > 0: 1 for relu / prelu / elu, the scale for selu; gelu needs c
                const float gsc = a.act == CFFM_ACT_SELU ? CFFM_SELU_SCALE : 1.f;
                const bool gel = a.act == CFFM_ACT_GELU;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = n0 + nt * 16 + r, tap = fast_div(n, invPp), p = n - tap * Pp;
                    ndh[nt] = tap >> 1;
                    noff[nt] = ((tap >> 1) * Sin + (tap & 1)) * Pp + p;
                }
                if (a.relu != nullptr) {
                    // The mask path of the wide shapes, one ROW of the tile (this lane's NT column tiles) at a time and one row AHEAD:
                    // the row's NT mask words and its two pool-gradient values are requested together, and the requests of row i + 1
                    // are issued BEFORE the stores of row i.  (Element by element - load the word, wait, store, load the next word -
                    // every load waited with vmcnt(0), which on this ISA also waits for the store issued just before it: 64 exposed
                    // memory round trips per tile and lane, profiles/r04_dgrad_epilogue.md.)  One 16-bit word per (pixel, 16 channels):
                    // the 16 lanes of a row share it; Pp is a multiple of 16, so lane r is channel (word, r).
                    // Addresses: everything that depends on the column tile only is wave-uniform (a group of 16 columns never straddles
                    // a tap: Pp is a multiple of 16) and lives in scalar registers; a row's position is kept RELATIVE to the first row
                    // of this wavefront (positions grow with m, a wavefront's 16 * RM rows span < 2^31 floats), so that an element is
                    // (scalar base)[32-bit offset] - no 64-bit arithmetic per element, 11 registers per row in flight.
                    int ucol[NT], uword[NT], udh[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int nb = n0 + (nt < nvalid ? nt : 0) * 16;             // a dead column tile repeats tile 0 (never stored)
                        const int tap = nb / Pp, p0 = nb - tap * Pp;
                        udh[nt] = __builtin_amdgcn_readfirstlane(tap >> 1);
                        ucol[nt] = __builtin_amdgcn_readfirstlane(((tap >> 1) * Sin + (tap & 1)) * Pp + p0);
                        uword[nt] = ucol[nt] >> 4;
                    }
                    int64_t m_first = m0 + wave * (16 * RM);
                    if (m_first >= a.Mtot) m_first = a.Mtot - 1;
                    const RowPos rf = row_pos(m_first, a.lgSo);
                    const int64_t base_v = (((int64_t)rf.b * Sin + 2 * rf.y) * Sin + 2 * rf.x) * Pp;
                    const int64_t base64 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(base_v >> 32)) << 32) |
                                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base_v);
                    float* dtile = a.dprev + base64;
                    const uint16_t* wtile = a.relu + (base64 >> 4);
                    struct RowReq { uint32_t rel; bool ok; float t0, t1v; uint32_t wm[NT]; };
                    // INTERIOR (all NT column tiles live, all BM rows below Mtot - every tile but those of the last row / column block):
                    // no predicate on any store, i.e. no exec-mask branch per element
                    auto sweep = [&](auto interior) {
                        constexpr bool IN = decltype(interior)::value;
                        auto request = [&](int idx, RowReq& q) {
                            const int64_t m = m0 + wave * (16 * RM) + (idx >> 2) * 16 + kk * 4 + (idx & 3);
                            q.ok = IN || m < a.Mtot;
                            const RowPos rp = row_pos(q.ok ? m : a.Mtot - 1, a.lgSo);
                            q.rel = (uint32_t)((((int64_t)rp.b * Sin + 2 * rp.y) * Sin + 2 * rp.x) * Pp - base64);
                            const float* tp = a.dt1 + (int64_t)rp.b * a.t1w + a.t1off + 2 * rp.y;
                            q.t0 = tp[0]; q.t1v = tp[1];
                            const uint32_t wrel = q.rel >> 4;
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) q.wm[nt] = wtile[wrel + (uint32_t)uword[nt]];
                        };
                        auto finish = [&](int idx, const RowReq& q) {
                            const int rm = idx >> 2, j = idx & 3;
                            const uint32_t er = q.rel + (uint32_t)r;
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                if (!IN && nt >= nvalid) continue;
                                const float g = acc[rm][nt][j] + (udh[nt] ? q.t1v : q.t0);
                                const float d = ((q.wm[nt] >> r) & 1u) ? gsc : 0.f;
                                if (IN || q.ok) __builtin_nontemporal_store(g * d, dtile + (er + (uint32_t)ucol[nt]));
                            }
                        };
                        RowReq qa, qb;
                        request(0, qa);
#pragma unroll
                        for (int idx = 0; idx < RM * 4; idx += 2) {
                            request(idx + 1, qb);
                            __builtin_amdgcn_sched_barrier(0);
                            finish(idx, qa);
                            __builtin_amdgcn_sched_barrier(0);
                            if (idx + 2 < RM * 4) request(idx + 2, qa);
                            __builtin_amdgcn_sched_barrier(0);
                            finish(idx + 1, qb);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    };
                    if (nvalid == NT && m0 + BM <= a.Mtot) sweep(std::true_type());
                    else sweep(std::false_type());
                } else {
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) {
                    const int64_t mrow = m0 + wave * (16 * RM) + rm * 16 + kk * 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int64_t m = mrow + j;
                        if (m >= a.Mtot) continue;
                        const RowPos rp = row_pos(m, a.lgSo);
                        const int64_t rowbase = (((int64_t)rp.b * Sin + 2 * rp.y) * Sin + 2 * rp.x) * Pp;
                        const float* tp = a.dt1 + (int64_t)rp.b * a.t1w + a.t1off + 2 * rp.y;
                        const float t0 = tp[0], t1v = tp[1];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            if (nt >= nvalid) continue;
                            const int64_t pos = rowbase + noff[nt];
                            const float g = acc[rm][nt][j] + (ndh[nt] ? t1v : t0);
                            const float c = a.Cprev[pos];
                            float d = c > 0.f ? gsc : 0.f;
                            if (gel) d = c > 0.f ? act_grad_f(c, CFFM_ACT_GELU) : 0.f;
                            __builtin_nontemporal_store(g * d, &a.dprev[pos]);
                        }
                    }
                }
                }
            }
#pragma unroll
            for (int rm = 0; rm < RM; ++rm) {
                if (!L0) break;
                const int64_t mrow = m0 + wave * (16 * RM) + rm * 16 + kk * 4;
                const RowPos rq = row_pos(mrow < a.Mtot ? mrow : a.Mtot - 1, a.lgSo);   // fast path: (b, y) of the tile
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (nt >= nvalid) continue;
                    const int n = n0 + nt * 16 + r, tap = fast_div(n, invPp), p = n - tap * Pp, dh = tap >> 1, dw = tap & 1;
                    if (!L0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int64_t m = mrow + j;
                            if (m < a.Mtot) {
                                const RowPos rp = row_pos(m, a.lgSo);
                                const int64_t pos = (((int64_t)rp.b * Sin + 2 * rp.y + dh) * Sin + 2 * rp.x + dw) * Pp + p;
                                const float g = acc[rm][nt][j] + a.dt1[(int64_t)rp.b * a.t1w + a.t1off + 2 * rp.y + dh];
                                a.dprev[pos] = g * act_relu_grad(a.Cprev[pos], a.act);
                            }
                        }
                    } else {
                        const uint32_t ij = lut[p];
                        const int fi = ij & 0xffff, fj = ij >> 16;
                        const bool pv = p < P;
                        if (fast) {
                            const int eb = (rq.b - b0) * exsz;
                            const float ei = Es[eb + fi * Dp + 2 * rq.y + dh];
                            float si = 0.f;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float v = (pv && mrow + j < a.Mtot) ? acc[rm][nt][j] : 0.f;
                                si += v * Es[eb + fj * Dp + 2 * (rq.x + j) + dw];
                                accj[rm][nt][j] += v * ei;
                            }
                            si += __shfl_xor(si, 16, 64);
                            si += __shfl_xor(si, 32, 64);
                            if (kk == 0 && pv) atomicAdd(&dEw[eb + fi * Dp + 2 * rq.y + dh], si);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int64_t m = mrow + j;
                                if (pv && m < a.Mtot) {
                                    const RowPos rp = row_pos(m, a.lgSo);
                                    const int eb = (rp.b - b0) * exsz;
                                    const int io = eb + fi * Dp + 2 * rp.y + dh, jo = eb + fj * Dp + 2 * rp.x + dw;
                                    const float v = acc[rm][nt][j];
                                    atomicAdd(&dEw[io], v * Es[jo]);
                                    atomicAdd(&dEw[jo], v * Es[io]);
                                }
                            }
                        }
                    }
                }
            }
        }
        if (fast) {
            // flush the j-side: in the fast path S2 >= 256 >= BM, so the workgroup holds ONE example and the
            // x of (rm, kk, j) does not depend on the m tile
#pragma unroll
            for (int rm = 0; rm < RM; ++rm) {
                const int64_t mrow = wg_m0 + wave * (16 * RM) + rm * 16 + kk * 4;
                const RowPos rq = row_pos(mrow < a.Mtot ? mrow : a.Mtot - 1, a.lgSo);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (nt >= nvalid) continue;
                    const int n = n0 + nt * 16 + r, tap = fast_div(n, invPp), p = n - tap * Pp, dw = tap & 1;
                    if (p < P) {
                        const int fj = lut[p] >> 16;
#pragma unroll
                        for (int j = 0; j < 4; ++j) atomicAdd(&dEw[fj * Dp + 2 * (rq.x + j) + dw], accj[rm][nt][j]);
                    }
                }
            }
        }
    }
    if (L0) {
        // merge the four private copies in wave order and add the closed-form sum-pool (s0) terms:
        //   dEo[i][h] += ds0[h] * R_i + Q_i,  R_i = sum_{j>i} rowsum(Eo[j]),  Q_i = sum_{i'<i} <ds0, Eo[i']>
        __syncthreads();
        float* acc0 = Es + n_ex * exsz;
        float* dots = rs + n_ex * a.F;
        for (int e = tid; e < n_ex * a.F; e += 256) {
            const int b = b0 + e / a.F;
            float s = 0.f, d = 0.f;
            if (b < a.B)
                for (int h = 0; h < a.D; ++h) {
                    const float v = Es[e * Dp + h];
                    s += v;
                    d += v * a.dt1[(int64_t)b * a.t1w + h];
                }
            rs[e] = s; dots[e] = d;
        }
        __syncthreads();
        const int per = a.F * a.D;
        const float invD = 1.f / (float)a.D, invF = 1.f / (float)a.F;
        for (int e = tid; e < n_ex * per; e += 256) {
            const int row = fast_div(e, invD), h = e - row * a.D, ex = fast_div(row, invF), f = row - ex * a.F, b = b0 + ex;
            if (b >= a.B) continue;
            float R = 0.f, Q = 0.f;
            for (int j = f + 1; j < a.F; ++j) R += rs[ex * a.F + j];
            for (int i = 0; i < f; ++i) Q += dots[ex * a.F + i];
            const int o = row * Dp + h, st = n_ex * exsz;
            const float conv = ((acc0[o] + acc0[st + o]) + acc0[2 * st + o]) + acc0[3 * st + o];
            a.dprev[(int64_t)b0 * per + e] = conv + a.dt1[(int64_t)b * a.t1w + h] * R + Q;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// wgrad: dW[(tap,p)][q] = sum_m A'[m][(tap,p)] * dC[m][q],  db[q] = sum_m dC[m][q]
// grid = (i-blocks * q-blocks, CFFM_NSLAB); slab s reduces a contiguous chunk of the M rows.
// -------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* in;     // GEN: Eo [B,F,D]; else C_{l-1} [B,Sin,Sin,Pp]
    const float* dC;     // [Mtot][Pp]
    float* slabW;        // slab 0 of conv_w[l]
    float* slabB;        // slab 0 of conv_b[l]
    int64_t slab_stride, slabB_stride, Mtot;
    int B, lgSo, P, Pp, F, D, act, qblocks;
    int nslab, nxy;      // wgrad_kernel: gradient slabs and output tiles per slab (1-D grid, see xcd_tile)
    const int32_t* idx = nullptr;   // tiled layer 0 only: non-NULL = `in` is the outer TABLE [M][D], row (b, f) = idx[b*F+f] (RowSrc)
    int idxM = 0;
    int idxStride = 0;              // floats between rows of the table (0 = D): see RowSrc
};

// Generic form (64 x 16*NT output tile per workgroup): the direct layer 0 (GEN, F >= 33) and column-tile counts other than
// 6 / 8; the layers >= 1 of the wide shapes run wgrad2_kernel below.
template <int NT, bool GEN>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs a) {
    constexpr int RI = 1;                                            // row tiles per wavefront
    constexpr int BI = 64 * RI, LDA = BI + 16, BQ = NT * 16, LDB = BQ + ((NT & 1) ? 0 : 16), KM = WG_KM;
    constexpr int NB = KM * BQ / 256, NA = KM * BI / 4 / 256;     // per-thread B' floats / A' float4s per step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Bs = reinterpret_cast<float*>(smem);                       // [KM][LDB]
    float* As = Bs + KM * LDB;                                        // [KM][LDA]      (!GEN)
    uint32_t* lut = reinterpret_cast<uint32_t*>(As + (GEN ? 0 : KM * LDA));   // [Pp]  (GEN)
    float* Es = reinterpret_cast<float*>(lut + (GEN ? a.Pp : 0));     // [n_ex][F][Dp] (GEN)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kk = lane >> 4;
    const int So = 1 << a.lgSo, Sin = 2 * So, Pp = a.Pp, P = a.P, Dp = a.D + 1, S2 = So * So;
    const float invPp = 1.f / (float)Pp;
    // all output tiles of ONE slab read the same rows of A' and dC: they run as consecutive blocks of one XCD, which
    // then works through its slabs one after the other (the operands reach the XCD's L2 once per slab)
    const int64_t tile = xcd_tile(xcd_per(a.nslab) * a.nxy);
    if (tile >= (int64_t)a.nslab * a.nxy) return;
    const int slab = (int)(tile / a.nxy), bxy = (int)(tile % a.nxy);
    const int ib = bxy / a.qblocks, qb = bxy - ib * a.qblocks;
    const int i0 = ib * BI, q0 = qb * BQ;
    const int nvalid = min(NT, (Pp - q0) / 16);
    const int64_t nsteps = (a.Mtot + KM - 1) / KM;
    const int64_t cps = (nsteps + a.nslab - 1) / a.nslab;
    const int64_t s_lo = slab * cps, s_hi = min(nsteps, s_lo + cps);
    const int n_ex = GEN ? (KM > S2 ? KM / S2 : 1) : 0;

    // the (tap, p) this lane's A' row belongs to (GEN only: RI == 1)
    const int irow = i0 + wave * 16 + r;
    const int tapA = fast_div(irow, invPp), pA = irow - tapA * Pp, dhA = tapA >> 1, dwA = tapA & 1;
    int fi = 0, fj = 0;
    if (GEN) {
        build_pair_lut(lut, a.F, Pp);
        __syncthreads();
        const uint32_t ij = lut[pA];
        fi = ij & 0xffff; fj = ij >> 16;
    }
    // staging geometry of this thread (fixed over the steps): BI/4 16-byte pieces per A' row
    int a_tap[NA], a_p[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int ii = i0 + 4 * ((tid + 256 * i) % (BI / 4));
        a_tap[i] = fast_div(ii, invPp);                          // 4 for the rows of a last, partial tile beyond 4*Pp: zeros
        a_p[i] = ii - a_tap[i] * Pp;
    }
    f32x4 acc[RI][NT];
#pragma unroll
    for (int ri = 0; ri < RI; ++ri)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[ri][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    int cur_b = -1;
    float breg[NB];
    float4 areg[NA > 0 ? NA : 1];

    auto fetch = [&](int64_t st) {            // global -> registers for step st
        const int64_t mbase = st * KM;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = tid + 256 * i, row = e / BQ, c = e % BQ;
            const int64_t m = mbase + row;
            breg[i] = (m < a.Mtot && q0 + c < Pp) ? a.dC[m * Pp + q0 + c] : 0.f;
        }
        if (!GEN) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int64_t m = mbase + (tid + 256 * i) / (BI / 4);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < a.Mtot && a_tap[i] < 4) {
                    const RowPos rp = row_pos(m, a.lgSo);
                    const int64_t pos = (((int64_t)rp.b * Sin + 2 * rp.y + (a_tap[i] >> 1)) * Sin + 2 * rp.x + (a_tap[i] & 1)) * Pp + a_p[i];
                    v = *reinterpret_cast<const float4*>(a.in + pos);
                }
                areg[i] = v;
            }
        }
    };

    if (s_lo < s_hi) fetch(s_lo);
    for (int64_t st = s_lo; st < s_hi; ++st) {
        const int64_t mbase = st * KM;
        __syncthreads();                                   // previous step's LDS reads are done
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = tid + 256 * i;
            Bs[(e / BQ) * LDB + (e % BQ)] = breg[i];
        }
        if (!GEN) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int u = tid + 256 * i;
                float4 v = areg[i];
                v.x = act_pos(v.x, a.act); v.y = act_pos(v.y, a.act);
                v.z = act_pos(v.z, a.act); v.w = act_pos(v.w, a.act);
                *reinterpret_cast<float4*>(&As[(u / (BI / 4)) * LDA + 4 * (u % (BI / 4))]) = v;
            }
        } else {
            const int bl = (int)(mbase >> (2 * a.lgSo));
            if (bl != cur_b) {
                stage_examples(Es, a.in, bl, n_ex, a.B, a.F, a.D, Dp);
                cur_b = bl;
            }
        }
        __syncthreads();
        if (st + 1 < s_hi) fetch(st + 1);                 // in flight while this step's MFMAs run
        // ---- MFMA over the KM rows ------------------------------------------------------------------
#pragma unroll 2
        for (int ks4 = 0; ks4 < KM; ks4 += 4) {
            float av[RI];
            if (GEN) {
                int64_t m = mbase + ks4 + kk;
                if (m >= a.Mtot) m = a.Mtot - 1;             // B' rows beyond Mtot are zero
                const RowPos rp = row_pos(m, a.lgSo);
                const int eb = (rp.b - cur_b) * a.F * Dp;
                av[0] = pA < P ? Es[eb + fi * Dp + 2 * rp.y + dhA] * Es[eb + fj * Dp + 2 * rp.x + dwA] : 0.f;
            } else {
#pragma unroll
                for (int ri = 0; ri < RI; ++ri) av[ri] = As[(ks4 + kk) * LDA + (wave * RI + ri) * 16 + r];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (nt >= nvalid) continue;
                const float bv = Bs[(ks4 + kk) * LDB + nt * 16 + r];
#pragma unroll
                for (int ri = 0; ri < RI; ++ri) acc[ri][nt] = mfma16(av[ri], bv, acc[ri][nt]);
            }
        }
        if (ib == 0 && tid < BQ) {
#pragma unroll 8
            for (int row = 0; row < KM; ++row) bsum += Bs[row * LDB + tid];
        }
    }
    // ---- write this slab (every element of the parameter range, zeros included) ---------------------
    float* sw = a.slabW + (int64_t)slab * a.slab_stride;
#pragma unroll
    for (int ri = 0; ri < RI; ++ri)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nt >= nvalid) continue;
            const int q = q0 + nt * 16 + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + (wave * RI + ri) * 16 + kk * 4 + j;
                if (i < 4 * Pp) sw[(int64_t)i * Pp + q] = acc[ri][nt][j];
            }
        }
    if (ib == 0 && tid < BQ && q0 + tid < Pp) a.slabB[(int64_t)slab * a.slab_stride + q0 + tid] = bsum;
}

// wgrad2: the weight / bias gradient of a conv layer >= 1 for WIDE filters (Pp > 64), same contraction and same split-K
// slabs as wgrad_kernel<.., GEN = false>, restructured after the rocprofv3 counters of round 2 (MFMA pipe 40 % busy, one
// exposed LDS round trip per two MFMAs: every B fragment was a ds_read_b32 followed by its own wait and a branch):
//   * both operands are fetched in 16-byte pieces as 4 x 4 blocks (4 consecutive reduction rows m x 4 consecutive
//     channels), transposed in registers and staged as [m/4][channel][4 m] records - the layout gemm_tile uses - so that a
//     lane's fragments for FOUR MFMAs come back with one ds_read_b128 (a float4 = 4 consecutive m; A' and dC use the same
//     k permutation);
//   * every tile of the wave is computed (columns beyond Pp and rows beyond 4*Pp are staged as zeros; the slab write skips
//     them): no branch and no wait sits between the MFMAs of a 16-row chunk;
//   * out-of-range rows / columns are read from a clamped address and zeroed by a select (no exec-mask branches).
// Output tile 128 x (16*NT) per workgroup, wave w owns rows [32 w, 32 w + 32): acc[2][NT].
template <int NT>
__global__ __launch_bounds__(256, 2) void wgrad2_kernel(WgradArgs a) {
    constexpr int RI = 2, BI = 64 * RI, BQ = NT * 16, KM = WG_KM, MQ = KM / 4;
    constexpr int NBA = MQ * (BI / 4) / 256, NBB = (MQ * (BQ / 4) + 255) / 256;   // 4x4 blocks per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                       // [MQ][BI][4]
    float* Bs = As + MQ * BI * 4;                                     // [MQ][BQ][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kk = lane >> 4;
    const int So = 1 << a.lgSo, Sin = 2 * So, Pp = a.Pp;
    const float invPp = 1.f / (float)Pp;
    const int64_t tile = xcd_tile(xcd_per(a.nslab) * a.nxy);
    if (tile >= (int64_t)a.nslab * a.nxy) return;
    const int slab = (int)(tile / a.nxy), bxy = (int)(tile % a.nxy);
    const int ib = bxy / a.qblocks, qb = bxy - ib * a.qblocks;
    const int i0 = ib * BI, q0 = qb * BQ;
    const int nvalid = min(NT, (Pp - q0) / 16);
    const int64_t nsteps = (a.Mtot + KM - 1) / KM;
    const int64_t cps = (nsteps + a.nslab - 1) / a.nslab;
    const int64_t s_lo = slab * cps, s_hi = min(nsteps, s_lo + cps);

    // staging geometry of this thread (fixed over the steps)
    int a_off[NBA], a_mq[NBA], a_i4[NBA];                         // A': float offset of (tap, p) inside a patch, or -1 (zeros)
#pragma unroll
    for (int j = 0; j < NBA; ++j) {
        const int bb = tid + 256 * j;
        a_mq[j] = bb / (BI / 4); a_i4[j] = bb % (BI / 4);
        const int ii = i0 + 4 * a_i4[j], tap = fast_div(ii, invPp), p = ii - tap * Pp;
        a_off[j] = tap < 4 ? ((tap >> 1) * Sin + (tap & 1)) * Pp + p : -1;
    }
    f32x4 acc[RI][NT];
#pragma unroll
    for (int ri = 0; ri < RI; ++ri)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[ri][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    float4 areg[NBA][4], breg[NBB][4];

    auto fetch = [&](int64_t st) {            // global -> registers for step st
        const int64_t mbase = st * KM;
#pragma unroll
        for (int j = 0; j < NBA; ++j) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int64_t m = mbase + 4 * a_mq[j] + t;
                const bool ok = m < a.Mtot && a_off[j] >= 0;
                const RowPos rp = row_pos(m < a.Mtot ? m : a.Mtot - 1, a.lgSo);
                const int64_t pos = (((int64_t)rp.b * Sin + 2 * rp.y) * Sin + 2 * rp.x) * Pp + (a_off[j] >= 0 ? a_off[j] : 0);
                const float4 v = *reinterpret_cast<const float4*>(a.in + pos);
                areg[j][t] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int j = 0; j < NBB; ++j) {
            const int bb = tid + 256 * j, mq = bb / (BQ / 4), c4 = bb % (BQ / 4);
            const int q = q0 + 4 * c4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int64_t m = mbase + 4 * mq + t;
                const bool ok = (MQ * (BQ / 4) % 256 == 0 || bb < MQ * (BQ / 4)) && m < a.Mtot && q < Pp;
                const int64_t mc = m < a.Mtot ? m : a.Mtot - 1;
                const float4 v = *reinterpret_cast<const float4*>(a.dC + mc * Pp + (q < Pp ? q : Pp - 4));
                breg[j][t] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    // 4 x 4 block (rows t = 4 consecutive m, columns = 4 consecutive channels) -> 4 records (one per channel) of 4 m each
    auto put = [&](float* dst, const float4 (&b)[4]) {
        *reinterpret_cast<float4*>(dst) = make_float4(b[0].x, b[1].x, b[2].x, b[3].x);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(b[0].y, b[1].y, b[2].y, b[3].y);
        *reinterpret_cast<float4*>(dst + 8) = make_float4(b[0].z, b[1].z, b[2].z, b[3].z);
        *reinterpret_cast<float4*>(dst + 12) = make_float4(b[0].w, b[1].w, b[2].w, b[3].w);
    };

    if (s_lo < s_hi) fetch(s_lo);
    for (int64_t st = s_lo; st < s_hi; ++st) {
        __syncthreads();                                   // previous step's LDS reads are done
        if (a.act != CFFM_ACT_RELU && a.act != CFFM_ACT_PRELU && a.act != CFFM_ACT_ELU) {
#pragma unroll
            for (int j = 0; j < NBA; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) act_pos4(areg[j][t], a.act);     // A' = act(C_{l-1}), C >= 0
        }
#pragma unroll
        for (int j = 0; j < NBA; ++j) put(As + (a_mq[j] * BI + 4 * a_i4[j]) * 4, areg[j]);
#pragma unroll
        for (int j = 0; j < NBB; ++j) {
            const int bb = tid + 256 * j, mq = bb / (BQ / 4), c4 = bb % (BQ / 4);
            if (MQ * (BQ / 4) % 256 == 0 || bb < MQ * (BQ / 4)) put(Bs + (mq * BQ + 4 * c4) * 4, breg[j]);
        }
        __syncthreads();
        if (st + 1 < s_hi) fetch(st + 1);                 // in flight while this step's MFMAs run
        // ---- MFMA: 4 chunks of 16 reduction rows; lane (r, kk) takes record m/4 = 4 * chunk + kk ---------------------
#pragma unroll
        for (int ch = 0; ch < MQ / 4; ++ch) {
            float4 af[RI], bf[NT];
#pragma unroll
            for (int ri = 0; ri < RI; ++ri)
                af[ri] = *reinterpret_cast<const float4*>(As + ((4 * ch + kk) * BI + (wave * RI + ri) * 16 + r) * 4);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                bf[nt] = *reinterpret_cast<const float4*>(Bs + ((4 * ch + kk) * BQ + nt * 16 + r) * 4);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float bv = t == 0 ? bf[nt].x : t == 1 ? bf[nt].y : t == 2 ? bf[nt].z : bf[nt].w;
#pragma unroll
                    for (int ri = 0; ri < RI; ++ri) {
                        const float av = t == 0 ? af[ri].x : t == 1 ? af[ri].y : t == 2 ? af[ri].z : af[ri].w;
                        acc[ri][nt] = mfma16(av, bv, acc[ri][nt]);
                    }
                }
        }
        if (ib == 0 && tid < BQ) {                         // db[q] += sum over the KM rows, in row order
#pragma unroll 4
            for (int mq = 0; mq < MQ; ++mq) {
                const float4 v = *reinterpret_cast<const float4*>(Bs + (mq * BQ + tid) * 4);
                bsum += v.x; bsum += v.y; bsum += v.z; bsum += v.w;
            }
        }
    }
    // ---- write this slab (every element of the parameter range, zeros included) ---------------------
    float* sw = a.slabW + (int64_t)slab * a.slab_stride;
#pragma unroll
    for (int ri = 0; ri < RI; ++ri)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nt >= nvalid) continue;
            const int q = q0 + nt * 16 + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + (wave * RI + ri) * 16 + kk * 4 + j;
                if (i < 4 * Pp) sw[(int64_t)i * Pp + q] = acc[ri][nt][j];
            }
        }
    if (ib == 0 && tid < BQ && q0 + tid < Pp) a.slabB[(int64_t)slab * a.slab_stride + q0 + tid] = bsum;
}

// wgrad3: the same weight gradient on the bf16 MFMA pipe at fp32 accuracy (bf16x3, see gemm_tile_b3): both operands are split
// without error into three bf16 pieces WHILE THEY ARE STAGED, the LDS images hold [piece][m-octet][channel] records of 8 bf16
// (8 consecutive reduction rows of one channel: one ds_read_b128 = the operand of one v_mfma_f32_16x16x32_bf16), and the six
// cross terms of weight >= 2^-16 are accumulated in fp32.  One 32-row step per barrier pair (48 KB of LDS: three workgroups per
// CU cover each other's staging); waves 0-1 stage A' = act(C_{l-1}) patches, waves 2-3 stage dC and keep the bias gradient.
template <int NT>
__global__ __launch_bounds__(256, 3) void wgrad3_kernel(WgradArgs a) {
    constexpr int RI = 2, BI = 64 * RI, BQ = NT * 16, KM = 32, NO = KM / 8;
    static_assert(BI / 4 * NO == 128 && BQ / 4 * NO <= 128, "one 8-row x 4-channel block per staging thread");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4_t* As = reinterpret_cast<u32x4_t*>(smem);                   // [3][NO][BI]
    u32x4_t* Bs = As + 3 * NO * BI;                                   // [3][NO][BQ]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, kk = lane >> 4;
    const int So = 1 << a.lgSo, Sin = 2 * So, Pp = a.Pp;
    const float invPp = 1.f / (float)Pp;
    const int64_t tile = xcd_tile(xcd_per(a.nslab) * a.nxy);
    if (tile >= (int64_t)a.nslab * a.nxy) return;
    const int slab = (int)(tile / a.nxy), bxy = (int)(tile % a.nxy);
    const int ib = bxy / a.qblocks, qb = bxy - ib * a.qblocks;
    const int i0 = ib * BI, q0 = qb * BQ;
    const int nvalid = min(NT, (Pp - q0) / 16);
    const int64_t nsteps = (a.Mtot + KM - 1) / KM;
    const int64_t cps = (nsteps + a.nslab - 1) / a.nslab;
    const int64_t s_lo = slab * cps, s_hi = min(nsteps, s_lo + cps);

    // staging role of this thread (fixed over the steps): an 8-row x 4-channel block of A' (waves 0, 1) or of dC (waves 2, 3)
    const bool stA = wave < 2;
    const int pb = tid & 127;
    const int oct = stA ? pb / (BI / 4) : pb / (BQ / 4), c4 = stA ? pb % (BI / 4) : pb % (BQ / 4);
    const bool stOn = stA || pb < NO * (BQ / 4);
    int a_off = -1;                                                   // A': float offset of (tap, p) inside a patch, or -1 (zeros)
    if (stA) {
        const int ii = i0 + 4 * c4, tap = fast_div(ii, invPp), p = ii - tap * Pp;
        a_off = tap < 4 ? ((tap >> 1) * Sin + (tap & 1)) * Pp + p : -1;
    }
    const int qcol = q0 + 4 * c4;                                     // dC: first channel of the block
    const int rbase = (r & 3) * 32 + (r >> 2);                         // MFMA lane r reads channel 16 x + r at record rbase + 4 (x ^ (r & 3))
    f32x4 acc[RI][NT];
#pragma unroll
    for (int ri = 0; ri < RI; ++ri)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[ri][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    float4 reg[8];

    // The eight rows of an octet are consecutive m (m0 a multiple of 8).  Where an octet cannot straddle anything - So >= 8: same
    // (b, y), x .. x + 7; Mtot a multiple of 8: all of it inside or all of it outside - its eight addresses are ONE base + t * pitch
    // and its validity one flag per thread (the per-row form costs a 64-bit position and a clamp per row: ~200 of the ~800
    // VALU instructions a staging wavefront ran per step, next to 96 MFMAs).
    const bool octA = a.lgSo >= 3, octB = (a.Mtot & 7) == 0;
    auto fetch = [&](int64_t st) {            // global -> registers for step st (unconditional loads from clamped addresses)
        const int64_t mbase = st * KM + 8 * oct;
        if (stA && octA) {
            const bool ok = mbase < a.Mtot && a_off >= 0;
            const RowPos rp = row_pos(mbase < a.Mtot ? mbase : 0, a.lgSo);
            const float* src = a.in + ((((int64_t)rp.b * Sin + 2 * rp.y) * Sin + 2 * rp.x) * Pp + (a_off >= 0 ? a_off : 0));
#pragma unroll
            for (int t = 0; t < 8; ++t) reg[t] = *reinterpret_cast<const float4*>(src + (int64_t)t * 2 * Pp);
            if (!ok) {
#pragma unroll
                for (int t = 0; t < 8; ++t) reg[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            return;
        }
        if (!stA && octB) {
            const bool ok = stOn && mbase < a.Mtot && qcol < Pp;
            const float* src = a.dC + ((mbase < a.Mtot ? mbase : 0) * Pp + (qcol < Pp ? qcol : Pp - 4));
#pragma unroll
            for (int t = 0; t < 8; ++t) reg[t] = *reinterpret_cast<const float4*>(src + (int64_t)t * Pp);
            if (!ok) {
#pragma unroll
                for (int t = 0; t < 8; ++t) reg[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int64_t m = mbase + t;
            const int64_t mc = m < a.Mtot ? m : a.Mtot - 1;
            float4 v;
            bool ok;
            if (stA) {
                const RowPos rp = row_pos(mc, a.lgSo);
                const int64_t pos = (((int64_t)rp.b * Sin + 2 * rp.y) * Sin + 2 * rp.x) * Pp + (a_off >= 0 ? a_off : 0);
                v = *reinterpret_cast<const float4*>(a.in + pos);
                ok = m < a.Mtot && a_off >= 0;
            } else {
                v = *reinterpret_cast<const float4*>(a.dC + mc * Pp + (qcol < Pp ? qcol : Pp - 4));
                ok = stOn && m < a.Mtot && qcol < Pp;
            }
            reg[t] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // the 8 x 4 block -> per channel the 8 reduction rows as three 8-bf16 records
    auto stage = [&]() {
        if (stA && a.act != CFFM_ACT_RELU && a.act != CFFM_ACT_PRELU && a.act != CFFM_ACT_ELU) {
#pragma unroll
            for (int t = 0; t < 8; ++t) act_pos4(reg[t], a.act);                  // A' = act(C_{l-1}), C >= 0
        }
        if (!stA && ib == 0) {                               // db[q] is written by the ib == 0 tiles only
#pragma unroll
            for (int t = 0; t < 8; ++t) { bsum[0] += reg[t].x; bsum[1] += reg[t].y; bsum[2] += reg[t].z; bsum[3] += reg[t].w; }
        }
        // Row image (BI or BQ records): channel (c4, ch) = 4 c4 + ch sits at record ch * 32 + (c4 ^ (ch << 2)).  Written as it comes -
        // a thread's four channels side by side - the 16-byte stores of consecutive lanes lie 64 bytes apart: 57 % of the LDS
        // cycles of this kernel were bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r04_syn1m_pmc_*).  In this
        // image store ch of lanes c4 = 0 .. 31 is a permutation of one contiguous 512-byte plane, and the read of 16 consecutive
        // channels (lane r: ch = r & 3, c4 = 4 x + (r >> 2)) touches the sixteen 4-bank groups 16 (r & 3 ^ x & 3) + 4 (r >> 2) once each.
        static_assert(BI == 128 && BQ == 128, "four channel planes of 32 records");
        u32x4_t* dst = (stA ? As : Bs) + oct * 128;
        const int pstride = NO * 128;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            float v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = ch == 0 ? reg[t].x : ch == 1 ? reg[t].y : ch == 2 ? reg[t].z : reg[t].w;
            u32x4_t p[3];
            split8_bf16x3(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), p);
            const int pos = ch * 32 + (c4 ^ (ch << 2));
            if (stOn) { dst[pos] = p[0]; dst[pstride + pos] = p[1]; dst[2 * pstride + pos] = p[2]; }
        }
    };

    if (s_lo < s_hi) fetch(s_lo);
    for (int64_t st = s_lo; st < s_hi; ++st) {
        __syncthreads();                                   // previous step's LDS reads are done
        stage();
        __syncthreads();
        if (st + 1 < s_hi) fetch(st + 1);                 // in flight while this step's MFMAs run
        // ---- MFMA: one 32-deep group; lane (r, kk) takes m-octet kk -----------------------------------------------------
        u32x4_t af[RI][3];
#pragma unroll
        for (int ri = 0; ri < RI; ++ri)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) af[ri][pc] = As[(pc * NO + kk) * 128 + rbase + 4 * ((wave * RI + ri) ^ (r & 3))];
        static_assert(NT % 2 == 0, "column tiles are taken in pairs");
#pragma unroll
        for (int nt = 0; nt < NT; nt += 2) {
            u32x4_t bf[2][3];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) bf[u][pc] = Bs[(pc * NO + kk) * 128 + rbase + 4 * ((nt + u) ^ (r & 3))];
            constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};      // the small terms first
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int ri = 0; ri < RI; ++ri) acc[ri][nt + u] = mfma_bf16(af[ri][TA[t]], bf[u][TB[t]], acc[ri][nt + u]);
        }
    }
    // ---- write this slab (every element of the parameter range, zeros included) ---------------------
    float* sw = a.slabW + (int64_t)slab * a.slab_stride;
#pragma unroll
    for (int ri = 0; ri < RI; ++ri)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nt >= nvalid) continue;
            const int q = q0 + nt * 16 + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + (wave * RI + ri) * 16 + kk * 4 + j;
                if (i < 4 * Pp) sw[(int64_t)i * Pp + q] = acc[ri][nt][j];
            }
        }
    if (ib == 0) {                                          // db[q]: the NO octet partials of every channel, in octet order
        float* red = reinterpret_cast<float*>(smem);        // [NO][BQ]
        __syncthreads();
        if (!stA && stOn) {
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) red[oct * BQ + 4 * c4 + ch] = bsum[ch];
        }
        __syncthreads();
        if (tid < BQ && q0 + tid < Pp) {
            float v = red[tid];
#pragma unroll
            for (int o = 1; o < NO; ++o) v += red[o * BQ + tid];
            a.slabB[(int64_t)slab * a.slab_stride + q0 + tid] = v;
        }
    }
}

// =================================================================================================
// Tap-split kernels for small channel counts (Pp = 16*NT <= 64, i.e. F <= 11: frappe, book-crossing, ml-tag).
//
// At these sizes a layer is a few hundred thousand MFMAs at most and a K loop with a barrier per step is
// latency-bound.  Here the four wavefronts of a workgroup take one filter tap each, every load of a
// wavefront is issued up front (one memory latency per workgroup instead of one per K step), and the only
// barriers are the ones around the cross-tap reduction.
//   conv_fwd_taps : wave t multiplies its 16*RM rows of tap t by W[t] (staged once into a wave-private LDS
//                   quarter), the four partial tiles are summed through LDS in tap order.
//   dgrad_taps    : wave t produces the gradient of tap t's input positions; the B fragments are rows of W
//                   read straight from L2 as 16-byte pieces (no LDS, no barrier at all for L0 = false).
// =================================================================================================
// rows [m0, m0 + 16*RM) clipped to m_hi (the fused forward kernel passes the rows of ONE example)
#ifdef CFFM_PHASE_TIMERS
extern "C" int cffm_debug_wg_times(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cffm_wg_times), sizeof(cffm_wg_times));
}
extern "C" int cffm_debug_bwd_times(unsigned long long* host32) {
    return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(cffm_bwd_times), sizeof(cffm_bwd_times));
}
extern "C" int cffm_debug_phase_times(unsigned long long* host16) {
    return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(cffm_phase_times), sizeof(cffm_phase_times));
}
#endif

// G groups of four tap-wavefronts (G = 1: 256 threads): group g takes the RM row tiles starting at m0 + g*16*RM; a group
// whose rows lie beyond m_hi only helps staging the filter and keeps the barriers.
// inL / outL (fused forward only): LDS copies of this example's input / output activations, [rows][PP], rows counted
// from m_base; the A operand then never touches global memory and the output is left where the next phase reads it.
template <int NT, int RM, bool GEN, int G = 1, int ACTC = -1>
__device__ __forceinline__ void conv_fwd_taps_body(const ConvArgs& a, int64_t m0_wg, int64_t m_hi, char* smem,
                                                   const float* inL = nullptr, float* outL = nullptr, int64_t m_base = 0) {
    const int act = ACTC >= 0 ? ACTC : a.act;           // ACTC >= 0: compile-time activation id (README shapes)
    constexpr int PP = NT * 16, LDW = PP + 4, BM = 16 * RM;
    float* Wl = reinterpret_cast<float*>(smem);                // [4][PP/4 row groups][4*PP + 16]; reused as the reduction buffer
    uint32_t* lut = reinterpret_cast<uint32_t*>(Wl + 4 * PP * LDW);      // [PP]            (GEN)
    float* Es = reinterpret_cast<float*>(lut + PP);           // [n_ex][F][Dp]   (GEN)
    const int tid = threadIdx.x, lane = tid & 63, tap = (tid >> 6) & 3, grp = tid >> 8, r = lane & 15, kk = lane >> 4;
    const int64_t m0 = m0_wg + (int64_t)grp * BM;
    const bool work = m0 < m_hi;
    const int So = 1 << a.lgSo, Sin = 2 * So, Dp = a.D + 1, dh = tap >> 1, dw = tap & 1;
    const int b0 = (int)(m0 >> (2 * a.lgSo));

    PHASE_MARK2(0);
    // ---- issue every global load of this wave ----------------------------------------------------------
    // W[tap] goes L2 -> LDS directly (global_load_lds_dwordx4), one instruction per group of four filter rows: 4 * PP floats =
    // 16 * NT lanes of 16 bytes, contiguous on both sides.  The LDS image keeps its 2-way-at-worst bank pattern with 16 floats
    // of padding BEHIND EVERY GROUP (row k at k * PP + (k >> 2) * 16: the four row groups kk of a fragment read lie 4 * PP + 16
    // floats apart, 16 banks) instead of 4 floats behind every row - same size, PP * LDW per tap.  Through registers
    // (load all, then store all) hipcc, out of registers in the fused forward, sank every load down to its LDS store: load,
    // s_waitcnt vmcnt(0), ds_write, five times in a row - five L2 round trips per layer and example (ISA of fwd_all_kernel).
    constexpr int NGRP = PP / 4, GSTR = 4 * PP + 16;          // row groups per tap, floats between groups
    static_assert(NGRP * GSTR == PP * LDW, "the grouped image fills the padded one exactly");
    float* Wt = Wl + tap * (PP * LDW);
    // the bias of this thread's output elements (used after the cross-tap reduction): requested now, so that nothing is left to
    // wait for between the stores of the epilogue
    constexpr int RT = RM * G;                                 // row tiles of the workgroup
    constexpr int EPI = (RT * NT * 64 + 256 * G - 1) / (256 * G);
    float bpre[EPI];
#pragma unroll
    for (int it = 0; it < EPI; ++it) {
        const int idx = tid + it * 256 * G, tile = idx >> 6;
        bpre[it] = a.bias[(tile % NT) * 16 + (idx & 15)];       // (tile % NT < NT also for idx beyond the last tile)
    }
    float4 av[RM][NT];
    if (!GEN && work) {
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) {
            int64_t m = m0 + rm * 16 + r;
            if (m >= m_hi) m = m_hi - 1;
            const RowPos rp = row_pos(m, a.lgSo);
            const float* src = inL ? inL + ((2 * rp.y + dh) * Sin + 2 * rp.x + dw) * PP + 4 * kk
                                   : a.in + (((int64_t)rp.b * Sin + 2 * rp.y + dh) * Sin + 2 * rp.x + dw) * PP + 4 * kk;
#pragma unroll
            for (int h = 0; h < NT; ++h) av[rm][h] = *reinterpret_cast<const float4*>(src + 16 * h);
        }
    } else if (GEN) {
        const int S2 = So * So;
        const int n_ex = BM > S2 ? BM / S2 : 1;
        build_pair_lut(lut, a.F, PP);
        stage_examples(Es, a.in, b0, n_ex, a.B, a.F, a.D, Dp);
    }
    // (the DMA is requested BEHIND the A operands / the staging above: hipcc waits with vmcnt(0) in front of any LDS access that
    //  follows a global_load_lds, and in the fused forward the A operands are LDS reads)
    {
        const float4* wsrc = reinterpret_cast<const float4*>(a.W + tap * PP * PP);
        if (lane < 16 * NT) {
#pragma unroll
            for (int gI = grp; gI < NGRP; gI += G)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(wsrc + gI * (16 * NT) + lane),
                                                 (void __attribute__((address_space(3)))*)(Wt + gI * GSTR), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's filter rows have landed (and its A operands, and the bias)
    lds_barrier();
    PHASE_MARK2(1);
    if (!work) {
    } else if (GEN) {
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) {
            int64_t m = m0 + rm * 16 + r;
            if (m >= m_hi) m = m_hi - 1;
            const RowPos rp = row_pos(m, a.lgSo);
            const int eoff = (rp.b - b0) * a.F * Dp;
            const int iy = eoff + 2 * rp.y + dh, jx = eoff + 2 * rp.x + dw;
#pragma unroll
            for (int h = 0; h < NT; ++h) {
                const uint4 l4 = *reinterpret_cast<const uint4*>(&lut[16 * h + 4 * kk]);
                av[rm][h].x = Es[iy + (l4.x & 0xffff) * Dp] * Es[jx + (l4.x >> 16) * Dp];
                av[rm][h].y = Es[iy + (l4.y & 0xffff) * Dp] * Es[jx + (l4.y >> 16) * Dp];
                av[rm][h].z = Es[iy + (l4.z & 0xffff) * Dp] * Es[jx + (l4.z >> 16) * Dp];
                av[rm][h].w = Es[iy + (l4.w & 0xffff) * Dp] * Es[jx + (l4.w >> 16) * Dp];
            }
        }
    } else {
#pragma unroll
        for (int rm = 0; rm < RM; ++rm)
#pragma unroll
            for (int h = 0; h < NT; ++h) {
                av[rm][h].x = act_pos(av[rm][h].x, act); av[rm][h].y = act_pos(av[rm][h].y, act);
                av[rm][h].z = act_pos(av[rm][h].z, act); av[rm][h].w = act_pos(av[rm][h].w, act);
            }
    }
    PHASE_MARK2(2);
    // ---- MFMA: K = PP channels of this tap -----------------------------------------------------------------
    f32x4 acc[RM][NT];
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[rm][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (work)
#pragma unroll
    for (int h = 0; h < NT; ++h) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int krow = 16 * h + 4 * kk + t;
            float bf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[nt] = Wt[krow * PP + (krow >> 2) * 16 + nt * 16 + r];
#pragma unroll
            for (int rm = 0; rm < RM; ++rm) {
                const float x = t == 0 ? av[rm][h].x : t == 1 ? av[rm][h].y : t == 2 ? av[rm][h].z : av[rm][h].w;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[rm][nt] = mfma16(x, bf[nt], acc[rm][nt]);
            }
        }
    }
    // ---- sum the four taps (fixed order) and write relu(conv + bias) ------------------------------------------
    PHASE_MARK2(3);
    lds_barrier();                                           // every wave is done with its W quarter
    PHASE_MARK2(4);
    f32x4* red = reinterpret_cast<f32x4*>(Wl);                 // [4 taps][RT*NT tiles][64 lanes]
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) red[(tap * RT * NT + (grp * RM + rm) * NT + nt) * 64 + lane] = acc[rm][nt];
    lds_barrier();
#pragma unroll
    for (int it = 0; it < EPI; ++it) {
        const int idx = tid + it * 256 * G;
        if (idx >= RT * NT * 64) break;
        const int tile = idx >> 6, ln = idx & 63, rm = tile / NT, nt = tile - rm * NT;
        f32x4 v = red[idx];
        v += red[RT * NT * 64 + idx];
        v += red[2 * RT * NT * 64 + idx];
        v += red[3 * RT * NT * 64 + idx];
        const int n = nt * 16 + (ln & 15);
        const float bv = bpre[it];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t m = m0_wg + rm * 16 + (ln >> 4) * 4 + j;
            const float c = fmaxf(v[j] + bv, 0.f);
            if (m < m_hi) {
                a.out[m * PP + n] = c;
                if (outL) outL[(m - m_base) * PP + n] = c;
            }
        }
    }
    PHASE_MARK2(5);
}

template <int NT, int RM, bool GEN>
__global__ __launch_bounds__(256) void conv_fwd_taps_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv_fwd_taps_body<NT, RM, GEN>(a, (int64_t)blockIdx.x * (16 * RM), a.Mtot, smem);
}

// conv_fwd_rows: many-row variant of the small-Pp forward (layer 0 at frappe: 65536 rows).  The whole padded
// filter [4*PP][PP] is staged once into LDS (36 KB at PP = 48, one barrier); each wavefront then runs all four
// taps over its own 16*RM rows, so there is no cross-wave reduction and the epilogue comes straight from the
// accumulators.
template <int NT, int RM, bool GEN>
__global__ __launch_bounds__(256) void conv_fwd_rows_kernel(ConvArgs a) {
    constexpr int PP = NT * 16, BM = 64 * RM;
    constexpr int NW4 = 4 * PP * PP / 4 / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Wl = reinterpret_cast<float*>(smem);                // [4*PP][PP], columns rotated by 16 on rows with bit 2 set
    uint32_t* lut = reinterpret_cast<uint32_t*>(Wl + 4 * PP * PP);       // [PP]            (GEN)
    float* Es = reinterpret_cast<float*>(lut + PP);           // [n_ex][F][Dp]   (GEN)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kk = lane >> 4;
    const int So = 1 << a.lgSo, Sin = 2 * So, Dp = a.D + 1;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int b0 = (int)(m0 >> (2 * a.lgSo));

    float4 wv[NW4];
    const float4* wsrc = reinterpret_cast<const float4*>(a.W);
#pragma unroll
    for (int i = 0; i < NW4; ++i) wv[i] = wsrc[tid + 256 * i];
    float4 av[RM][4][NT];
    int iy[RM], jx[RM];
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) {
        int64_t m = m0 + wave * (16 * RM) + rm * 16 + r;
        if (m >= a.Mtot) m = a.Mtot - 1;
        const RowPos rp = row_pos(m, a.lgSo);
        if (GEN) {
            const int eoff = (rp.b - b0) * a.F * Dp;
            iy[rm] = eoff + 2 * rp.y;
            jx[rm] = eoff + 2 * rp.x;
        } else {
            const float* src = a.in + (((int64_t)rp.b * Sin + 2 * rp.y) * Sin + 2 * rp.x) * PP + 4 * kk;
#pragma unroll
            for (int tap = 0; tap < 4; ++tap)
#pragma unroll
                for (int h = 0; h < NT; ++h)
                    av[rm][tap][h] = *reinterpret_cast<const float4*>(src + ((tap >> 1) * Sin + (tap & 1)) * PP + 16 * h);
        }
    }
    if (GEN) {
        const int S2 = So * So;
        const int n_ex = BM > S2 ? BM / S2 : 1;
        build_pair_lut(lut, a.F, PP);
        stage_examples(Es, a.in, b0, n_ex, a.B, a.F, a.D, Dp);
    }
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
        const int u = tid + 256 * i, row = u / (PP / 4);
        int cc = 4 * (u % (PP / 4)) + ((row >> 2) & 1) * 16;      // unpadded rows stay conflict-free for the B reads:
        if (cc >= PP) cc -= PP;                                   // the two k rows of a half-wave use disjoint bank halves
        *reinterpret_cast<float4*>(&Wl[row * PP + cc]) = wv[i];
    }
    __syncthreads();

    f32x4 acc[RM][NT];
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[rm][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int bcol[NT];                    // rows 4*kk+t have bit 2 = kk & 1 -> their columns are rotated by 16
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int c = nt * 16 + r + (kk & 1) * 16;
        bcol[nt] = c >= PP ? c - PP : c;
    }
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
        const int dh = tap >> 1, dw = tap & 1;
#pragma unroll
        for (int h = 0; h < NT; ++h) {
            float4 x4[RM];
            if (GEN) {
                const uint4 l4 = *reinterpret_cast<const uint4*>(&lut[16 * h + 4 * kk]);
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) {
                    const int i0 = iy[rm] + dh, j0 = jx[rm] + dw;
                    x4[rm].x = Es[i0 + (l4.x & 0xffff) * Dp] * Es[j0 + (l4.x >> 16) * Dp];
                    x4[rm].y = Es[i0 + (l4.y & 0xffff) * Dp] * Es[j0 + (l4.y >> 16) * Dp];
                    x4[rm].z = Es[i0 + (l4.z & 0xffff) * Dp] * Es[j0 + (l4.z >> 16) * Dp];
                    x4[rm].w = Es[i0 + (l4.w & 0xffff) * Dp] * Es[j0 + (l4.w >> 16) * Dp];
                }
            } else {
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) {
                    x4[rm].x = act_pos(av[rm][tap][h].x, a.act); x4[rm].y = act_pos(av[rm][tap][h].y, a.act);
                    x4[rm].z = act_pos(av[rm][tap][h].z, a.act); x4[rm].w = act_pos(av[rm][tap][h].w, a.act);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int krow = tap * PP + 16 * h + 4 * kk + t;
                float bf[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bf[nt] = Wl[krow * PP + bcol[nt]];
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) {
                    const float x = t == 0 ? x4[rm].x : t == 1 ? x4[rm].y : t == 2 ? x4[rm].z : x4[rm].w;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[rm][nt] = mfma16(x, bf[nt], acc[rm][nt]);
                }
            }
        }
    }
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 16 + r;
            const float bv = a.bias[n];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t m = m0 + wave * (16 * RM) + rm * 16 + kk * 4 + j;
                if (m < a.Mtot) a.out[m * PP + n] = fmaxf(acc[rm][nt][j] + bv, 0.f);
            }
        }
}

// conv0_fact_fwd: layer 0 in FACTORISED form (small Pp).  The input channels of layer 0 are rank-1,
// A[(y,x),(dh,dw,(i,j))] = E[i][2y+dh] * E[j][2x+dw], so the contraction splits exactly:
//     T[dh][i][x][q] = sum_{dw} sum_{j>i} E[j][2x+dw] * W[dh][dw][(i,j)][q]          step 1: 4*S*P^2 MACs
//     C[y][x][q]     = relu(b[q] + sum_{dh} sum_i E[i][2y+dh] * T[dh][i][x][q])       step 2: 2*F*S^2*P MACs
// against S^2 * 4P * P for the direct form: 5.8x fewer MFMAs at frappe (F10 D32), executed on the same
// v_mfma_f32_16x16x4_f32.  One workgroup per example: the embedding tile, the whole filter (36 KB) and T
// (2F x S x Pp floats, 61 KB at frappe) live in LDS; nothing but C goes back to HBM.
//   step 1, unit (dh, i, nt): rows x, k = (dw, j > i) - the pairs of a fixed i are CONTIGUOUS filter rows;
//                             A fragment = one ds_read of E, B fragment = one ds_read of the staged filter.
//   step 2, wave w owns x = w, w+4, ...: rows y, k = (dh, i), B fragment = T[(dh,i)][x][q] (row pitch padded by
//                             16 floats so the two k rows of a half-wave hit disjoint banks).
template <int NT, int NW = 4>
// staged (fused forward): the caller has put the filter and the embedding tile into LDS and passed the barrier
__device__ __forceinline__ void conv0_fact_fwd_body(const ConvArgs& a, int b, char* smem, float* outL = nullptr, bool staged = false) {
    constexpr int PP = NT * 16, NTH = 64 * NW, XQ = 16 / NW;      // NW wavefronts; step 2 gives each XQ columns at a time
    const int F = a.F, D = a.D, S = D / 2, Dp = D + 1, RT = S / 16;
    const int TP = S * PP + 16;                                 // pitch of one (dh, i) plane of T
    float* Wl = reinterpret_cast<float*>(smem);                // [4*PP][PP]
    float* T = Wl + 4 * PP * PP;                                // [2F][TP]
    float* Es = T + 2 * F * TP;                                 // [F][Dp]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kk = lane >> 4;
    if (!staged) {   // stage the filter and the embedding tile (one barrier)
        const float4* wsrc = reinterpret_cast<const float4*>(a.W);
        for (int i = tid; i < 4 * PP * PP / 4; i += NTH) reinterpret_cast<float4*>(Wl)[i] = wsrc[i];
        const float* e = a.in + (int64_t)b * F * D;
        const float invD = 1.f / (float)D;
        for (int i = tid; i < F * D; i += NTH) {
            const int f = fast_div(i, invD), d = i - f * D;
            Es[f * Dp + d] = e[i];
        }
        lds_barrier();
    }
    // ---- step 1: unit (dh, i, rt) with all NT column tiles at once (one A fragment feeds NT MFMAs) ---------------
    const int units = 2 * (F - 1) * RT;
    for (int u = wave; u < units; u += NW) {
        int t = u;
        const int rt = t % RT; t /= RT;
        const int i = t % (F - 1), dh = t / (F - 1);
        const int nj = F - 1 - i, K = 2 * nj;                   // k = dw * nj + (j - i - 1)
        const int base = i * (2 * F - i - 1) / 2;               // first pair (i, i+1)
        const int x = rt * 16 + r;
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < K; k0 += 4) {
            const int k = k0 + kk;
            const bool ok = k < K;
            const int dw = (ok && k >= nj) ? 1 : 0, jj = ok ? k - dw * nj : 0;
            const float av = ok ? Es[(i + 1 + jj) * Dp + 2 * x + dw] : 0.f;
            const float* wr = Wl + ((dh * 2 + dw) * PP + base + jj) * PP + r;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(av, ok ? wr[nt * 16] : 0.f, acc[nt]);
        }
        float* tp = T + (dh * F + i) * TP + r;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) tp[(rt * 16 + kk * 4 + j) * PP + nt * 16] = acc[nt][j];
    }
    // planes (dh, F-1) have no pairs: zero them so that step 2 can run a dense k
    for (int e = tid; e < 2 * S * PP; e += NTH) {
        const int dh = e / (S * PP), o = e - dh * (S * PP);
        T[(dh * F + F - 1) * TP + o] = 0.f;
    }
    lds_barrier();
    // ---- step 2: a wave takes four x at a time (one A fragment feeds 4*NT MFMAs) ----------------------------------
    const int K2 = 2 * F, ks2 = (K2 + 3) / 4;
    float bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = a.bias[nt * 16 + r];
    for (int rt = 0; rt < RT; ++rt) {
        const int y = rt * 16 + r;
        for (int xg = wave * XQ; xg < S; xg += 16) {
            f32x4 acc[XQ][NT];
#pragma unroll
            for (int q4 = 0; q4 < XQ; ++q4)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[q4][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int s2 = 0; s2 < ks2; ++s2) {
                const int k = 4 * s2 + kk;
                const bool ok = k < K2;
                const int dh = (ok && k >= F) ? 1 : 0, i = ok ? k - dh * F : 0;
                const float av = ok ? Es[i * Dp + 2 * y + dh] : 0.f;
                const float* tb = T + (ok ? k : 0) * TP + xg * PP + r;
                float bv[XQ][NT];
#pragma unroll
                for (int q4 = 0; q4 < XQ; ++q4)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv[q4][nt] = ok ? tb[q4 * PP + nt * 16] : 0.f;
#pragma unroll
                for (int q4 = 0; q4 < XQ; ++q4)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[q4][nt] = mfma16(av, bv[q4][nt], acc[q4][nt]);
            }
#pragma unroll
            for (int q4 = 0; q4 < XQ; ++q4)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int yy = rt * 16 + kk * 4 + j;
                        const float c = fmaxf(acc[q4][nt][j] + bias[nt], 0.f);
                        a.out[(((int64_t)b * S + yy) * S + xg + q4) * PP + nt * 16 + r] = c;
                        if (outL) outL[(yy * S + xg + q4) * PP + nt * 16 + r] = c;
                    }
        }
    }
}

template <int NT>
__global__ __launch_bounds__(256) void conv0_fact_fwd_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv0_fact_fwd_body<NT>(a, blockIdx.x, smem);
}

// =================================================================================================
// fwd_all: the whole forward of the fused train step in ONE launch (small Pp, D = 32/64 with factorisable layer 0).
// Workgroup b < B runs example b end to end - gather + inner branch, factorised layer 0, the remaining conv
// layers (tap-split, rows of this example only), pooling + heads + loss term - with every intermediate that
// the backward pass needs written to the workspace exactly as the separate kernels write it.  A workgroup only
// ever re-reads global data it wrote itself (its own C_l rows), which __syncthreads() orders.  Workgroup B sorts
// the packed (id, slot) keys of the sparse update straight from the ids, so the sort costs no launch either.
// At the frappe shape this replaces 7 launches (~4 us of dispatch + drain each) by one.
// =================================================================================================
struct FwdAllArgs {
    InnerFwdArgs inner;
    ConvArgs conv[CFFM_MAX_LAYERS];
    HeadArgs head;
    const int32_t* ids;
    unsigned long long* keys_sorted;
    int live, n_rows, id_bits, B;
    int rank_keys;                // 0: the keys are placed later, by the inner-branch role of bwd_top_kernel
    int c0_off, c1_off;           // LDS byte offsets of the copy of C_0 and of the packed copies of C_1.. (fused_c_off); c0_off < 0:
                                  // the activations go through global memory
    int early_off, es_off;        // early_off > 0: LDS byte offset of the inner branch's scratch (= the T planes of layer 0), es_off: of
                                  // the embedding tile of layer 0 - the layer-0 operands are then staged during the gather
};


// Stable sort of the B*F sparse-update keys (id << 32 | slot) without a sort: the keys are unique, so the place of a key
// is the number of keys below it.  The workgroup of example b places its own F keys - n*F/256 compares per thread,
// every workgroup in parallel, no extra launch and no serial tail.
#define RANK_MAXF 12
template <int NW>
__device__ __forceinline__ void rank_keys_body(const int32_t* __restrict__ ids, int n, int b, int F,
                                               unsigned long long* __restrict__ out, char* smem) {
    float* cnt = reinterpret_cast<float*>(smem);              // [NW waves][RANK_MAXF]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long mine[RANK_MAXF];
#pragma unroll
    for (int f = 0; f < RANK_MAXF; ++f) {
        const int slot = b * F + (f < F ? f : 0);
        mine[f] = f < F ? (((unsigned long long)(unsigned)ids[slot] << 32) | (unsigned)slot) : 0ull;
    }
    int c[RANK_MAXF];
#pragma unroll
    for (int f = 0; f < RANK_MAXF; ++f) c[f] = 0;
    // n <= 4096 candidates: this thread's ids are fetched in groups of 8 independent loads (one L2 latency per group
    // instead of one per candidate: 10 in a row at the frappe shape)
    constexpr int GRP = 8;
    for (int j0 = tid; j0 < n; j0 += 64 * NW * GRP) {
        unsigned idj[GRP];
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            const int j = j0 + 64 * NW * u;
            idj[u] = (unsigned)ids[j < n ? j : n - 1];
        }
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            const int j = j0 + 64 * NW * u;
            const unsigned long long kj = j < n ? (((unsigned long long)idj[u] << 32) | (unsigned)j) : ~0ull;   // ~0: below no key
#pragma unroll
            for (int f = 0; f < RANK_MAXF; ++f) c[f] += kj < mine[f] ? 1 : 0;
        }
    }
#pragma unroll
    for (int f = 0; f < RANK_MAXF; ++f) {
        const float t = wave_sum((float)c[f]);               // counts <= 4096: exact in fp32
        if (lane == 0) cnt[wave * RANK_MAXF + f] = t;
    }
    __syncthreads();
    if (tid < F) {
        float c = cnt[tid];
#pragma unroll
        for (int w = 1; w < NW; ++w) c += cnt[w * RANK_MAXF + tid];
        const int rank = (int)c;
        const int slot = b * F + tid;
        out[rank] = ((unsigned long long)(unsigned)ids[slot] << 32) | (unsigned)slot;
    }
    __syncthreads();
}

// NW wavefronts per workgroup (8 at the README shapes): one example still owns one workgroup, but every SIMD now has
// two wavefronts to switch between, which is what hides the LDS / MFMA / L2 latencies of the per-example phases.
// ACT >= 0: the activation id compiled in (selu / elu / relu of the three README commands): the act switches of the inner
// branch, of the A operands of the conv layers and of the pooling sweep fold into straight code
// The layer-0 filter on its way to LDS: fetch() issues this thread's float4 loads (before the gather's), operator() - called by
// inner_fwd_body while the gathered rows are in flight - stores them at their LDS place.  Held by value, static indices only.
template <int N, int NTH>
struct ParkFilter {
    float4 w[N];
    float4* dst;
    int n4;
    __device__ __forceinline__ void fetch(const float4* src, float4* dst_, int n4_) {
        dst = dst_; n4 = src != nullptr ? n4_ : 0;
#pragma unroll
        for (int ii = 0; ii < N; ++ii) {
            const int i = threadIdx.x + NTH * ii;
            w[ii] = i < n4 ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __device__ __forceinline__ void operator()() const {
#pragma unroll
        for (int ii = 0; ii < N; ++ii) {
            const int i = threadIdx.x + NTH * ii;
            if (i < n4) dst[i] = w[ii];
        }
    }
};

template <int NT, int NW, int ACT = -1>
__global__ __launch_bounds__(64 * NW) void fwd_all_kernel(FwdAllArgs fa) {
    constexpr int G = NW / 4, PP = NT * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS copies of the conv outputs of this example (fa.c_off[l] >= 0): layer l+1 and the head read them there, the
    // global copies (needed by the backward) are written behind the LDS-only barriers and drain in the background
    auto CL = [&](int l) -> float* {
        return fa.c0_off >= 0 ? reinterpret_cast<float*>(smem + fused_c_off(l, fa.c0_off, fa.c1_off, fa.inner.g.D, PP)) : nullptr;
    };
    const bool lds_act = fa.c0_off >= 0;
#ifdef CFFM_PHASE_TIMERS
    if (threadIdx.x == 0) cffm_wg_times[2 * blockIdx.x] = wall_clock64();
#endif
    for (int b = blockIdx.x; b < fa.B; b += gridDim.x) {
        if (b != (int)blockIdx.x) __syncthreads();
        // early (fa.early_off > 0): the layer-0 filter is fetched first of all, the inner branch works in the LDS that the T
        // planes of layer 0 take later, so that W_0 reaches its LDS place while the gather is in flight, and the gather leaves
        // the outer rows in the embedding tile: layer 0 then starts on LDS-resident operands
        ParkFilter<(4 * PP * PP / 4 + 64 * NW - 1) / (64 * NW), 64 * NW> park;
        const bool early = fa.early_off > 0;
        park.fetch(early ? reinterpret_cast<const float4*>(fa.conv[0].W) : nullptr, reinterpret_cast<float4*>(smem), 4 * PP * PP / 4);
        char* smem_i = smem + fa.early_off;
        if (fa.rank_keys) rank_keys_body<NW>(fa.ids, fa.n_rows, b, fa.inner.g.F, fa.keys_sorted, smem_i);
        PHASE_MARK(0);
        // gathers Ei/Eo/fb of example b (full barrier inside), inner_out[b]
        inner_fwd_body<ACT>(fa.inner, b, smem_i, early ? reinterpret_cast<float*>(smem + fa.es_off) : nullptr, fa.inner.g.D + 1, park);
        if (lds_act) lds_barrier(); else __syncthreads();
        PHASE_MARK(1);
        conv0_fact_fwd_body<NT, NW>(fa.conv[0], b, smem, CL(0), early);   // reads Eo[b], writes C_0[b]
        for (int l = 1; l < fa.live; ++l) {
            if (lds_act) lds_barrier(); else __syncthreads();
            PHASE_MARK(1 + l);
            const ConvArgs& ca = fa.conv[l];
            const int64_t rows = 1ll << (2 * ca.lgSo), m_lo = (int64_t)b * rows, m_hi = m_lo + rows;
            if (rows >= 64) {
                for (int64_t m0 = m_lo; m0 < m_hi; m0 += 64) {
                    if (m0 > m_lo) lds_barrier();
                    conv_fwd_taps_body<NT, 4 / G, false, G, ACT>(ca, m0, m_hi, smem, CL(l - 1), CL(l), m_lo);
                }
            } else if (rows >= 32) {
                conv_fwd_taps_body<NT, (G >= 2 ? 1 : 2), false, G, ACT>(ca, m_lo, m_hi, smem, CL(l - 1), CL(l), m_lo);
            } else {
                conv_fwd_taps_body<NT, 1, false, G, ACT>(ca, m_lo, m_hi, smem, CL(l - 1), CL(l), m_lo);
            }
        }
        if (lds_act) lds_barrier(); else __syncthreads();
        PHASE_MARK(1 + fa.live);
        head_fwd_body<NW, ACT>(fa.head, b, smem, lds_act, fa.c0_off, fa.c1_off);
        PHASE_MARK(2 + fa.live);
    }
#ifdef CFFM_PHASE_TIMERS
    if (threadIdx.x == 0) cffm_wg_times[2 * blockIdx.x + 1] = wall_clock64();
#endif
}

// conv0_fact_bwd: the whole backward of layer 0 in factorised form (S = 16, small Pp), one workgroup per
// gradient slab looping over its examples.  With T as in conv0_fact_fwd and dC the gradient wrt the pre-relu
// conv output:
//   A  T[dh][i][x][q]   recomputed (150 MFMAs at frappe)                                   -> LDS
//   B  dEi[(dh,i)][y]   = sum_{x,q} dC[y][x][q] * T[dh][i][x][q]        rows y, k = (x,q), cols (dh,i)
//   C  dT[dh][i][x][q]  = sum_y E[i][2y+dh] * dC[y][x][q]               rows (dh,i), k = y, cols (x,q) -> LDS (over T)
//      db[q]            = sum_{y,x} dC[y][x][q]                         (from the B fragments of C)
//   D  dW[dh][dw][(i,j)][q] += sum_x E[j][2x+dw] * dT[dh][i][x][q]      rows (dw,j>i), k = x, cols q -> slab
//   E  dEj[(dw,j)][x]   = sum_{dh,i<j,q} dT[dh][i][x][q] * W[dh][dw][(i,j)][q]   rows x, k = q, cols (dw,j)
//   F  dEo[f][h] = dEi[(h&1,f)][h>>1] + dEj[(h&1,f)][h>>1] + ds0[h]*R_f + Q_f  (s0 pool gradient in closed form)
// ~1640 MFMAs per example against 4608 for the direct wgrad + dgrad, no atomics, fixed summation order.
// wg / nwg: this workgroup's slab and the number of slabs (= workgroups of the launch)
template <int NT, int F_, int D_, int NW>
__device__ __forceinline__ void conv0_fact_bwd_body(const DgradArgs& a, float* __restrict__ slabW, float* __restrict__ slabB,
                                                    int64_t slab_stride, char* smem, int wg, int nwg) {
    constexpr int PP = NT * 16, S = 16, NTH = 64 * NW, XQ = 16 / NW;     // NW wavefronts (4, 8 or 16), XQ x columns per wave
    // F_/D_ != 0: the README shapes compiled in, so that the pair indexing, the divisions and the loop bounds fold
    const int F = F_ ? F_ : a.F, D = D_ ? D_ : a.D, Dp = D + 1, P = F_ ? F_ * (F_ - 1) / 2 : a.P, F2 = 2 * F;
    const int TP = S * PP + 16;
    float* Wl = reinterpret_cast<float*>(smem);                // [4*PP][PP]
    float* T = Wl + 4 * PP * PP;                                // [2F][TP]   T, later dT
    float* part = T + 2 * F * TP;                               // [NW waves][16][32] partial tiles (phases B, E)
    float* dEi = part + NW * 16 * 32;                            // [32][16]  (n = (dh,i), y)
    float* dEj = dEi + 32 * 16;                                 // [32][16]  (n = (dw,j), x)
    float* bred = dEj + 32 * 16;                                // [NW][PP] bias partials
    float* rs = bred + NW * PP;                                  // [F] row sums, [F] dots
    float* Es = rs + 2 * F;                                     // [F][Dp]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kk = lane >> 4;
    float* sw = slabW + (int64_t)wg * slab_stride;
    float* sb = slabB + (int64_t)wg * slab_stride;
    PHASE_MARKB(12, wg);
    {
        const float4* wsrc = reinterpret_cast<const float4*>(a.W);
        for (int i = tid; i < 4 * PP * PP / 4; i += NTH) reinterpret_cast<float4*>(Wl)[i] = wsrc[i];
        // rows of padded pairs (p >= P) are never produced below: they must read as zeros in the reduction
        for (int e = tid; e < 4 * (PP - P) * PP; e += NTH) {
            const int tap = e / ((PP - P) * PP), o = e - tap * ((PP - P) * PP);
            sw[(tap * PP + P) * PP + o] = 0.f;
        }
    }
    bool first = true;
    float bacc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bacc[nt] = 0.f;

    for (int b = wg; b < a.B; b += nwg) {
        __syncthreads();
        {
            const float* e = a.Cprev + (int64_t)b * F * D;      // Eo rows of this example
            const float invD = 1.f / (float)D;
            for (int i = tid; i < F * D; i += NTH) {
                const int f = fast_div(i, invD), d = i - f * D;
                Es[f * Dp + d] = e[i];
            }
        }
        __syncthreads();
        const float* dCb = a.dC + (int64_t)b * S * S * PP;
        PHASE_MARKB(13, wg);
        // ---- A: T ----------------------------------------------------------------------------------------------
        for (int u = wave; u < 2 * (F - 1); u += NW) {
            const int i = u % (F - 1), dh = u / (F - 1);
            const int nj = F - 1 - i, K = 2 * nj, base = i * (2 * F - i - 1) / 2;
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < K; k0 += 4) {
                const int k = k0 + kk;
                const bool ok = k < K;
                const int dw = (ok && k >= nj) ? 1 : 0, jj = ok ? k - dw * nj : 0;
                const float av = ok ? Es[(i + 1 + jj) * Dp + 2 * r + dw] : 0.f;
                const float* wr = Wl + ((dh * 2 + dw) * PP + base + jj) * PP + r;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(av, ok ? wr[nt * 16] : 0.f, acc[nt]);
            }
            float* tp = T + (dh * F + i) * TP + r;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) tp[(kk * 4 + j) * PP + nt * 16] = acc[nt][j];
        }
        for (int e = tid; e < 2 * S * PP; e += NTH) {
            const int dh = e / (S * PP), o = e - dh * (S * PP);
            T[(dh * F + F - 1) * TP + o] = 0.f;
        }
        __syncthreads();
        PHASE_MARKB(14, wg);
        // ---- B: dEi = dC (rows y) x T^T, this wave's k = its four x -----------------------------------------------
        {
            f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
            float4 av[XQ][NT];
#pragma unroll
            for (int q4 = 0; q4 < XQ; ++q4)
#pragma unroll
                for (int h = 0; h < NT; ++h)
                    av[q4][h] = *reinterpret_cast<const float4*>(dCb + ((int64_t)r * S + wave + NW * q4) * PP + 16 * h + 4 * kk);
#pragma unroll
            for (int q4 = 0; q4 < XQ; ++q4) {
                const int x = wave + NW * q4;
#pragma unroll
                for (int h = 0; h < NT; ++h) {
                    float4 bv[2];
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const int n = ct * 16 + r;
                        bv[ct] = n < F2 ? *reinterpret_cast<const float4*>(T + n * TP + x * PP + 16 * h + 4 * kk)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        acc[ct] = mfma16(av[q4][h].x, bv[ct].x, acc[ct]);
                        acc[ct] = mfma16(av[q4][h].y, bv[ct].y, acc[ct]);
                        acc[ct] = mfma16(av[q4][h].z, bv[ct].z, acc[ct]);
                        acc[ct] = mfma16(av[q4][h].w, bv[ct].w, acc[ct]);
                    }
                }
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int j = 0; j < 4; ++j) part[(wave * 16 + kk * 4 + j) * 32 + ct * 16 + r] = acc[ct][j];
        }
        __syncthreads();
        for (int e = tid; e < 16 * 32; e += NTH) {               // e = y * 32 + n
            float v = part[e];
#pragma unroll
            for (int w = 1; w < NW; ++w) v += part[512 * w + e];
            dEi[(e & 31) * 16 + (e >> 5)] = v;
        }
        __syncthreads();                                          // T fully consumed: phase C may overwrite it
        PHASE_MARKB(15, wg);
        // ---- C: dT = E^T (rows (dh,i)) x dC, this wave's columns = its four x; db from the B fragments ------------
        {
            float av[2][4];                                      // A[m = rt*16 + r][k = y = 4s + kk]
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int m = rt * 16 + r, dh = m >= F ? 1 : 0, i = m - dh * F;
                    av[rt][s4] = m < F2 ? Es[i * Dp + 2 * (4 * s4 + kk) + dh] : 0.f;
                }
#pragma unroll
            for (int q4 = 0; q4 < XQ; ++q4) {
                const int x = wave + NW * q4;
                float bv[4][NT];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv[s4][nt] = dCb[((int64_t)(4 * s4 + kk) * S + x) * PP + nt * 16 + r];
                f32x4 acc[2][NT];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        bacc[nt] += bv[s4][nt];
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) acc[rt][nt] = mfma16(av[rt][s4], bv[s4][nt], acc[rt][nt]);
                    }
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int m = rt * 16 + kk * 4 + j;
                            if (m < F2) T[m * TP + x * PP + nt * 16 + r] = acc[rt][nt][j];
                        }
            }
        }
        __syncthreads();
        PHASE_MARKB(16, wg);
        // ---- D: dW slab rows (dh, dw, (i, j>i)) ---------------------------------------------------------------------
        {
            int ucount = 0;
            for (int dh = 0; dh < 2; ++dh)
                for (int i = 0; i < F - 1; ++i) {
                    const int nj = F - 1 - i, K = 2 * nj, base = i * (2 * F - i - 1) / 2;
                    for (int rt = 0; rt * 16 < K; ++rt, ++ucount) {
                        if ((ucount & (NW - 1)) != wave) continue;
                        const int m = rt * 16 + r;
                        const bool okm = m < K;
                        const int dw = (okm && m >= nj) ? 1 : 0, jj = okm ? m - dw * nj : 0;
                        f32x4 acc[NT];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) {
                            const int x = 4 * s4 + kk;
                            const float avv = okm ? Es[(i + 1 + jj) * Dp + 2 * x + dw] : 0.f;
                            const float* tb = T + (dh * F + i) * TP + x * PP + r;
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(avv, tb[nt * 16], acc[nt]);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int mm = rt * 16 + kk * 4 + j;
                            if (mm < K) {
                                const int dw2 = mm >= nj ? 1 : 0, jj2 = mm - dw2 * nj;
                                float* dst = sw + ((dh * 2 + dw2) * PP + base + jj2) * PP + r;
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) dst[nt * 16] = first ? acc[nt][j] : dst[nt * 16] + acc[nt][j];
                            }
                        }
                    }
                }
        }
        PHASE_MARKB(17, wg);
        // ---- E: dEj = dT (rows x) x W^T over this wave's (dh, i) units -------------------------------------------------
        {
            f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
            // no barrier separates D from E: E's units are dealt starting at the first wavefront that had one D unit less
            // (20 D units and 18 E units on 16 wavefronts at frappe: 3 units on the slowest wavefront instead of 4)
            int ud = 0;
            for (int i = 0; i < F - 1; ++i) ud += 2 * ((2 * (F - 1 - i) + 15) / 16);
            for (int u = (wave + NW - ud % NW) % NW; u < 2 * (F - 1); u += NW) {
                const int i = u % (F - 1), dh = u / (F - 1), base = i * (2 * F - i - 1) / 2;
                const float* ta = T + (dh * F + i) * TP + r * PP + 4 * kk;
                const float* wrow[2];
                bool okc[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int n = ct * 16 + r, dw = n >= F ? 1 : 0, j = n - dw * F;
                    okc[ct] = n < F2 && j > i;
                    wrow[ct] = Wl + ((dh * 2 + dw) * PP + base + (okc[ct] ? j - i - 1 : 0)) * PP + 4 * kk;
                }
#pragma unroll
                for (int h = 0; h < NT; ++h) {
                    const float4 av = *reinterpret_cast<const float4*>(ta + 16 * h);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const float4 bv = okc[ct] ? *reinterpret_cast<const float4*>(wrow[ct] + 16 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
                        acc[ct] = mfma16(av.x, bv.x, acc[ct]);
                        acc[ct] = mfma16(av.y, bv.y, acc[ct]);
                        acc[ct] = mfma16(av.z, bv.z, acc[ct]);
                        acc[ct] = mfma16(av.w, bv.w, acc[ct]);
                    }
                }
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int j = 0; j < 4; ++j) part[(wave * 16 + kk * 4 + j) * 32 + ct * 16 + r] = acc[ct][j];
        }
        PHASE_MARKB(18, wg);
        if (tid < 2 * F) {                                       // row sums and <ds0, E[f]> for the closed-form s0 terms
            const int f = tid % F;
            float sacc = 0.f;
            if (tid < F) { for (int h = 0; h < D; ++h) sacc += Es[f * Dp + h]; }
            else { for (int h = 0; h < D; ++h) sacc += Es[f * Dp + h] * a.dt1[(int64_t)b * a.t1w + h]; }
            rs[tid] = sacc;
        }
        __syncthreads();
        for (int e = tid; e < 16 * 32; e += NTH) {               // e = x * 32 + n
            float v = part[e];
#pragma unroll
            for (int w = 1; w < NW; ++w) v += part[512 * w + e];
            dEj[(e & 31) * 16 + (e >> 5)] = v;
        }
        __syncthreads();
        PHASE_MARKB(19, wg);
        // ---- F -----------------------------------------------------------------------------------------------------
        for (int e = tid; e < F * D; e += NTH) {
            const int f = e / D, h = e - f * D, lo = h & 1, hh = h >> 1;
            float R = 0.f, Q = 0.f;
            for (int j = f + 1; j < F; ++j) R += rs[j];
            for (int i = 0; i < f; ++i) Q += rs[F + i];
            a.dprev[(int64_t)b * F * D + e] = (dEi[(lo * F + f) * 16 + hh] + dEj[(lo * F + f) * 16 + hh])
                                              + a.dt1[(int64_t)b * a.t1w + h] * R + Q;
        }
        first = false;
    }
    PHASE_MARKB(20, wg);
    // ---- bias gradient and empty-slab zeros -----------------------------------------------------------------------------
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float v = bacc[nt];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (kk == 0) bred[wave * PP + nt * 16 + r] = v;
    }
    __syncthreads();
    if (tid < PP) {
        float v = bred[tid];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += bred[w * PP + tid];
        sb[tid] = v;
    }
    if (first) {                                                  // no example for this slab
        for (int e = tid; e < 4 * PP * PP; e += NTH) sw[e] = 0.f;
    }
    PHASE_MARKB(25, wg);
}

template <int NT, int F_, int D_, int NW>
__global__ __launch_bounds__(64 * NW) void conv0_fact_bwd_kernel(DgradArgs a, float* __restrict__ slabW,
                                                              float* __restrict__ slabB, int64_t slab_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv0_fact_bwd_body<NT, F_, D_, NW>(a, slabW, slabB, slab_stride, smem, blockIdx.x, gridDim.x);
}

// HALVES = 2 (layer 0 only): 8 wavefronts, the upper four take the second half of the workgroup's m tiles, so
// that every SIMD has two wavefronts whose MFMA and epilogue phases can overlap.
//
// Layer 0 "fast" path (So >= 16, one example per workgroup): no LDS atomics at all.  A 16-row tile is 16
// consecutive x of one (b, y).  i-side: the tile's contribution to dEo[i_p][2y+dh] is reduced over x in
// registers + two cross-lane adds and STORED at Ti[tap][p][y] (each (tap, p, y) is produced exactly once).
// j-side: dEo[j_p][2x+dw] sums over y, which the lane keeps in registers across its m tiles and stores at
// Tj[half][tap][p][x].  A final pass adds, for every (field, h), the Ti / Tj entries of the pairs that contain
// the field, in pair order: bitwise reproducible, and ~100x cheaper than ds_add_f32 (measured: the atomic
// version spent 83 K LDS cycles per CU).
// m_lo / m_end (fused top-of-backward kernel): the tiles start at m_lo and rows >= m_end are masked instead of a.Mtot
template <int NT, int RM, bool L0, int HALVES, int ACTC = -1>
// tiles (layers >= 1): 16*RM-row tiles per workgroup; with HALVES groups of four tap-wavefronts each group takes its share
// tid_in >= 0 (conv01_bwd_kernel): the body runs on a sub-range of a larger workgroup, tid_in = thread index inside it
__device__ __forceinline__ void dgrad_taps_body(const DgradArgs& a, int wg, char* smem, int64_t m_lo = 0, int64_t m_end = -1,
                                                int tiles = 1, int tid_in = -1) {
    const int act = ACTC >= 0 ? ACTC : a.act;           // ACTC >= 0: compile-time activation id (README shapes)
    constexpr int PP = NT * 16, BM = 16 * RM, NTH = 256 * HALVES, NCOPY = 4 * HALVES;
    uint32_t* lut = reinterpret_cast<uint32_t*>(smem);        // L0 only: [PP]
    float* Es = reinterpret_cast<float*>(lut + PP);           // [n_ex][F][Dp]
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63, wid = tid >> 6, tap = wid & 3, half = wid >> 2, r = lane & 15, kk = lane >> 4;
    const int So = 1 << a.lgSo, Sin = 2 * So, P = a.P, Dp = a.D + 1, S2 = So * So, dh = tap >> 1, dw = tap & 1;
    const int rows_per_wg = L0 ? (S2 > BM ? S2 : BM) : BM * tiles;
    const int n_ex = L0 ? rows_per_wg / S2 : 0;
    const int mtiles = rows_per_wg / BM;
    const int64_t Mend = m_end < 0 ? a.Mtot : m_end;
    const int64_t wg_m0 = m_lo + (int64_t)wg * rows_per_wg;
    const int b0 = (int)(wg_m0 >> (2 * a.lgSo));
    const int exsz = a.F * Dp;
    const bool fast = L0 && RM == 4 && a.lgSo >= 4 && a.lgSo <= 6;      // tiles per y = So/16 divides RM
    float* rs = Es + n_ex * exsz;                               // [n_ex][F] row sums, then [n_ex][F] dots
    float* scratch = rs + 2 * n_ex * a.F;
    float* dEw = scratch + wid * (n_ex * exsz);                // slow path: this wave's private accumulators
    float* Ti = scratch;                                        // fast path: [4][PP][So]
    float* Tj = Ti + 4 * PP * So;                               //            [HALVES][4][PP][So]

    // B fragments: rows (tap, p = nt*16 + r) of W, 4 consecutive q per lane - kept for every m tile
    float4 bw[NT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float* src = a.W + (int64_t)(tap * PP + nt * 16 + r) * PP + 4 * kk;
#pragma unroll
        for (int h = 0; h < NT; ++h) bw[nt][h] = *reinterpret_cast<const float4*>(src + 16 * h);
    }
    uint32_t ij[NT];
    if (L0) {
        build_pair_lut(lut, a.F, PP);
        stage_examples(Es, a.Cprev, b0, n_ex, a.B, a.F, a.D, Dp);
        if (!fast)
            for (int e = tid; e < NCOPY * n_ex * exsz; e += NTH) scratch[e] = 0.f;
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) ij[nt] = lut[nt * 16 + r];
    }
    float accj[RM][NT][4];
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) accj[rm][nt][j] = 0.f;

    const int mt_per = (mtiles + HALVES - 1) / HALVES;
    const int mt_lo = half * mt_per, mt_hi = min(mtiles, mt_lo + mt_per);
    const int tpy = So >> 4;                                    // fast path: 16-row tiles per y
    for (int mt = mt_lo; mt < mt_hi; ++mt) {
        const int64_t m0 = wg_m0 + (int64_t)mt * BM;
        float4 av[RM][NT];
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) {
            int64_t m = m0 + rm * 16 + r;
            if (m >= Mend) m = Mend - 1;
#pragma unroll
            for (int h = 0; h < NT; ++h) av[rm][h] = *reinterpret_cast<const float4*>(a.dC + m * PP + 16 * h + 4 * kk);
        }
        // !L0: the mask operand C_{l-1} and the pool gradient do not depend on the MFMAs - fetch them now so that
        // their latency hides behind the matrix work instead of sitting in the epilogue
        float cpre[L0 ? 1 : RM][L0 ? 1 : NT][4], dpre[L0 ? 1 : RM][4];
        int64_t ppos[L0 ? 1 : RM][4];
        if (!L0) {
#pragma unroll
            for (int rm = 0; rm < RM; ++rm)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int64_t m = m0 + rm * 16 + kk * 4 + j;
                    if (m >= Mend) m = Mend - 1;
                    const RowPos rp = row_pos(m, a.lgSo);
                    ppos[rm][j] = (((int64_t)rp.b * Sin + 2 * rp.y + dh) * Sin + 2 * rp.x + dw) * PP + r;
                    dpre[rm][j] = a.dt1[(int64_t)rp.b * a.t1w + a.t1off + 2 * rp.y + dh];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) cpre[rm][nt][j] = a.Cprev[ppos[rm][j] + nt * 16];
                }
        }
        f32x4 acc[RM][NT];
#pragma unroll
        for (int rm = 0; rm < RM; ++rm)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[rm][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < NT; ++h)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) {
                    const float x = t == 0 ? av[rm][h].x : t == 1 ? av[rm][h].y : t == 2 ? av[rm][h].z : av[rm][h].w;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float y = t == 0 ? bw[nt][h].x : t == 1 ? bw[nt][h].y : t == 2 ? bw[nt][h].z : bw[nt][h].w;
                        acc[rm][nt] = mfma16(x, y, acc[rm][nt]);
                    }
                }
        // ---- epilogue ----------------------------------------------------------------------------------
        float si_run[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) si_run[nt] = 0.f;
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) {
            const int64_t mrow = m0 + rm * 16 + kk * 4;
            const RowPos rq = row_pos(mrow < Mend ? mrow : Mend - 1, a.lgSo);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int p = nt * 16 + r;
                if (!L0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (mrow + j < Mend) {
                            const float g = acc[rm][nt][j] + dpre[L0 ? 0 : rm][j];
                            a.dprev[ppos[L0 ? 0 : rm][j] + nt * 16] = g * act_relu_grad(cpre[L0 ? 0 : rm][L0 ? 0 : nt][j], act);
                        }
                    }
                } else {
                    const int fi = ij[nt] & 0xffff, fj = ij[nt] >> 16;
                    const bool pv = p < P;
                    if (fast) {
                        const float ei = Es[fi * Dp + 2 * rq.y + dh];
                        float si = 0.f;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = (pv && mrow + j < Mend) ? acc[rm][nt][j] : 0.f;
                            si += v * Es[fj * Dp + 2 * (rq.x + j) + dw];
                            accj[rm][nt][j] += v * ei;
                        }
                        si += __shfl_xor(si, 16, 64);
                        si += __shfl_xor(si, 32, 64);
                        si_run[nt] += si;
                        if (((rm + 1) & (tpy - 1)) == 0) {       // last tile of this y
                            if (kk == 0) Ti[(tap * PP + p) * So + rq.y] = si_run[nt];
                            si_run[nt] = 0.f;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int64_t m = mrow + j;
                            if (pv && m < Mend) {
                                const RowPos rp = row_pos(m, a.lgSo);
                                const int eb = (rp.b - b0) * exsz;
                                const int io = eb + fi * Dp + 2 * rp.y + dh, jo = eb + fj * Dp + 2 * rp.x + dw;
                                const float v = acc[rm][nt][j];
                                atomicAdd(&dEw[io], v * Es[jo]);
                                atomicAdd(&dEw[jo], v * Es[io]);
                            }
                        }
                    }
                }
            }
        }
    }
    if constexpr (L0) {
        if (fast) {   // x of (rm, kk, j) is the same for every m tile: x = (rm*16 + kk*4 + j) & (So - 1)
            float* tj = Tj + (half * 4 + tap) * PP * So;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int p = nt * 16 + r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int xb = kk * 4 + j;
                    if (a.lgSo == 4) {
                        tj[p * So + xb] = ((accj[0][nt][j] + accj[1][nt][j]) + accj[2][nt][j]) + accj[3][nt][j];
                    } else if (a.lgSo == 5) {
                        tj[p * So + xb] = accj[0][nt][j] + accj[2][nt][j];
                        tj[p * So + 16 + xb] = accj[1][nt][j] + accj[3][nt][j];
                    } else {
#pragma unroll
                        for (int rm = 0; rm < RM; ++rm) tj[p * So + rm * 16 + xb] = accj[rm][nt][j];
                    }
                }
            }
        }
        __syncthreads();
        float* dots = rs + n_ex * a.F;
        for (int e = tid; e < n_ex * a.F; e += NTH) {
            const int b = b0 + e / a.F;
            float s = 0.f, d = 0.f;
            if (b < a.B)
                for (int h = 0; h < a.D; ++h) {
                    const float v = Es[e * Dp + h];
                    s += v;
                    d += v * a.dt1[(int64_t)b * a.t1w + h];
                }
            rs[e] = s; dots[e] = d;
        }
        __syncthreads();
        const int per = a.F * a.D;
        const float invD = 1.f / (float)a.D, invF = 1.f / (float)a.F;
        for (int e = tid; e < n_ex * per; e += NTH) {
            const int row = fast_div(e, invD), h = e - row * a.D, ex = fast_div(row, invF), f = row - ex * a.F, b = b0 + ex;
            if (b >= a.B) continue;
            float R = 0.f, Q = 0.f;
            for (int j = f + 1; j < a.F; ++j) R += rs[ex * a.F + j];
            for (int i = 0; i < f; ++i) Q += dots[ex * a.F + i];
            float conv = 0.f;
            if (fast) {
                const int hh = h >> 1, lo = h & 1;               // h = 2y+dh on the i-side, 2x+dw on the j-side
                const int base_f = f * (2 * a.F - f - 1) / 2;
                for (int j = f + 1; j < a.F; ++j) {              // pairs (f, j): f is the i field
                    const int p = base_f + j - f - 1;
                    conv += Ti[((lo * 2 + 0) * PP + p) * So + hh];
                    conv += Ti[((lo * 2 + 1) * PP + p) * So + hh];
                }
                for (int i = 0; i < f; ++i) {                    // pairs (i, f): f is the j field
                    const int p = i * (2 * a.F - i - 1) / 2 + f - i - 1;
#pragma unroll
                    for (int hf = 0; hf < HALVES; ++hf) {
                        conv += Tj[((hf * 4 + 0 + lo) * PP + p) * So + hh];
                        conv += Tj[((hf * 4 + 2 + lo) * PP + p) * So + hh];
                    }
                }
            } else {
                const int o = row * Dp + h, st = n_ex * exsz;
                conv = scratch[o];
#pragma unroll
                for (int c = 1; c < NCOPY; ++c) conv += scratch[c * st + o];
            }
            a.dprev[(int64_t)b0 * per + e] = conv + a.dt1[(int64_t)b * a.t1w + h] * R + Q;
        }
    }
}

template <int NT, int RM, bool L0, int HALVES>
__global__ __launch_bounds__(256 * HALVES) void dgrad_taps_kernel(DgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    dgrad_taps_body<NT, RM, L0, HALVES>(a, blockIdx.x, smem);
}

// wgrad_taps: wave t accumulates the [PP x PP] weight-gradient block of tap t over its share of the
// workgroup's chunk of rows.  The dC rows of a sub-chunk (<= WGT_SUB rows) are staged ONCE into LDS with
// 16-byte loads and shared by all taps (B' fragments = conflict-free ds_read_b32: the row pitch PP = 16 mod 32
// puts the two k rows of a half-wave on disjoint banks); A' = act(C_{l-1}) patch channels are dword fragment
// loads from L2, or, for layer 0, generated from the embedding tiles held in LDS.  With HALVES = 2 the
// sub-chunk is cut in two and the two partial blocks of a tap are added (lower half first) through LDS.
#define WGT_SUB 256
// tid_in >= 0 (conv01_bwd_kernel): the body runs on ONE group of 256 threads of a larger workgroup, tid_in = thread index inside it;
// store = false: a group without a slab of its own only keeps the barriers
template <int NT, bool GEN, int HALVES, int ACTC = -1>
__device__ __forceinline__ void wgrad_taps_body(const WgradArgs& a, int slab, int nslab, char* smem, int64_t m_lo_o = -1,
                                                int64_t m_hi_o = -1, int tid_in = -1, bool store = true) {
    const int act = ACTC >= 0 ? ACTC : a.act;           // ACTC >= 0: compile-time activation id (README shapes)
    constexpr int PP = NT * 16, UNR = 4, NTH = 256 * HALVES;
    float* Bs = reinterpret_cast<float*>(smem);               // [WGT_SUB][PP]
    uint32_t* lut = reinterpret_cast<uint32_t*>(Bs + WGT_SUB * PP);   // GEN: [PP]
    float* Es = reinterpret_cast<float*>(lut + (GEN ? PP : 0));       // GEN: [n_ex][F][Dp] of the sub-chunk's examples
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63, wid = tid >> 6, tap = wid & 3, half = wid >> 2;
    const int r = lane & 15, kk = lane >> 4;
    const int So = 1 << a.lgSo, Sin = 2 * So, P = a.P, Dp = a.D + 1, S2 = So * So, dh = tap >> 1, dw = tap & 1;
    const int n_ex_max = GEN ? WGT_SUB / S2 + 2 : 0;
    f32x4* red = reinterpret_cast<f32x4*>(Es + (GEN ? (n_ex_max * a.F * Dp + 7) / 4 * 4 : 0));   // HALVES == 2
    const int64_t rows_per_slab = ((a.Mtot + nslab - 1) / nslab + 3) / 4 * 4;
    const int64_t m_lo = m_lo_o >= 0 ? m_lo_o : slab * rows_per_slab;
    const int64_t m_hi = m_lo_o >= 0 ? m_hi_o : min(a.Mtot, m_lo + rows_per_slab);

    f32x4 acc[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bs[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) bs[q] = 0.f;
    int fi[NT], fj[NT];
    if (GEN) {
        build_pair_lut(lut, a.F, PP);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NT; ++i) { const uint32_t ij = lut[i * 16 + r]; fi[i] = (ij & 0xffff) * Dp; fj[i] = (ij >> 16) * Dp; }
    }

    // One example of a small layer (<= 64 rows, the fused README launches): every A' load of the wave is issued before
    // dC is staged, so the two memory latencies overlap and no k-step waits for a load of its own (the generic loop
    // below pays one global-load latency per group of UNR k-steps: 4 in a row at 64 rows).
    bool done_small = false;
    if constexpr (!GEN && HALVES == 1 && NT <= 4) {
        if (m_hi - m_lo <= 16 * UNR) {
            done_small = true;
            const int dbgw = (a.lgSo == 3 && m_lo_o < 0) ? slab : -1;
            PHASE_MARKB(32, dbgw);
            const int nrow = (int)max((int64_t)0, m_hi - m_lo), nk = (nrow + 3) / 4;
            float av[4][UNR][NT];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (g * UNR < nk) {
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        int64_t m = m_lo + 4 * (g * UNR + u) + kk;
                        if (g * UNR + u >= nk || m >= a.Mtot) m = a.Mtot - 1;      // B' is zero there, any finite A' will do
                        const RowPos rp = row_pos(m, a.lgSo);
                        const float* arow = a.in + (((int64_t)rp.b * Sin + 2 * rp.y + dh) * Sin + 2 * rp.x + dw) * PP + r;
#pragma unroll
                        for (int i = 0; i < NT; ++i) av[g][u][i] = arow[16 * i];
                    }
                }
            __syncthreads();                                   // the caller's previous use of this LDS is over
            {
                const float4* src = reinterpret_cast<const float4*>(a.dC + m_lo * PP);
                for (int e = tid; e < nrow * (PP / 4); e += NTH) reinterpret_cast<float4*>(Bs)[e] = src[e];
                for (int e = nrow * (PP / 4) + tid; e < 4 * nk * (PP / 4); e += NTH)
                    reinterpret_cast<float4*>(Bs)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __syncthreads();
            PHASE_MARKB(33, dbgw);
            // (tried and taken out again, DESIGN 3.5 #5: making the A' values opaque here keeps hipcc from hoisting the selu multiply of the
            //  loop below into the four load blocks above - 48 loads in flight in the ISA instead of 4 x 12 with a wait each - and
            //  conv01_bwd_kernel ran 0.40 us SLOWER with it, same box, 2 x 330 launches: profiles/r04_ab_old_new_trace_same_box.txt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (g * UNR < nk) {
#pragma unroll
                    for (int u = 0; u < UNR; ++u)
                        if (g * UNR + u < nk) {
                            float bv[NT];
#pragma unroll
                            for (int q = 0; q < NT; ++q) bv[q] = Bs[(4 * (g * UNR + u) + kk) * PP + q * 16 + r];
#pragma unroll
                            for (int q = 0; q < NT; ++q) bs[q] += bv[q];
#pragma unroll
                            for (int i = 0; i < NT; ++i) {
                                const float x = act_pos(av[g][u][i], act);
#pragma unroll
                                for (int q = 0; q < NT; ++q) acc[i][q] = mfma16(x, bv[q], acc[i][q]);
                            }
                        }
                }
        }
    }
    PHASE_MARKB(34, (a.lgSo == 3 && m_lo_o < 0 && done_small) ? slab : -1);
    for (int64_t ms = m_lo; ms < m_hi && !done_small; ms += WGT_SUB) {       // sub-chunks of the slab's rows
        const int nrow = (int)min((int64_t)WGT_SUB, m_hi - ms);
        const int b_lo = (int)(ms >> (2 * a.lgSo));
        __syncthreads();                                       // previous sub-chunk fully consumed
        {
            const float4* src = reinterpret_cast<const float4*>(a.dC + ms * PP);
            for (int e = tid; e < nrow * (PP / 4); e += NTH) reinterpret_cast<float4*>(Bs)[e] = src[e];
            for (int e = nrow * (PP / 4) + tid; e < ((nrow + 3) / 4 * 4) * (PP / 4); e += NTH)
                reinterpret_cast<float4*>(Bs)[e] = make_float4(0.f, 0.f, 0.f, 0.f);     // pad to a whole k-step
        }
        if (GEN) {
            const int b_hi = (int)((ms + nrow - 1) >> (2 * a.lgSo));
            stage_examples(Es, a.in, b_lo, b_hi - b_lo + 1, a.B, a.F, a.D, Dp);
        }
        __syncthreads();
        // this wave's k-steps (4 rows each) of the sub-chunk
        const int nk = (nrow + 3) / 4;
        int k0 = 0, k1 = nk;
        if (HALVES == 2) {
            const int mid = min(nk, ((nk + 1) / 2 + UNR - 1) / UNR * UNR);
            k0 = half == 0 ? 0 : mid;
            k1 = half == 0 ? mid : nk;
        }
        auto load_a = [&](int k, float (&av)[UNR][NT]) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int row = 4 * (k + u) + kk;
                int64_t m = ms + row;
                if (k + u >= k1 || m >= a.Mtot) m = a.Mtot - 1;     // B' is zero there, any finite A' will do
                const RowPos rp = row_pos(m, a.lgSo);
                if (GEN) {
                    const int eb = (rp.b - b_lo) * a.F * Dp, iy = eb + 2 * rp.y + dh, jx = eb + 2 * rp.x + dw;
#pragma unroll
                    for (int i = 0; i < NT; ++i) av[u][i] = (i * 16 + r < P) ? Es[fi[i] + iy] * Es[fj[i] + jx] : 0.f;
                } else {
                    const float* arow = a.in + (((int64_t)rp.b * Sin + 2 * rp.y + dh) * Sin + 2 * rp.x + dw) * PP + r;
#pragma unroll
                    for (int i = 0; i < NT; ++i) av[u][i] = arow[16 * i];
                }
            }
        };
        if (k0 < k1) {
            float av[UNR][NT], an[UNR][NT];
            load_a(k0, av);
            for (int k = k0; k < k1; k += UNR) {
                const bool more = k + UNR < k1;
                if (more && !GEN) load_a(k + UNR, an);              // global A' of the next step in flight
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    if (k + u < k1) {
                        float bv[NT];
#pragma unroll
                        for (int q = 0; q < NT; ++q) bv[q] = Bs[(4 * (k + u) + kk) * PP + q * 16 + r];
#pragma unroll
                        for (int q = 0; q < NT; ++q) bs[q] += bv[q];
#pragma unroll
                        for (int i = 0; i < NT; ++i) {
                            const float x = GEN ? av[u][i] : act_pos(av[u][i], act);
#pragma unroll
                            for (int q = 0; q < NT; ++q) acc[i][q] = mfma16(x, bv[q], acc[i][q]);
                        }
                    }
                }
                if (more) {
                    if (GEN) load_a(k + UNR, av);
                    else {
#pragma unroll
                        for (int u = 0; u < UNR; ++u)
#pragma unroll
                            for (int i = 0; i < NT; ++i) av[u][i] = an[u][i];
                    }
                }
            }
        }
    }
    if (HALVES == 2) {                                         // upper half -> LDS -> added by the lower half
        __syncthreads();
        if (half == 1) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int q = 0; q < NT; ++q) red[(tap * NT * NT + i * NT + q) * 64 + lane] = acc[i][q];
            if (tap == 0) {
#pragma unroll
                for (int q = 0; q < NT; ++q) reinterpret_cast<float*>(red + 4 * NT * NT * 64)[q * 64 + lane] = bs[q];
            }
        }
        __syncthreads();
        if (half == 1) return;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int q = 0; q < NT; ++q) acc[i][q] += red[(tap * NT * NT + i * NT + q) * 64 + lane];
        if (tap == 0) {
#pragma unroll
            for (int q = 0; q < NT; ++q) bs[q] += reinterpret_cast<float*>(red + 4 * NT * NT * 64)[q * 64 + lane];
        }
    }
    // ---- write this slab -----------------------------------------------------------------------------------
    if (!store) return;
    float* sw = a.slabW + (int64_t)slab * a.slab_stride + (int64_t)tap * PP * PP;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) sw[(i * 16 + kk * 4 + j) * PP + q * 16 + r] = acc[i][q][j];
    if (tap == 0) {
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            float v = bs[q];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kk == 0) a.slabB[(int64_t)slab * a.slabB_stride + q * 16 + r] = v;
        }
    }
}

template <int NT, bool GEN, int HALVES>
__global__ __launch_bounds__(256 * HALVES) void wgrad_taps_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wgrad_taps_body<NT, GEN, HALVES>(a, blockIdx.x, gridDim.x, smem);
}

// Backward of one conv layer l >= 1 in ONE launch: the input gradient (n_d workgroups) and the weight/bias gradient
// (n_w workgroups) only share their inputs, so the two roles run side by side on the chip instead of back to back -
// one launch, one cold-cache ramp and one drain less per layer.  The top layer's launch can also carry the backward
// of the inner branch (n_i workgroups, first in dispatch order: the longest role), which depends on dL/dout only
// and would otherwise sit on a mostly idle chip while the small top layers run.
// tw (top_wgrad_deferred): the weight gradients of the fused top's conv layers, over 64-row slabs - few workgroups, dispatched
// right after the inner role.
struct TopWgrad { WgradArgs w[2]; int n[2]; };
// 4 wavefronts per SIMD up to Pp = 48 (110 VGPRs, nothing spilled): 1024 resident workgroups.  Without the bound the kernel took
// 96 + 36 accumulation registers = 3 per SIMD, and the last 80 of the 848 workgroups of the frappe launch started 9 us late.
template <int NT, int RM, int ACT = -1>
__global__ __launch_bounds__(256, (NT <= 3 ? 4 : 1)) void conv_bwd_pair_kernel(DgradArgs d, WgradArgs w, int n_d, int n_w, InnerBwdArgs ib, int n_i,
                                                            TopWgrad tw, int xcd_align) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int bid = blockIdx.x;
#ifdef CFFM_PHASE_TIMERS
    if (threadIdx.x == 0 && blockIdx.x < 1024) cffm_wg_times[2 * blockIdx.x] = wall_clock64();
#endif
    // dispatch order = longest first: the inner branch, the layer's weight gradient (one workgroup per CU at 256 slabs), the
    // deferred top-layer weight gradients, then the many short input-gradient workgroups, which fill the gaps
    if (bid < n_i) inner_bwd_body<ACT>(ib, bid, n_i, smem);
    else if ((bid -= n_i) < n_w) {
        PHASE_MARKB(23, bid);
        wgrad_taps_body<NT, false, 1, ACT>(w, bid, n_w, smem);
        PHASE_MARKB(24, bid);
    } else if ((bid -= n_w) < tw.n[0]) {
        PHASE_MARKB(30, bid);
        wgrad_taps_body<NT, false, 1, ACT>(tw.w[0], bid, tw.n[0], smem);
        PHASE_MARKB(31, bid);
    } else if ((bid -= tw.n[0]) < tw.n[1]) wgrad_taps_body<NT, false, 1, ACT>(tw.w[1], bid, tw.n[1], smem);
    else {
        // Both the weight- and the input-gradient role read C_{l-1} of their example (A' operand / mask).  Workgroups are
        // dealt to the 8 XCDs round-robin and the weight-gradient workgroup of example e sits on XCD e % 8 when xcd_align is
        // set (the host checks that both roles start at a block index that is a multiple of 8): the input-gradient
        // workgroups of e are mapped to the same XCD, so that its L2 fetches the rows once for both.
        bid -= tw.n[1];
        int wg = bid;
        const int wpe = (1 << (2 * d.lgSo)) / (16 * RM);             // workgroups per example
        if (xcd_align && wpe >= 1) {
            const int x = bid & 7, q = bid >> 3;
            wg = (8 * (q / wpe) + x) * wpe + q % wpe;
        }
        PHASE_MARKB(21, bid);
        dgrad_taps_body<NT, RM, false, 1, ACT>(d, wg, smem);
        PHASE_MARKB(22, bid);
    }
#ifdef CFFM_PHASE_TIMERS
    if (threadIdx.x == 0 && blockIdx.x < 1024) cffm_wg_times[2 * blockIdx.x + 1] = wall_clock64();
#endif
}

// Everything below the fused top for one example per workgroup (bwd_fused01_ok: D = 32, Pp <= 48, the frappe command): with
// 16 wavefronts = 4 groups of four tap-wavefronts,
//   1. the input gradient of layer 1 -> dC_0 of the example (64 rows: one 16-row tile per group),
//   2. the weight gradients of layers 1, 2, 3 side by side, one group each (64 / 16 / 4 rows: the small path of
//      wgrad_taps_body, two barriers for every group), slab = example,
//   3. the factorised backward of layer 0, which finds dC_0 in this CU's L2 slice instead of HBM.
// One launch boundary and one operand burst less than conv_bwd_pair_kernel + conv0_fact_bwd_kernel, and the top-layer weight
// gradients leave the critical chain of bwd_top_kernel.
struct Conv01Args {
    DgradArgs d0;                 // layer 0
    float *slabW0, *slabB0;
    int64_t stride0;
    DgradArgs d1;                 // input gradient of layer 1
    WgradArgs w1, w2, w3;         // weight gradients of layers 1, 2, 3
    int n2, n3;                   // slabs of layers 2 and 3, CFFM_TOP_SLAB_ROWS rows each (workgroup b < n carries slab b of that layer)
};

template <int NT, int F_, int D_, int ACT>
__global__ __launch_bounds__(1024) void conv01_bwd_kernel(Conv01Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // grp as a SCALAR (readfirstlane): whatever is selected by it below is selected per wavefront with scalar branches.  (A first
    // version branched on the per-lane value around code with barriers: that is compiled to exec masking, under which a
    // wavefront can run through the barriers of a branch it did not take - the launch hung.)
    const int b = blockIdx.x, grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)), ltid = threadIdx.x & 255;
    PHASE_MARKB(40, b);
    dgrad_taps_body<NT, 1, false, 4, ACT>(a.d1, b, smem, 0, -1, 4);
    PHASE_MARKB(41, b);
    {   // ONE call site for the four groups: its barriers sit in straight-line code that every wavefront runs.  Group 0:
        // layer 1, slab = example b.  Groups 1 / 2: layers 2 / 3 over slabs of CFFM_TOP_SLAB_ROWS rows (2 / 8 examples each, their
        // dC is complete since the fused top), carried by the first n2 / n3 workgroups only - one slab per example would cost
        // 2 x 256 slabs of 4*Pp*Pp floats for 5,120 rows of work.  Group 3 only keeps the barriers.  Measured and not kept:
        // the k-steps of layer 1 shared between groups 0 and 3 with an exchange through LDS (33.8 -> 34.9 us), and the input
        // gradient on groups 2 / 3 at the same time as the weight gradients on groups 0 / 1 (34.6): these phases are bound by
        // the L2-level traffic of the 256 workgroups (64 MB in 12 us), not by MFMA time or by a wavefront's latency chain.  Nor
        // did cutting that traffic help: C_0 of the example staged ONCE in LDS (coalesced 16-byte loads issued first, parked behind
        // the input gradient's own loads) for both the mask and the A' operand - 12.6 MB less - ran 32.9 -> 35.7 us.
        char* gsm = smem + (size_t)grp * (64 * NT * 16 * 4);
        const WgradArgs& w = grp == 0 ? a.w1 : (grp == 1 ? a.w2 : a.w3);
        const int nsl = grp == 0 ? (int)gridDim.x : (grp == 1 ? a.n2 : a.n3);
        // the slabs of layer 2 go to the LAST n2 workgroups, those of layer 3 to the first n3: no workgroup carries all three
        const int slab = grp == 1 ? b - ((int)gridDim.x - a.n2) : b;
        const bool mine = grp < 3 && slab >= 0 && slab < nsl;
        int64_t m_lo = 0, m_hi = 0;
        if (mine) {
            const int64_t per = grp == 0 ? (1ll << (2 * w.lgSo)) : CFFM_TOP_SLAB_ROWS;
            m_lo = (int64_t)slab * per;
            m_hi = min(w.Mtot, m_lo + per);
            if (m_hi < m_lo) m_hi = m_lo;
        }
        wgrad_taps_body<NT, false, 1, ACT>(w, mine ? slab : 0, nsl, gsm, m_lo, m_hi, ltid, mine);
    }
    __syncthreads();                                         // the weight-gradient LDS is free, dC_0 is complete
    PHASE_MARKB(42, b);
    conv0_fact_bwd_body<NT, F_, D_, 16>(a.d0, a.slabW0, a.slabB0, a.stride0, smem, b, (int)gridDim.x);
}

// Top of the backward in ONE launch (B <= 256, Pp <= 64): workgroup b runs, for example b, the head backward and then
// the weight- and input-gradient of the top two conv layers (4 and 16 rows per example at D = 32) - five launch
// boundaries of the per-stage pipeline collapse into barriers of one workgroup.  A second block range carries the
// inner-branch backward, which recomputes dL/dout from (out, y, L) itself so that it does not wait for the first range.
struct BwdTopArgs {
    HeadBwdArgs hb;
    DgradArgs d[2];
    WgradArgs w[2];
    int lgSo[2];
    int n_layers;                 // 1 or 2 conv layers (top first)
    int wgrad_here;               // 0: the weight gradients of these layers run in the pair launch that follows (top_wgrad_deferred)
    InnerBwdArgs ib;
    int n_inner;                  // workgroups of the inner-branch role (0: none)
    int n_keys;                   // > 0: the key placement runs in that many workgroups of its own (3 x 256 workgroups of 4 wavefronts fit the chip)
    const int32_t* rank_ids;      // non-NULL: the inner-branch role also places the sparse-update keys of its examples
    unsigned long long* keys_sorted;   // (moved here from the forward launch, where it sat on the critical path)
    int n_rows;
};

template <int NT, int ACT = -1>
__global__ __launch_bounds__(256) void bwd_top_kernel(BwdTopArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float dh1s[CFFM_HEAD_UNITS];
    __shared__ float dt1s[1024];
    __shared__ float red[4];
    const int bid = blockIdx.x;
#ifdef CFFM_PHASE_TIMERS
    if (threadIdx.x == 0) cffm_wg_times[2 * blockIdx.x] = wall_clock64();
#endif
    if (bid < a.n_inner) {                                   // ---- role 1: inner branch
        PHASE_MARKB(36, bid);
        const float L = head_bwd_loss(a.hb, false, red);
        __syncthreads();
        PHASE_MARKB(37, bid);
        if (a.rank_ids != nullptr && a.n_keys == 0) {
            for (int b = bid; b < a.hb.B; b += a.n_inner) rank_keys_body<4>(a.rank_ids, a.n_rows, b, a.hb.g.F, a.keys_sorted, smem);
        }
        PHASE_MARKB(38, bid);
        inner_bwd_body<ACT>(a.ib, bid, a.n_inner, smem, L);
        PHASE_MARKB(39, bid);
#ifdef CFFM_PHASE_TIMERS
        if (threadIdx.x == 0) cffm_wg_times[2 * blockIdx.x + 1] = wall_clock64();
#endif
        return;
    }
    if (bid < a.n_inner + a.n_keys) {                        // ---- role 3 (n_keys > 0): the key placement as workgroups of its own
        for (int b = bid - a.n_inner; b < a.hb.B; b += a.n_keys) rank_keys_body<4>(a.rank_ids, a.n_rows, b, a.hb.g.F, a.keys_sorted, smem);
#ifdef CFFM_PHASE_TIMERS
        if (threadIdx.x == 0) cffm_wg_times[2 * blockIdx.x + 1] = wall_clock64();
#endif
        return;
    }
    const int b = bid - a.n_inner - a.n_keys;                // ---- role 2: example b (also slab b of every range)
    PHASE_MARKB(0, b);
    HeadBwdState st;
    head_bwd_begin(a.hb, b, st);
    PHASE_MARKB(1, b);
    const float L = head_bwd_loss(a.hb, b == 0, red);
    PHASE_MARKB(2, b);
    if (b < a.hb.B)
        head_bwd_example<ACT>(a.hb, b, st, b, head_dout(a.hb.loss, a.hb.out[b], a.hb.y[b], 1.f / (float)a.hb.Bg, L), dh1s, dt1s);
    PHASE_MARKB(3, b);
    head_bwd_end(a.hb, b, st);
    PHASE_MARKB(4, b);
    for (int t = 0; t < a.n_layers; ++t) {
        __syncthreads();                                     // dC of this layer (global, written above) is complete
        const int64_t rows = 1ll << (2 * a.lgSo[t]);
        const int64_t m_lo = (int64_t)b * rows, m_hi = b < a.hb.B ? m_lo + rows : m_lo;
        if (a.wgrad_here) wgrad_taps_body<NT, false, 1, ACT>(a.w[t], b, (int)gridDim.x - a.n_inner - a.n_keys, smem, m_lo, m_hi);
        PHASE_MARKB(5 + 2 * t, b);
        for (int64_t m0 = m_lo; m0 < m_hi; m0 += 16) dgrad_taps_body<NT, 1, false, 1, ACT>(a.d[t], 0, smem, m0, m_hi);
        PHASE_MARKB(6 + 2 * t, b);
    }
#ifdef CFFM_PHASE_TIMERS
    if (threadIdx.x == 0) cffm_wg_times[2 * blockIdx.x + 1] = wall_clock64();
#endif
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
#define DISPATCH_NT(NTV, CALL)                     \
    switch (NTV) {                                 \
        case 1: { constexpr int NT_ = 1; CALL; } break; \
        case 2: { constexpr int NT_ = 2; CALL; } break; \
        case 3: { constexpr int NT_ = 3; CALL; } break; \
        case 4: { constexpr int NT_ = 4; CALL; } break; \
        case 6: { constexpr int NT_ = 6; CALL; } break; \
        default: { constexpr int NT_ = 8; CALL; } break; \
    }

template <class KernelT>
static inline int set_lds(KernelT k, size_t lds) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

// bf16x3 K loop for the 128 x 128 instance of the direct layers (the one the wide shapes run); CFFM_CONV_FP32=1: the fp32 MFMA loop
static inline bool conv_b3_on() {
    static const bool on = getenv("CFFM_CONV_FP32") == nullptr;
    return on;
}
template <int NT, int RM, bool GEN>
static int launch_conv_fwd(const ConvArgs& a, int nblk, hipStream_t st) {
    constexpr int BM = 64 * RM;
    const int S2 = 1 << (2 * a.lgSo);
    const int n_ex = GEN ? (BM > S2 ? BM / S2 : 1) : 0;
    ConvArgs b = a;
    b.nblk = nblk;
    const int64_t nb1 = 8 * xcd_per((a.Mtot + BM - 1) / BM) * nblk;
    if (nb1 > 0x7fffffffll) return CFFM_ERR_UNSUPPORTED;
    if constexpr (NT == 8 && RM == 2 && !GEN) {
        if (conv_b3_on() && b.wb3 != nullptr) {
            const size_t lds3 = (size_t)gemm_b3_lds_bytes<NT>() + 16;
            int rc3 = set_lds(conv_fwd_kernel<NT, RM, GEN, true>, lds3);
            if (rc3) return rc3;
            b.wb3_npad = (a.Pp + 127) / 128 * 128;        // the filter changes every step: split it once for this launch
            if ((rc3 = pack_w_b3<false>(a.W, a.Pp, 4 * a.Pp, a.Pp, b.wb3, st))) return rc3;
            hipLaunchKernelGGL((conv_fwd_kernel<NT, RM, GEN, true>), dim3((unsigned)nb1), dim3(256), lds3, st, b);
            CFFM_CHECK_LAUNCH();
            return 0;
        }
    }
    const size_t lds = (size_t)(2 * KSTEP * (NT * 16 + 4) + (GEN ? a.Pp + n_ex * a.F * (a.D + 1) : 0) + 4) * 4;
    int rc = set_lds(conv_fwd_kernel<NT, RM, GEN>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((conv_fwd_kernel<NT, RM, GEN>), dim3((unsigned)nb1), dim3(256), lds, st, b);
    CFFM_CHECK_LAUNCH();
    return 0;
}

template <int NT, int RM, bool L0>
static int launch_dgrad(const DgradArgs& a, int nblk, hipStream_t st) {
    constexpr int BM = 64 * RM;
    const int S2 = 1 << (2 * a.lgSo);
    const int rows_per_wg = L0 ? (S2 > BM ? S2 : BM) : BM;
    const int n_ex = L0 ? rows_per_wg / S2 : 0;
    const size_t lds = (size_t)(2 * KSTEP * (NT * 16 + 4) + (L0 ? a.Pp + 5 * n_ex * a.F * (a.D + 1) + 2 * n_ex * a.F : 0) + 4) * 4;
    DgradArgs b = a;
    b.nblk = L0 ? 1 : nblk;
    const int64_t nb1 = 8 * xcd_per((a.Mtot + rows_per_wg - 1) / rows_per_wg) * b.nblk;
    if (nb1 > 0x7fffffffll) return CFFM_ERR_UNSUPPORTED;
    if constexpr (NT == 8 && RM == 2 && !L0) {
        if (conv_b3_on() && b.wb3 != nullptr) {
            const size_t lds3 = (size_t)gemm_b3_lds_bytes<NT>() + 16;
            int rc3 = set_lds(dgrad_kernel<NT, RM, L0, true>, lds3);
            if (rc3) return rc3;
            b.wb3_npad = (4 * a.Pp + 127) / 128 * 128;
            if ((rc3 = pack_w_b3<true>(a.W, a.Pp, a.Pp, 4 * a.Pp, b.wb3, st))) return rc3;
            hipLaunchKernelGGL((dgrad_kernel<NT, RM, L0, true>), dim3((unsigned)nb1), dim3(256), lds3, st, b);
            CFFM_CHECK_LAUNCH();
            return 0;
        }
    }
    int rc = set_lds(dgrad_kernel<NT, RM, L0>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((dgrad_kernel<NT, RM, L0>), dim3((unsigned)nb1), dim3(256), lds, st, b);
    CFFM_CHECK_LAUNCH();
    return 0;
}

template <int NT, bool GEN>
static int launch_wgrad(const WgradArgs& a, hipStream_t st) {
    constexpr int BI = 64, BQ = NT * 16, LDB = BQ + ((NT & 1) ? 0 : 16);
    const int S2 = 1 << (2 * a.lgSo);
    const int n_ex = GEN ? (WG_KM > S2 ? WG_KM / S2 : 1) : 0;
    const size_t lds = (size_t)(WG_KM * LDB + (GEN ? a.Pp + n_ex * a.F * (a.D + 1) : WG_KM * (BI + 16)) + 4) * 4;
    int rc = set_lds(wgrad_kernel<NT, GEN>, lds);
    if (rc) return rc;
    WgradArgs b = a;
    b.nslab = CFFM_NSLAB;                                                       // Pp > 64: conv_slabs() == CFFM_NSLAB
    b.nxy = ((4 * a.Pp + BI - 1) / BI) * a.qblocks;
    dim3 grid((unsigned)(8 * xcd_per(b.nslab) * b.nxy));
    hipLaunchKernelGGL((wgrad_kernel<NT, GEN>), grid, dim3(256), lds, st, b);
    CFFM_CHECK_LAUNCH();
    return 0;
}

template <int NT>
static int launch_wgrad2(const WgradArgs& a, hipStream_t st) {
    constexpr int BI = 128, BQ = NT * 16;
    WgradArgs b = a;
    b.nslab = CFFM_NSLAB;                                                       // Pp > 64: conv_slabs() == CFFM_NSLAB
    b.nxy = ((4 * a.Pp + BI - 1) / BI) * a.qblocks;
    if constexpr (NT == 8) {
        if (conv_b3_on()) {                                 // the same tile on the bf16 pipe (bf16x3 split, fp32 accumulate)
            const size_t lds3 = (size_t)3 * 4 * (BI + BQ) * 16 + 16;
            int rc3 = set_lds(wgrad3_kernel<NT>, lds3);
            if (rc3) return rc3;
            hipLaunchKernelGGL((wgrad3_kernel<NT>), dim3((unsigned)(8 * xcd_per(b.nslab) * b.nxy)), dim3(256), lds3, st, b);
            CFFM_CHECK_LAUNCH();
            return 0;
        }
    }
    const size_t lds = (size_t)(WG_KM * BI + WG_KM * BQ) * 4 + 16;
    int rc = set_lds(wgrad2_kernel<NT>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((wgrad2_kernel<NT>), dim3((unsigned)(8 * xcd_per(b.nslab) * b.nxy)), dim3(256), lds, st, b);
    CFFM_CHECK_LAUNCH();
    return 0;
}

template <int NT, int RM, bool GEN>
static int launch_conv_fwd_taps(const ConvArgs& a, hipStream_t st) {
    constexpr int PP = NT * 16, BM = 16 * RM;
    const int S2 = 1 << (2 * a.lgSo);
    const int n_ex = GEN ? (BM > S2 ? BM / S2 : 1) : 0;
    size_t lds = (size_t)4 * PP * (PP + 4) * 4;
    const size_t red = (size_t)4 * RM * NT * 64 * 16;
    if (red > lds) lds = red;
    lds += (size_t)(GEN ? PP + n_ex * a.F * (a.D + 1) : 0) * 4 + 16;
    int rc = set_lds(conv_fwd_taps_kernel<NT, RM, GEN>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((conv_fwd_taps_kernel<NT, RM, GEN>), dim3((unsigned)((a.Mtot + BM - 1) / BM)), dim3(256), lds, st, a);
    CFFM_CHECK_LAUNCH();
    return 0;
}

template <int NT, int RM, bool GEN>
static int launch_conv_fwd_rows(const ConvArgs& a, hipStream_t st) {
    constexpr int PP = NT * 16, BM = 64 * RM;
    const int S2 = 1 << (2 * a.lgSo);
    const int n_ex = GEN ? (BM > S2 ? BM / S2 : 1) : 0;
    const size_t lds = (size_t)4 * PP * PP * 4 + (size_t)(GEN ? PP + n_ex * a.F * (a.D + 1) : 0) * 4 + 16;
    int rc = set_lds(conv_fwd_rows_kernel<NT, RM, GEN>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((conv_fwd_rows_kernel<NT, RM, GEN>), dim3((unsigned)((a.Mtot + BM - 1) / BM)), dim3(256), lds, st, a);
    CFFM_CHECK_LAUNCH();
    return 0;
}

// Layer 0 in factorised form for WIDE filters (Pp > 64, e.g. F = 32: P = 496): the T planes of all channels no longer
// fit LDS (4 MB per example), so a workgroup takes one example, ONE tile of 16 output channels q0 .. q0+15 and ONE tile
// of 16 output columns x0 .. x0+15 (T[.., x, q] depends on nothing outside its (x, q)):
//   step 1  T[dh][i][x][q] = sum_{dw, j>i} E_j[2x+dw] * W[dh,dw,(i,j),q]     2(F-1) units, K = 2(F-1-i)
//   step 2  C[y][x][q]     = relu(b[q] + sum_{dh,i} E_i[2y+dh] * T[dh][i][x][q])        S/16 row tiles, K = 2F
// 15.7x fewer MFMAs than the direct contraction at F = 32, D = 64 (128 MFLOP against 2,015 per example).  W fragments come
// straight from L2 (a 16-channel column slice of the filter, 127 KB, shared by every example of the tile: the grid is
// tile-major); T (2F planes of 16 x 16, 70 KB: two workgroups per CU) and the embedding tile live in LDS.
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void conv0_fact_tile_fwd_kernel(ConvArgs a) {
    constexpr int NTH = 64 * NW, XQ = 16 / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int F = a.F, D = a.D, S = D / 2, Dp = D + 1, RT = S / 16, PpT = a.Pp;
    constexpr int TP = 16 * 16 + 16;
    float* T = reinterpret_cast<float*>(smem);                 // [2F][TP]
    float* Es = T + 2 * F * TP;                                 // [F][Dp]
    float* PS = Es + F * Dp;                                    // [NW][S] pool partials of the wavefronts
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kk = lane >> 4;
    int bid = blockIdx.x;
    const int b = bid % a.B; bid /= a.B;
    const int xt = bid % RT, qt = bid / RT, q0 = qt * 16, x0 = xt * 16;
#ifdef CFFM_TILE_DBG
    const int dbg = a.dbg;
#else
    constexpr int dbg = 0;
#endif
    stage_example_rows(Es, a.in, a.idx, a.idxM, b, F, D, Dp, tid, NTH, a.idxStride);
    lds_barrier();
    // ---- step 1 -----------------------------------------------------------------------------------------------------
    // The kernel is instruction-issue-bound (with every load, MFMA and store switched off it still ran 11.6 of its 29.2 ms
    // at F32 D64: tools/dbg_tile.py), and most of those instructions were the per-unit operand bookkeeping.  So the k index
    // runs over ALL fields, k = dw * F4 + j (F4 = F rounded up to 4): the E fragment of a k-step then does not depend on the
    // unit (16 LDS reads per WAVE instead of per unit), the W fragment is one load at a lane-constant offset from a
    // wave-uniform row pointer, and only the first k-quad of a unit (the one that contains j = i + 1) needs a mask.
    const int units = 2 * (F - 1);
    const int F4 = (F + 3) & ~3, nq = F4 >> 2;                  // k-quads per dw; 2 * nq <= C0T_MAXKS
    float avA[C0T_MAXKS];
    int woff[C0T_MAXKS];
    {
        const int x = x0 + r;
#pragma unroll
        for (int ks = 0; ks < C0T_MAXKS; ++ks) {
            const int dwk = ks >= nq ? 1 : 0, j = 4 * (ks - dwk * nq) + kk;
            const bool valid = ks < 2 * nq && j < F;
            const float ev = Es[(valid ? j : 0) * Dp + 2 * x + dwk];
            avA[ks] = valid ? ev : 0.f;
            woff[ks] = valid ? (dwk * PpT + j) * PpT + q0 + r : q0 + r;
        }
    }
    for (int u = wave; u < units; u += NW) {
        const int i = u % (F - 1), dh = u / (F - 1);
        const int base = i * (2 * F - i - 1) / 2;               // first pair (i, i+1); pair (i, j) is row base + j - i - 1
        const int s0 = (i + 1) >> 2;                            // first k-quad with a j > i
        const float* wb = a.W + ((int64_t)(dh * 2) * PpT + base - i - 1) * PpT;     // + woff = row of (dw, j), column q
        float bw[C0T_MAXKS];
#pragma unroll
        for (int ks = 0; ks < C0T_MAXKS; ++ks) {
            const int dwk = ks >= nq ? 1 : 0, t = ks - dwk * nq;
            bw[ks] = 0.f;
            if (t >= s0 && ks < 2 * nq) {                       // wave-uniform
                if (t == s0) {                                  // the quad that holds j = i + 1: lanes with j <= i are masked
                    const int j = 4 * t + kk;                   // (their address is moved onto the row of j = i + 1)
                    const float wv = wb[woff[ks] + (j > i ? 0 : (i + 1 - j) * PpT)];
                    bw[ks] = (j > i && !(dbg & 2)) ? wv : ((dbg & 2) ? 1.f : 0.f);
                } else {
                    bw[ks] = (dbg & 2) ? 1.f : wb[woff[ks]];
                }
            }
        }
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!(dbg & 4))
#pragma unroll
        for (int ks = 0; ks < C0T_MAXKS; ++ks) {
            const int dwk = ks >= nq ? 1 : 0, t = ks - dwk * nq;
            if (t >= s0 && ks < 2 * nq) acc = mfma16(avA[ks], bw[ks], acc);      // wave-uniform skip of the empty k-quads
        }
        float* tp = T + (dh * F + i) * TP + r;
#pragma unroll
        for (int j = 0; j < 4; ++j) tp[(kk * 4 + j) * 16] = acc[j];
    }
    for (int e = tid; e < 2 * 16 * 16; e += NTH) {            // planes (dh, F-1) have no pairs
        const int dh = e / 256, o = e - dh * 256;
        T[(dh * F + F - 1) * TP + o] = 0.f;
    }
    lds_barrier();
    // ---- step 2 -----------------------------------------------------------------------------------------------------
    const int K2 = 2 * F, ks2 = (K2 + 3) / 4;
    const float bias = a.bias[q0 + r];
    for (int rt = 0; rt < RT; ++rt) {
        const int y = rt * 16 + r;
        const int xg = wave * XQ;                               // NW * XQ = 16 columns of this tile
        f32x4 acc[XQ];
#pragma unroll
        for (int q4 = 0; q4 < XQ; ++q4) acc[q4] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s2 = 0; s2 < ((dbg & 8) ? 0 : ks2); ++s2) {
            const int k = 4 * s2 + kk;
            const bool ok = k < K2;
            const int dh = (ok && k >= F) ? 1 : 0, i = ok ? k - dh * F : 0;
            const float av = ok ? Es[i * Dp + 2 * y + dh] : 0.f;
            const float* tb = T + (ok ? k : 0) * TP + xg * 16 + r;
#pragma unroll
            for (int q4 = 0; q4 < XQ; ++q4) acc[q4] = mfma16(av, ok ? tb[q4 * 16] : 0.f, acc[q4]);
        }
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q4 = 0; q4 < XQ; ++q4)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int yy = rt * 16 + kk * 4 + j;
                const float c = fmaxf(acc[q4][j] + bias, 0.f);
                if (!(dbg & 1) || c == 12345.678f)
                    a.out[(((int64_t)b * S + yy) * S + x0 + xg + q4) * PpT + q0 + r] = c;
                ps[j] += act_pos(c, a.act);
            }
        if (a.pool != nullptr) {                                // pool partial of this (column tile, channel tile): over q in the DPP row
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = row_group_sum<true>(ps[j]);
                if (r == 0) PS[(wave * RT + rt) * 16 + kk * 4 + j] = v;          // [NW][S], behind the embedding tile
            }
        }
    }
    if (a.pool != nullptr) {
        lds_barrier();
        if (tid < S) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += PS[w * S + tid];                      // wave order: x0 + w * XQ ascending
            a.pool[((int64_t)b * S + tid) * a.pool_np + xt * (PpT / 16) + qt] = v;
        }
    }
}

// Weight / bias gradient of layer 0 for wide filters in factorised form (the direct contraction is 2,015 MFLOP per
// example at F32 D64, this is 128):
//   dT[dh][i][x][q]        = sum_y E_i[2y+dh] * dC[y][x][q]                  rows (dh,i), K = y, cols (x,q)
//   dW[dh][dw][(i,j)][q]  += sum_x E_j[2x+dw] * dT[dh][i][x][q]              rows (dw,j>i), K = x, cols q
// Workgroup = (16-channel tile q0, group g of 16 (dh,i) rows, gradient slab): it walks the examples of its slab and their
// S/16 column tiles, keeps the dW blocks of its 16 units in registers (<= 4 row tiles each, 4 units per wavefront) and
// writes them once.  dC tile [S][16][16] and dT [16][16][16] live in LDS (41 KB: two workgroups per CU).
template <int SMAX, int NW>
__global__ __launch_bounds__(64 * NW, 2) void conv0_fact_tile_wgrad_kernel(WgradArgs a, int nslab) {
    constexpr int NTH = 64 * NW, UPW = 16 / NW;                // units (and phase-C columns) per wavefront
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int F = a.F, D = a.D, S = D / 2, Dp = D + 1, RT = S / 16, PpT = a.Pp, P = a.P, G = (2 * F + 15) / 16;
    float* Es = reinterpret_cast<float*>(smem);                // [F][Dp]
    float* dCt = Es + (F * Dp + 3) / 4 * 4;                    // [S][16 x][16 q]
    float* dTg = dCt + SMAX * 256;                              // [16 m][16 x][16 q]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kk = lane >> 4;
    int bid = blockIdx.x;
    const int slab = bid % nslab; bid /= nslab;
    const int g = bid % G, qt = bid / G, q0 = qt * 16;
    float* sw = a.slabW + (int64_t)slab * a.slab_stride;
    float* sb = a.slabB + (int64_t)slab * a.slabB_stride;
    // this wave's four units (dh, i) and the A-row identity of this lane for phase C
    const int mC = g * 16 + r;                                  // phase C row of this lane
    const bool mC_ok = mC < 2 * F;
    const int dhC = mC_ok && mC >= F ? 1 : 0, iC = mC_ok ? mC - dhC * F : 0;
    f32x4 accD[UPW][4];
#pragma unroll
    for (int u4 = 0; u4 < UPW; ++u4)
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) accD[u4][t4] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;                                          // bias partial of channel q0 + (tid & 15), rows tid >> 4 (g == 0 only)
    for (int b = slab; b < a.B; b += nslab) {
        __syncthreads();                                       // previous example fully consumed
        stage_example_rows(Es, a.in, a.idx, a.idxM, b, F, D, Dp, tid, NTH, a.idxStride);
        for (int xt = 0; xt < RT; ++xt) {
            const int x0 = xt * 16;
            if (xt > 0) __syncthreads();                       // dCt / dTg of the previous column tile consumed
            // dC[b][y][x0 + x][q0 .. q0+15] -> dCt[y][x][q] : 64-byte pieces
            for (int e4 = tid; e4 < S * 16 * 4; e4 += NTH) {
                const int q4 = e4 & 3, x = (e4 >> 2) & 15, y = e4 >> 6;
                const float4 v = *reinterpret_cast<const float4*>(a.dC + (((int64_t)b * S + y) * S + x0 + x) * PpT + q0 + 4 * q4);
                *reinterpret_cast<float4*>(dCt + (y * 16 + x) * 16 + 4 * q4) = v;
            }
            __syncthreads();
            if (g == 0) {                                      // db[q] += sum_{y,x} dC: thread (q = tid & 15, part = tid >> 4)
                const int q = tid & 15, part = tid >> 4;
                for (int e = part; e < S * 16; e += NTH / 16) bsum += dCt[e * 16 + q];
            }
            // ---- phase C: dT rows mC (this group), wave's x columns 4*wave .. 4*wave+3 ------------------------------
            {
                f32x4 acc[UPW];
#pragma unroll
                for (int xl = 0; xl < UPW; ++xl) acc[xl] = (f32x4){0.f, 0.f, 0.f, 0.f};
                for (int s4 = 0; s4 < S / 4; ++s4) {
                    const int y = 4 * s4 + kk;
                    const float av = mC_ok ? Es[iC * Dp + 2 * y + dhC] : 0.f;
#pragma unroll
                    for (int xl = 0; xl < UPW; ++xl) acc[xl] = mfma16(av, dCt[(y * 16 + UPW * wave + xl) * 16 + r], acc[xl]);
                }
#pragma unroll
                for (int xl = 0; xl < UPW; ++xl)
#pragma unroll
                    for (int j = 0; j < 4; ++j) dTg[((kk * 4 + j) * 16 + UPW * wave + xl) * 16 + r] = acc[xl][j];
            }
            __syncthreads();
            // ---- phase D: the wave's four units -----------------------------------------------------------------------
#pragma unroll
            for (int u4 = 0; u4 < UPW; ++u4) {
                const int ml = wave * UPW + u4, m = g * 16 + ml;
                if (m >= 2 * F) continue;
                const int dh = m >= F ? 1 : 0, i = m - dh * F;
                (void)dh;
                const int nj = F - 1 - i, K2 = 2 * nj;
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    if (t4 * 16 >= K2) continue;
                    const int m2 = t4 * 16 + r;
                    const bool ok = m2 < K2;
                    const int dw = (ok && m2 >= nj) ? 1 : 0, jj = ok ? m2 - dw * nj : 0;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int x = 4 * s4 + kk;
                        const float av = ok ? Es[(i + 1 + jj) * Dp + 2 * (x0 + x) + dw] : 0.f;
                        accD[u4][t4] = mfma16(av, dTg[(ml * 16 + x) * 16 + r], accD[u4][t4]);
                    }
                }
            }
        }
    }
    // ---- write this workgroup's part of the slab ------------------------------------------------------------------
#pragma unroll
    for (int u4 = 0; u4 < UPW; ++u4) {
        const int m = g * 16 + wave * UPW + u4;
        if (m >= 2 * F) continue;
        const int dh = m >= F ? 1 : 0, i = m - dh * F;
        const int nj = F - 1 - i, K2 = 2 * nj, base = i * (2 * F - i - 1) / 2;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            if (t4 * 16 >= K2) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m2 = t4 * 16 + kk * 4 + j;
                if (m2 < K2) {
                    const int dw = m2 >= nj ? 1 : 0, jj = m2 - dw * nj;
                    sw[((int64_t)(dh * 2 + dw) * PpT + base + jj) * PpT + q0 + r] = accD[u4][t4][j];
                }
            }
        }
    }
    if (g == 0) {
        // rows of padded pairs (p >= P) are never produced: they must read as zeros in the reduction
        for (int e = tid; e < 4 * (PpT - P) * 16; e += NTH) {
            const int q = e & 15, rest = e >> 4, p = P + rest % (PpT - P), tap = rest / (PpT - P);
            sw[((int64_t)tap * PpT + p) * PpT + q0 + q] = 0.f;
        }
        __syncthreads();
        float* red = dTg;                                      // [NTH/16 parts][16 q]
        red[(tid >> 4) * 16 + (tid & 15)] = bsum;
        __syncthreads();
        if (tid < 16) {
            float v = 0.f;
#pragma unroll
            for (int part = 0; part < NTH / 16; ++part) v += red[part * 16 + tid];
            sb[q0 + tid] = v;
        }
    }
}

// Round 3: the same contraction with ALL (dh,i) groups of a (q tile, slab) in ONE workgroup of 16 wavefronts (group = wave >> 2,
// four wavefronts and four units each).  The version above gives every group a workgroup of its own, and each of them stages the
// SAME dC tile: rocprofv3 counted 51.6 GB of FETCH_SIZE per launch (103 GB with the gfx950 correction) against a 16.6 GB dC_0 -
// the four workgroups of a tile sit on one XCD but drift apart, so the L2 does not catch the re-reads, and at 3.7 TB/s the kernel was
// bound by that traffic.  Here the tile is staged once per (example, column tile) and feeds the four groups out of LDS (dCt 32 KB
// shared + one dT block of 16 KB per group + the embedding tile = 105 KB: one workgroup of 16 wavefronts per CU, the same number of
// wavefronts per CU as two 8-wavefront workgroups).
// The dC tile of the NEXT (example, column tile) and the embedding rows of the next example go HBM/L2 -> LDS directly
// (global_load_lds: 16-byte pieces land at 16 x their index, which IS the dCt layout; one 4-byte piece per lane = one embedding
// row per wave instruction at the padded pitch) into the other half of a double buffer while the current tile is computed: two
// barriers per tile instead of three and no load latency in front of phase C.
// GPW groups per workgroup (4: the tile is read once, 4 units per wavefront and 64 accumulation registers; 2: read twice, 2 units per
// wavefront, 32 accumulation registers and nothing spilled under the 128-register budget of a 1024-thread workgroup)
template <int SMAX, int GPW>
__global__ __launch_bounds__(1024) void conv0_fact_tile_wgrad_all_kernel(WgradArgs a, int nslab) {
    constexpr int NTH = 1024, WPG = 16 / GPW, UPW = 16 / WPG;   // wavefronts per group; units (and phase-C columns) per wavefront
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int F = a.F, D = a.D, S = D / 2, Dp = D + 1, RT = S / 16, PpT = a.Pp, P = a.P, G = (2 * F + 15) / 16;
    const int EsN = (F * Dp + 3) / 4 * 4;
    float* Es0 = reinterpret_cast<float*>(smem);               // [2][F][Dp]
    float* dCt0 = Es0 + 2 * EsN;                                // [2][S][16 x][16 q]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kk = lane >> 4;
    int bid = blockIdx.x;
    const int slab = bid % nslab; bid /= nslab;
    const int gp = bid % (4 / GPW), qt = bid / (4 / GPW), q0 = qt * 16;          // gp: which GPW of the four groups this workgroup carries
    const int gl = __builtin_amdgcn_readfirstlane(wave / WPG), g = gp * GPW + gl;   // group (scalar)
    const int wl = wave % WPG, gtid = tid % (64 * WPG);                             // wave / thread inside the group
    float* dTg = dCt0 + 2 * SMAX * 256 + gl * 4096;             // [16 m][16 x][16 q] of this group
    float* sw = a.slabW + (int64_t)slab * a.slab_stride;
    float* sb = a.slabB + (int64_t)slab * a.slabB_stride;
    const bool g_on = g < G;                                    // F < 32: the last groups only keep the barriers
    const int mC = g * 16 + r;                                  // phase C row of this lane
    const bool mC_ok = g_on && mC < 2 * F;
    const int dhC = mC_ok && mC >= F ? 1 : 0, iC = mC_ok ? mC - dhC * F : 0;
    f32x4 accD[UPW][4];
#pragma unroll
    for (int u4 = 0; u4 < UPW; ++u4)
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) accD[u4][t4] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Phase D multiplies rows n = 2j + dw over ALL fields (rows with j <= i are computed and dropped; interleaved, the wanted rows
    // are the last 2(F-1-i): 2.4 row tiles per unit on average at F = 32 where n = dw*F + j needed 3): the operand address of row
    // tile t is then a lane constant (4 registers) + a scalar, shared by the wavefront's units - with the rows compressed to j > i
    // every MFMA carried ten integer instructions of (unit, tile) -> (dw, j) arithmetic and two selects.  A tile none of whose rows
    // has j > i is skipped (wave-uniform mask per unit).
    int offD[4];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
        const int n = t4 * 16 + r, dwn = n & 1, jn = n >> 1;
        offD[t4] = (n < 2 * F ? jn * Dp + dwn : 0) + 2 * kk;
    }
    unsigned mkU[UPW];
#pragma unroll
    for (int u4 = 0; u4 < UPW; ++u4) {
        const int m = g * 16 + wl * UPW + u4, dh = m >= F ? 1 : 0, i = m - dh * F;
        unsigned mk = 0;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const int any = (g_on && m < 2 * F ? 1 : 0) & (t4 * 16 < 2 * F ? 1 : 0) & (min(F - 1, 8 * t4 + 7) > i ? 1 : 0);
            mk |= (unsigned)any << t4;
        }
        mkU[u4] = __builtin_amdgcn_readfirstlane(mk);
    }
    float bsum = 0.f;                                          // bias partial of channel q0 + (gtid & 15), rows gtid >> 4 (group 0 only)
    // fetch of tile (b, xt) into buffer `buf`: dC pieces e4 = tid, tid + 1024 (S * 64 of them), embedding rows f = wave, wave + 16
    auto fetch_tile = [&](int b, int xt, int buf) {
        const int x0 = xt * 16;
        for (int e0 = wave * 64; e0 < S * 64; e0 += NTH) {
            const int e4 = e0 + lane, q4 = e4 & 3, x = (e4 >> 2) & 15, y = e4 >> 6;
            __builtin_amdgcn_global_load_lds(
                (const void __attribute__((address_space(1)))*)(a.dC + (((int64_t)b * S + y) * S + x0 + x) * PpT + q0 + 4 * q4),
                (void __attribute__((address_space(3)))*)(dCt0 + buf * SMAX * 256 + e0 * 4), 16, 0, 0);
        }
    };
    auto fetch_rows = [&](int b, int buf) {
        for (int f = wave; f < F; f += 16) {
            const float* row = row_ptr(a.in, a.idx, a.idxM, (int64_t)b * F + f, D, a.idxStride);
            if (lane < D)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(row + lane),
                                                 (void __attribute__((address_space(3)))*)(Es0 + buf * EsN + f * Dp), 4, 0, 0);
        }
    };
    int it = 0;                                                // tiles done: tile `it` lives in buffer it & 1, its rows in (it / RT) & 1
    if (slab < a.B) { fetch_rows(slab, 0); fetch_tile(slab, 0, 0); }
    for (int b = slab; b < a.B; b += nslab) {
        const float* Es = Es0 + ((it / RT) & 1) * EsN;
        for (int xt = 0; xt < RT; ++xt, ++it) {
            const int x0 = xt * 16;
            const float* dCt = dCt0 + (it & 1) * SMAX * 256;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile `it` (and of its rows) have landed
            __syncthreads();                                   // ... and everybody's; phase D of the previous tile is over
            {   // the next tile, and with the last column tile of an example the rows of the next example
                const bool last_x = xt + 1 == RT;
                const int nb = last_x ? b + nslab : b, nxt = last_x ? 0 : xt + 1;
                if (nb < a.B) {
                    if (last_x) fetch_rows(nb, ((it + 1) / RT) & 1);
                    fetch_tile(nb, nxt, (it + 1) & 1);
                }
            }
            if (g == 0) {                                      // db[q] += sum_{y,x} dC: thread (q = gtid & 15, part = gtid >> 4)
                const int q = gtid & 15, part = gtid >> 4;
                for (int e = part; e < S * 16; e += 4 * WPG) bsum += dCt[e * 16 + q];
            }
            // ---- phase C: dT rows mC (this group), wave's x columns 4*wl .. 4*wl+3 -----------------------------------
            if (g_on) {
                f32x4 acc[UPW];
#pragma unroll
                for (int xl = 0; xl < UPW; ++xl) acc[xl] = (f32x4){0.f, 0.f, 0.f, 0.f};
                for (int s4 = 0; s4 < S / 4; ++s4) {
                    const int y = 4 * s4 + kk;
                    const float av = mC_ok ? Es[iC * Dp + 2 * y + dhC] : 0.f;
#pragma unroll
                    for (int xl = 0; xl < UPW; ++xl) acc[xl] = mfma16(av, dCt[(y * 16 + UPW * wl + xl) * 16 + r], acc[xl]);
                }
#pragma unroll
                for (int xl = 0; xl < UPW; ++xl)
#pragma unroll
                    for (int j = 0; j < 4; ++j) dTg[((kk * 4 + j) * 16 + UPW * wl + xl) * 16 + r] = acc[xl][j];
            }
            __syncthreads();
            // ---- phase D: the wave's units: rows (dw,j) of ALL fields, K = x, cols q -----------------------------------
            if (g_on) {
                float bvv[UPW][4];
#pragma unroll
                for (int u4 = 0; u4 < UPW; ++u4)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) bvv[u4][s4] = dTg[((wl * UPW + u4) * 16 + 4 * s4 + kk) * 16 + r];
                const float* ex = Es + 2 * x0;
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    unsigned anyu = 0;
#pragma unroll
                    for (int u4 = 0; u4 < UPW; ++u4) anyu |= (mkU[u4] >> t4) & 1u;
                    if (!anyu) continue;
                    float av[4];
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) av[s4] = ex[offD[t4] + 8 * s4];
#pragma unroll
                    for (int u4 = 0; u4 < UPW; ++u4) {
                        if (!((mkU[u4] >> t4) & 1u)) continue;
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) accD[u4][t4] = mfma16(av[s4], bvv[u4][s4], accD[u4][t4]);
                    }
                }
            }
        }
    }
    // ---- write this workgroup's part of the slab ------------------------------------------------------------------
    if (g_on) {
#pragma unroll
        for (int u4 = 0; u4 < UPW; ++u4) {
            const int m = g * 16 + wl * UPW + u4;
            if (m >= 2 * F) continue;
            const int dh = m >= F ? 1 : 0, i = m - dh * F, base = i * (2 * F - i - 1) / 2;
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                if (!((mkU[u4] >> t4) & 1u)) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = t4 * 16 + kk * 4 + j, dwn = n & 1, jn = n >> 1;
                    if (n < 2 * F && jn > i) sw[((int64_t)(dh * 2 + dwn) * PpT + base + jn - i - 1) * PpT + q0 + r] = accD[u4][t4][j];
                }
            }
        }
    }
    // rows of padded pairs (p >= P) are never produced: they must read as zeros in the reduction
    for (int e = tid; e < 4 * (PpT - P) * 16 && gp == 0; e += NTH) {
        const int q = e & 15, rest = e >> 4, p = P + rest % (PpT - P), tap = rest / (PpT - P);
        sw[((int64_t)tap * PpT + p) * PpT + q0 + q] = 0.f;
    }
    __syncthreads();                                           // every group is done with the tiles
    float* red = dCt0;                                         // [4 * WPG parts][16 q]
    if (g == 0) red[(gtid >> 4) * 16 + (gtid & 15)] = bsum;
    __syncthreads();
    if (gp == 0 && tid < 16) {
        float v = 0.f;
#pragma unroll
        for (int part = 0; part < 4 * WPG; ++part) v += red[part * 16 + tid];
        sb[q0 + tid] = v;
    }
}

static int launch_conv0_fact_tile_wgrad(const WgradArgs& a, int nslab, hipStream_t st) {
    const int S = a.D / 2, G = (2 * a.F + 15) / 16;
    if (S > 32) return CFFM_ERR_UNSUPPORTED;
    const char* ver = getenv("CFFM_TILE_WGRAD");                // debug: 1 = round-2 kernel, 2 = two groups per workgroup
    if (G <= 4 && !(ver && ver[0] == '1')) {
        if (!(ver && ver[0] == '2')) {
            // all four groups in one workgroup of 16 wavefronts: the dC tile is staged ONCE per (example, column tile); 4 units per
            // wavefront, 64 accumulation registers - 122 registers and nothing spilled since phase D stopped computing an operand
            // address per MFMA (before: 31-49 spilled, slower than two groups)
            const size_t lds = (size_t)(2 * ((a.F * (a.D + 1) + 3) / 4 * 4) + 2 * 32 * 256 + 4 * 16 * 256) * 4 + 16;
            int rc = set_lds(conv0_fact_tile_wgrad_all_kernel<32, 4>, lds);
            if (rc) return rc;
            const int64_t grid = (int64_t)(a.Pp / 16) * nslab;
            hipLaunchKernelGGL((conv0_fact_tile_wgrad_all_kernel<32, 4>), dim3((unsigned)grid), dim3(1024), lds, st, a, nslab);
            CFFM_CHECK_LAUNCH();
            return 0;
        }
        // two groups per workgroup: the tile is staged twice
        const size_t lds = (size_t)(2 * ((a.F * (a.D + 1) + 3) / 4 * 4) + 2 * 32 * 256 + 2 * 16 * 256) * 4 + 16;
        int rc = set_lds(conv0_fact_tile_wgrad_all_kernel<32, 2>, lds);
        if (rc) return rc;
        const int64_t grid = (int64_t)(a.Pp / 16) * 2 * nslab;
        hipLaunchKernelGGL((conv0_fact_tile_wgrad_all_kernel<32, 2>), dim3((unsigned)grid), dim3(1024), lds, st, a, nslab);
        CFFM_CHECK_LAUNCH();
        return 0;
    }
    const size_t lds = (size_t)((a.F * (a.D + 1) + 3) / 4 * 4 + 32 * 256 + 16 * 256) * 4 + 16;
    constexpr int NW = 8;
    int rc = set_lds(conv0_fact_tile_wgrad_kernel<32, NW>, lds);
    if (rc) return rc;
    const int64_t grid = (int64_t)(a.Pp / 16) * G * nslab;
    hipLaunchKernelGGL((conv0_fact_tile_wgrad_kernel<32, NW>), dim3((unsigned)grid), dim3(64 * NW), lds, st, a, nslab);
    CFFM_CHECK_LAUNCH();
    return 0;
}

// Input gradient of layer 0 (dEo) for wide filters in factorised form.  With T and dT as above,
//   dEi[(dh,i)][y] = sum_{x,q} dC[y][x][q] * T[dh][i][x][q]                              rows y, K = (x,q), cols (dh,i)
//   dEj[(dw,j)][x] = sum_{dh,i<j,q} W[dh,dw,(i,j),q] * dT[dh][i][x][q]                   rows (dw,j), K = q, cols x
//   dEo[f][h]      = dEi[(h&1,f)][h>>1] + dEj[(h&1,f)][h>>1] + dt1[h]*R_f + Q_f           (s0 pool gradient in closed form)
// One workgroup per example walks (group g of 16 (dh,i) rows) x (column tile) x (channel tile); per step it stages the
// dC tile, recomputes the 16 T planes and the 16 dT planes of the group in LDS and feeds two accumulators that live in
// registers across the walk: dEi of the group (K split over the wavefronts, summed at the end of the group) and dEj
// (each wavefront sums its own four units, summed at the very end).  ~325 MFLOP per example at F32 D64 against 2,015.
template <int SMAX, int NW>
__global__ __launch_bounds__(64 * NW) void conv0_fact_tile_dgrad_kernel(DgradArgs a) {
    constexpr int NTH = 64 * NW, UPW = 16 / NW, KPW = 64 / NW;   // units / phase-C columns per wavefront, phase-B k-steps per wavefront
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int F = a.F, D = a.D, S = D / 2, Dp = D + 1, RT = S / 16, YT = S / 16, PpT = a.Pp, QT = PpT / 16, G = (2 * F + 15) / 16;
    // LDS plan (74.6 KB at F32 D64: TWO workgroups per CU).  Phase B reads Tg and dCt with the LANE walking the plane /
    // row index (m = r, y = yt*16 + r): with the natural pitch of 256 floats all sixteen lanes of a k group hit one bank
    // (rocprofv3: SQ_LDS_BANK_CONFLICT = 82 % of SQ_LDS_IDX_ACTIVE, profiles/r02_syn1m_pmc.md).  Pitch 258 makes the Tg
    // read conflict-free (bank = 2r + kk); dCt is staged in 16-byte pieces, so its pitch stays a multiple of 4: 260
    // (bank = 4r + kk, 2-way).  The buffers of the epilogue (cross-wave partial sums, dEi / dEj) reuse Tg / dTg.
    constexpr int TGP = 258, DCP = 260;
    float* Es = reinterpret_cast<float*>(smem);                // [F][Dp]
    float* dCt = Es + (F * Dp + 3) / 4 * 4;                    // [S][DCP]: (x, q) at x*16 + q
    float* Tg = dCt + SMAX * DCP;                               // [16 m][TGP]
    float* dTg = Tg + 16 * TGP;                                 // [16 m][16 x][16 q]  (16 * 258 floats keep it 16-byte aligned)
    static_assert(NW * 4 * 256 <= 16 * TGP, "the cross-wave partial sums reuse Tg");
    float* part = Tg;                                           // [NW waves][4 tiles][64 lanes][4]   cross-wave sums (epilogue)
    float* dEi = dTg;                                           // [G*16][SMAX]    (n = dh*F + i, y)            (epilogue)
    float* dEj = dEi + 64 * SMAX;                               // [64][SMAX]      (n = dw*F + j, x)            (epilogue)
    float* rs = dTg + 4096;                                     // [F] row sums, [F] dots
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kk = lane >> 4;
    const int b = blockIdx.x;
    stage_example_rows(Es, a.Cprev, a.idx, a.idxM, b, F, D, Dp, tid, NTH, a.idxStride);
    f32x4 accE[2][4];                                          // [column tile][row tile of (dw,j)]
#pragma unroll
    for (int xt = 0; xt < 2; ++xt)
#pragma unroll
        for (int t = 0; t < 4; ++t) accE[xt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 accB[4][2];                                          // [row group g][y tile] x (16 rows of the group)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int yt = 0; yt < 2; ++yt) accB[g][yt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* dCb = a.dC + (int64_t)b * S * S * PpT;
    // The dC tile of the NEXT (column tile, channel tile) is fetched into registers while the current one is worked on (two
    // wavefronts per SIMD do not hide a global load that is issued and consumed in the same step: 4 of the 64 ms).  Measured and
    // not kept: the filter slices of phase E requested two phases ahead as well (64 more registers: 256 with spills and
    // scratch-resident operand arrays, 64 -> 104 ms).
    constexpr int NX = SMAX * 64 / NTH;                        // 16-byte pieces of a tile per thread
    static_assert(NX == 8, "the prefetch registers are eight named float4s (an array ends up in scratch memory here)");
    float4 nx0, nx1, nx2, nx3, nx4, nx5, nx6, nx7;
#define CFFM_TILE_FETCH(X0N, Q0N)                                                                                              \
    do {                                                                                                                      \
        const float* tb_ = dCb + (int64_t)(X0N) * PpT + (Q0N) + 4 * (tid & 3) + (int64_t)((tid >> 2) & 15) * PpT;              \
        const int64_t ys_ = (int64_t)S * PpT;                  /* e4 = tid + u * 256: y = (tid >> 6) + 4u */                   \
        const int y_ = tid >> 6;                                                                                              \
        nx0 = *reinterpret_cast<const float4*>(tb_ + min(y_, S - 1) * ys_);                                                   \
        nx1 = *reinterpret_cast<const float4*>(tb_ + min(y_ + 4, S - 1) * ys_);                                               \
        nx2 = *reinterpret_cast<const float4*>(tb_ + min(y_ + 8, S - 1) * ys_);                                               \
        nx3 = *reinterpret_cast<const float4*>(tb_ + min(y_ + 12, S - 1) * ys_);                                              \
        nx4 = *reinterpret_cast<const float4*>(tb_ + min(y_ + 16, S - 1) * ys_);                                              \
        nx5 = *reinterpret_cast<const float4*>(tb_ + min(y_ + 20, S - 1) * ys_);                                              \
        nx6 = *reinterpret_cast<const float4*>(tb_ + min(y_ + 24, S - 1) * ys_);                                              \
        nx7 = *reinterpret_cast<const float4*>(tb_ + min(y_ + 28, S - 1) * ys_);                                              \
    } while (0)
    static_assert(NTH == 256, "y = (tid >> 6) + 4u above");
    CFFM_TILE_FETCH(0, 0);
#pragma unroll
    for (int xt = 0; xt < 2; ++xt) {
        if (xt >= RT) continue;
        const int x0 = xt * 16;
        for (int qt = 0; qt < QT; ++qt) {
            const int q0 = qt * 16;
            __syncthreads();                                   // dCt / Tg / dTg of the previous step consumed
            {                                                  // the dC tile is staged ONCE for the four row groups
                float* tl_ = dCt + ((tid >> 2) & 15) * 16 + 4 * (tid & 3);
                const int y_ = tid >> 6;
                if (y_ < S) *reinterpret_cast<float4*>(tl_ + y_ * DCP) = nx0;
                if (y_ + 4 < S) *reinterpret_cast<float4*>(tl_ + (y_ + 4) * DCP) = nx1;
                if (y_ + 8 < S) *reinterpret_cast<float4*>(tl_ + (y_ + 8) * DCP) = nx2;
                if (y_ + 12 < S) *reinterpret_cast<float4*>(tl_ + (y_ + 12) * DCP) = nx3;
                if (y_ + 16 < S) *reinterpret_cast<float4*>(tl_ + (y_ + 16) * DCP) = nx4;
                if (y_ + 20 < S) *reinterpret_cast<float4*>(tl_ + (y_ + 20) * DCP) = nx5;
                if (y_ + 24 < S) *reinterpret_cast<float4*>(tl_ + (y_ + 24) * DCP) = nx6;
                if (y_ + 28 < S) *reinterpret_cast<float4*>(tl_ + (y_ + 28) * DCP) = nx7;
            }
            {   // the next tile (clamped to the last one: a harmless re-read at the very end)
                const bool nq = qt + 1 < QT, nxt = xt + 1 < RT;
                CFFM_TILE_FETCH(nq ? x0 : (nxt ? x0 + 16 : x0), nq ? q0 + 16 : (nxt ? 0 : q0));
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g >= G) continue;
                const int mC = g * 16 + r;                      // phase C row of this lane
                const bool mC_ok = mC < 2 * F;
                const int dhC = mC_ok && mC >= F ? 1 : 0, iC = mC_ok ? mC - dhC * F : 0;
                if (g > 0) __syncthreads();                    // Tg / dTg of the previous group consumed
                // ---- A: T planes of this wave's four units ---------------------------------------------------------
#pragma unroll
                for (int u4 = 0; u4 < UPW; ++u4) {
                    const int ml = wave * UPW + u4, m = g * 16 + ml;
                    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (m < 2 * F) {
                        const int dh = m >= F ? 1 : 0, i = m - dh * F;
                        const int nj = F - 1 - i, K = 2 * nj, base = i * (2 * F - i - 1) / 2;
                        float bw[C0T_MAXKS], av[C0T_MAXKS];
#pragma unroll
                        for (int ks = 0; ks < C0T_MAXKS; ++ks) {
                            const int k = 4 * ks + kk;
                            const bool ok = k < K;
                            const int dw = (ok && k >= nj) ? 1 : 0, jj = ok ? k - dw * nj : 0;
                            // (i = F-1 has no pairs: K = 0, base = P - keep the discarded reads inside W and Es)
                            const float wv = a.W[((int64_t)(dh * 2 + dw) * PpT + min(base + jj, a.P - 1)) * PpT + q0 + r];
                            const float ev = Es[min(i + 1 + jj, F - 1) * Dp + 2 * (x0 + r) + dw];
                            bw[ks] = ok ? wv : 0.f;
                            av[ks] = ok ? ev : 0.f;
                        }
                        // two straight chains instead of a test per MFMA (a branch and a full wait in front of each):
                        // 64.2 ms against 68.7 with the tests and 67.9 with four chain lengths (F32 D64 B8192)
                        if (K > 2 * C0T_MAXKS) {
#pragma unroll
                            for (int ks = 0; ks < C0T_MAXKS; ++ks) acc = mfma16(av[ks], bw[ks], acc);
                        } else {
#pragma unroll
                            for (int ks = 0; ks < C0T_MAXKS / 2; ++ks) acc = mfma16(av[ks], bw[ks], acc);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) Tg[ml * TGP + (kk * 4 + j) * 16 + r] = acc[j];
                }
                if (g == 0) __syncthreads();                   // dCt staged
                // ---- C: dT planes of the group, wave's columns 4*wave .. 4*wave+3 --------------------------------------
                {
                    f32x4 acc[UPW];
#pragma unroll
                    for (int xl = 0; xl < UPW; ++xl) acc[xl] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    for (int s4 = 0; s4 < S / 4; ++s4) {
                        const int y = 4 * s4 + kk;
                        const float av = mC_ok ? Es[iC * Dp + 2 * y + dhC] : 0.f;
#pragma unroll
                        for (int xl = 0; xl < UPW; ++xl) acc[xl] = mfma16(av, dCt[y * DCP + (UPW * wave + xl) * 16 + r], acc[xl]);
                    }
#pragma unroll
                    for (int xl = 0; xl < UPW; ++xl)
#pragma unroll
                        for (int j = 0; j < 4; ++j) dTg[((kk * 4 + j) * 16 + UPW * wave + xl) * 16 + r] = acc[xl][j];
                }
                __syncthreads();                               // Tg and dTg written
                // ---- B: dEi of the group: rows y, K = (x,q) of this tile (64 k-steps, 16 per wave), cols m -----------------
                for (int ks = 0; ks < KPW; ++ks) {
                    const int kf = 4 * (wave * KPW + ks) + kk, x = kf >> 4, q = kf & 15;
                    const float bv = Tg[r * TGP + x * 16 + q];
#pragma unroll
                    for (int yt = 0; yt < 2; ++yt)
                        if (yt < YT) accB[g][yt] = mfma16(dCt[(yt * 16 + r) * DCP + x * 16 + q], bv, accB[g][yt]);
                }
                // ---- E: dEj rows (dw,j), K = q, cols x: this wave's four units ------------------------------------------
#pragma unroll
                for (int u4 = 0; u4 < UPW; ++u4) {
                    const int ml = wave * UPW + u4, m = g * 16 + ml;
                    if (m >= 2 * F) continue;
                    const int dh = m >= F ? 1 : 0, i = m - dh * F, base = i * (2 * F - i - 1) / 2;
                    const float4 bv = *reinterpret_cast<const float4*>(dTg + (ml * 16 + r) * 16 + 4 * kk);   // [k = q = 4kk+t][n = x = r]
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int n = t * 16 + r, dw = n >= F ? 1 : 0, j = n - dw * F;
                        // the tile is empty when none of its rows has j > i (wave-uniform test on the tile's last rows)
                        const int jmax_lo = min(F - 1, t * 16 + 15), jmax_hi = t * 16 + 15 - F;
                        const bool any = (t * 16 < F && jmax_lo > i) || (t * 16 + 15 >= F && jmax_hi > i && t * 16 < 2 * F);
                        if (!any) continue;
                        const bool ok = n < 2 * F && j > i;
                        const int jc = ok ? j - i - 1 : 0, dwc = ok ? dw : 0;            // clamped row, zeroed by the select
                        float4 wv = *reinterpret_cast<const float4*>(a.W + ((int64_t)(dh * 2 + dwc) * PpT + base + jc) * PpT + q0 + 4 * kk);
                        if (!ok) wv = make_float4(0.f, 0.f, 0.f, 0.f);
                        accE[xt][t] = mfma16(wv.x, bv.x, accE[xt][t]);
                        accE[xt][t] = mfma16(wv.y, bv.y, accE[xt][t]);
                        accE[xt][t] = mfma16(wv.z, bv.z, accE[xt][t]);
                        accE[xt][t] = mfma16(wv.w, bv.w, accE[xt][t]);
                    }
                }
            }
        }
    }
    // ---- dEi: sum the four wavefronts' K slices, group by group -------------------------------------------------------
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g >= G) continue;
        __syncthreads();
#pragma unroll
        for (int yt = 0; yt < 2; ++yt)
            *reinterpret_cast<f32x4*>(part + ((wave * 4 + yt) * 64 + lane) * 4) = accB[g][yt];
        __syncthreads();
        for (int e = tid; e < YT * 256; e += NTH) {              // e = (yt, lane, j)
            const int j = e & 3, ln = (e >> 2) & 63, yt = e >> 8;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += part[((w * 4 + yt) * 64 + ln) * 4 + j];
            const int y = yt * 16 + (ln >> 4) * 4 + j, mm = g * 16 + (ln & 15);          // D layout: row y, col m
            dEi[mm * SMAX + y] = v;
        }
    }
    // ---- dEj: sum the four wavefronts' unit subsets -----------------------------------------------------------------
#pragma unroll
    for (int xt = 0; xt < 2; ++xt) {
        if (xt >= RT) continue;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(part + ((wave * 4 + t) * 64 + lane) * 4) = accE[xt][t];
        __syncthreads();
        for (int e = tid; e < 4 * 256; e += NTH) {              // e = (t, lane, j)
            const int j = e & 3, ln = (e >> 2) & 63, t = e >> 8;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += part[((w * 4 + t) * 64 + ln) * 4 + j];
            const int n = t * 16 + (ln >> 4) * 4 + j, x = xt * 16 + (ln & 15);           // D layout: row (dw,j), col x
            dEj[n * SMAX + x] = v;
        }
    }
    if (tid < 2 * F) {                                           // row sums and <ds0, E[f]> for the closed-form s0 terms
        const int f = tid % F;
        float sacc = 0.f;
        if (tid < F) { for (int h = 0; h < D; ++h) sacc += Es[f * Dp + h]; }
        else { for (int h = 0; h < D; ++h) sacc += Es[f * Dp + h] * a.dt1[(int64_t)b * a.t1w + h]; }
        rs[tid] = sacc;
    }
    __syncthreads();
    for (int e = tid; e < F * D; e += NTH) {
        const int f = e / D, h = e - f * D, lo = h & 1, hh = h >> 1;
        float R = 0.f, Q = 0.f;
        for (int j = f + 1; j < F; ++j) R += rs[j];
        for (int i = 0; i < f; ++i) Q += rs[F + i];
        a.dprev[(int64_t)b * F * D + e] = (dEi[(lo * F + f) * SMAX + hh] + dEj[(lo * F + f) * SMAX + hh])
                                          + a.dt1[(int64_t)b * a.t1w + h] * R + Q;
    }
}

#undef CFFM_TILE_FETCH

// ---- the same input gradient with the filter read as PRE-PACKED MFMA fragments ------------------------------------------
// rocprofv3 + the ISA of the kernel above: 19.7 K instructions for 784 MFMAs - per MFMA of phases A and E a dozen integer
// instructions of (pair, tap) address arithmetic, two selects and a 64-bit multiply-add, unrolled over 4 row groups x 2 column
// tiles (an instruction stream larger than the instruction cache); MFMA pipe busy 30 %.  Here the layer-0 filter is laid out
// ONCE per backward pass (pack_w0_tile_kernel, 16 MB written, a few us) exactly as the two phases consume it:
//   WA[m = (dh,i)][qt][ks = dw*8 + c][lane = (kk,r)]      = W[dh,dw,(i,j),q0+r],  j = 4c + kk  (0 where j <= i or j >= F)
//   WE[m = (dh,i)][qt][t][lane = (kk,r)] (16 bytes)       = W[dh,dw,(i,j),q0+4kk .. +3],  row n = 16t + r = 2j + dw   (0 likewise)
// so a fragment is ONE load at (wave-uniform base) + lane with no select, k of phase A runs over ALL fields (the E operand of
// a k-step no longer depends on the unit: one LDS read pair feeds the wave's UPW units) and the k-steps whose fields all lie
// at or below i are skipped (8.5 MFMAs per unit on average instead of 12).
__global__ __launch_bounds__(256) void pack_w0_tile_kernel(const float* __restrict__ W, float* __restrict__ WA, float4* __restrict__ WE,
                                                          int F, int Pp) {
    const int QT = Pp / 16, m = blockIdx.x / QT, qt = blockIdx.x - m * QT, q0 = qt * 16;
    const int dh = m >= F ? 1 : 0, i = m - dh * F, base = i * (2 * F - i - 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, kk = lane >> 4;
    float* wa = WA + (int64_t)blockIdx.x * 1024;
    for (int e = tid; e < 1024; e += 256) {
        const int ks = e >> 6, dw = ks >> 3, j = 4 * (ks & 7) + kk;
        const bool ok = j < F && j > i;
        wa[e] = ok ? W[((int64_t)(dh * 2 + dw) * Pp + base + (j - i - 1)) * Pp + q0 + r] : 0.f;
    }
    {
        const int t = tid >> 6, n = t * 16 + r, dw = n & 1, j = n >> 1;     // rows interleaved, n = 2j + dw: the rows with j > i are the
        const bool ok = n < 2 * F && j > i;                                  // LAST 2(F-1-i) of them - whole leading tiles are empty
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *reinterpret_cast<const float4*>(W + ((int64_t)(dh * 2 + dw) * Pp + base + (j - i - 1)) * Pp + q0 + 4 * kk);
        WE[(int64_t)blockIdx.x * 256 + tid] = v;
    }
}

template <int SMAX, int NW>
__global__ __launch_bounds__(64 * NW, 2) void conv0_fact_tile_dgrad2_kernel(DgradArgs a, const float* __restrict__ WA,
                                                                           const float4* __restrict__ WE) {
    constexpr int NTH = 64 * NW, UPW = 16 / NW, KPW = 64 / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int F = a.F, D = a.D, S = D / 2, Dp = D + 1, RT = S / 16, YT = S / 16, PpT = a.Pp, QT = PpT / 16, G = (2 * F + 15) / 16;
    const int CE = (F + 3) / 4, Fp = 4 * CE, NTE = (2 * F + 15) / 16;
    constexpr int TGP = 258, DCP = 260;
    float* Es = reinterpret_cast<float*>(smem);                // [Fp + 1][Dp], rows F .. Fp zero (Fp: the row of lanes without a unit)
    float* dCt = Es + ((Fp + 1) * Dp + 3) / 4 * 4;             // [S][DCP]: (x, q) at x*16 + q
    float* Tg = dCt + SMAX * DCP;                               // [16 m][TGP]
    float* dTg = Tg + 16 * TGP;                                 // [16 m][16 x][16 q]
    static_assert(NW * 4 * 256 <= 16 * TGP + 4096, "the cross-wave partial sums reuse Tg (and dTg)");
    static_assert(2 * 64 * SMAX + 128 <= SMAX * DCP, "the epilogue outputs reuse dCt");
    float* part = Tg;                                           // [NW waves][4 tiles][64 lanes][4]   cross-wave sums (epilogue)
    float* dEi = dCt;                                           // [G*16][SMAX]    (n = dh*F + i, y)            (epilogue)
    float* dEj = dEi + 64 * SMAX;                               // [64][SMAX]      (n = dw*F + j, x)            (epilogue)
    float* rs = dEj + 64 * SMAX;                                // [F] row sums, [F] dots
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, kk = lane >> 4;
    const int b = blockIdx.x;
    const unsigned ulane = lane;
    stage_example_rows(Es, a.Cprev, a.idx, a.idxM, b, F, D, Dp, tid, NTH, a.idxStride);
    for (int e = tid; e < (Fp + 1 - F) * Dp; e += NTH) Es[F * Dp + e] = 0.f;
    f32x4 accE[2][4];                                          // [column tile][row tile of (dw,j)]
#pragma unroll
    for (int xt = 0; xt < 2; ++xt)
#pragma unroll
        for (int t = 0; t < 4; ++t) accE[xt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 accB[4][2];                                          // [row group g][y tile] x (16 rows of the group)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int yt = 0; yt < 2; ++yt) accB[g][yt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* dCb = a.dC + (int64_t)b * S * S * PpT;
    // The dC tile goes HBM -> LDS directly (global_load_lds_dwordx4: the 64 pieces of row y = (x, q4) land as 1 KB at
    // dCt + y*DCP): no staging registers (eight float4s per lane, four of them spilled next to 96 accumulation registers).
    // The tile of step n+1 is requested in the LAST row group of step n, once phase B has read dCt for the last time, and
    // lands behind that group's phase E.  Rows beyond S are never read (clamped source, S = 16).
    auto fetch_tile = [&](int x0n, int q0n) {
        const float* src = dCb + (int64_t)(x0n + (lane >> 2)) * PpT + q0n + 4 * (lane & 3);
#pragma unroll
        for (int k = 0; k < SMAX / NW; ++k) {
            const int y = wave + NW * k;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (int64_t)min(y, S - 1) * S * PpT),
                                             (void __attribute__((address_space(3)))*)(dCt + y * DCP), 16, 0, 0);
        }
    };
    fetch_tile(0, 0);
    // xt and g are RUN-TIME loops (one copy of phases A and C in the instruction stream); the accumulators that live across
    // the walk are picked by a switch around phases B and E only
#pragma clang loop unroll(disable)
    for (int xt = 0; xt < RT; ++xt) {
        const int x0 = xt * 16;
        const float* eb = Es + kk * Dp + 2 * (x0 + r);        // phase A: E operand of k-step c, tap dw: eb[4*c*Dp + dw]
        for (int qt = 0; qt < QT; ++qt) {
            const int q0 = qt * 16;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's rows of the tile have landed
            __syncthreads();                                   // ... and everybody's; Tg / dTg of the previous step consumed
#pragma clang loop unroll(disable)
            for (int g = 0; g < G; ++g) {
                const int mC = g * 16 + r;                      // phase C row of this lane (a row beyond 2F reads the zero row of Es)
                const int dhC = (mC < 2 * F && mC >= F) ? 1 : 0, iC = mC < 2 * F ? mC - dhC * F : Fp;
                // the wave's UPW units of this group (wave-uniform): m = (dh, i), first useful k-step, fragment bases
                int mU[UPW], iU[UPW], okU[UPW];
                int cs = CE;
#pragma clang loop unroll(full)
                for (int u = 0; u < UPW; ++u) {
                    const int m = g * 16 + wave * UPW + u;
                    okU[u] = m < 2 * F ? 1 : 0;
                    mU[u] = min(m, 2 * F - 1);
                    iU[u] = mU[u] - (mU[u] >= F ? F : 0);
                    cs = min(cs, (iU[u] + 1) >> 2);
                }
                cs = __builtin_amdgcn_readfirstlane(cs);       // (hipcc takes the minimum in a vector register: keep the fragment
                                                               //  addresses below scalar base + lane)
                // ---- A: T planes of this wave's units: rows x, K = (dw, j) over all fields from k-step cs on, cols q; the first
                //         DEPTH k-steps are requested before the barrier (they depend on (unit, qt) only), then k-step d + DEPTH
                //         is requested when k-step d is multiplied
                f32x4 accA[UPW];
                constexpr int DEPTH = 3;
                float wv0[DEPTH][UPW], wv1[DEPTH][UPW], ev0[DEPTH], ev1[DEPTH];
                const float* wa[UPW];
#pragma clang loop unroll(full)
                for (int u = 0; u < UPW; ++u) {
                    accA[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    wa[u] = WA + __builtin_amdgcn_readfirstlane((mU[u] * QT + qt) * 1024);   // wave-uniform: fragment = (scalar base)[lane]
                }
#define CFFM_A_REQ(SLOT, C)                                                                                                   \
    if ((C) < CE) {                                                                                                           \
        _Pragma("clang loop unroll(full)") for (int u = 0; u < UPW; ++u) {                                                    \
            wv0[SLOT][u] = (wa[u] + (C) * 64)[ulane]; wv1[SLOT][u] = (wa[u] + ((C) + 8) * 64)[ulane];                                             \
        }                                                                                                                     \
        ev0[SLOT] = eb[4 * (C) * Dp]; ev1[SLOT] = eb[4 * (C) * Dp + 1];                                                       \
    }
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) { CFFM_A_REQ(d, cs + d) }
                if (g > 0) __syncthreads();                    // Tg / dTg of the previous group consumed
                for (int c = cs; c < CE; c += DEPTH) {
#pragma unroll
                    for (int d = 0; d < DEPTH; ++d) {
                        if (c + d < CE) {
#pragma clang loop unroll(full)
                            for (int u = 0; u < UPW; ++u) {
                                accA[u] = mfma16(ev0[d], wv0[d][u], accA[u]);
                                accA[u] = mfma16(ev1[d], wv1[d][u], accA[u]);
                            }
                            CFFM_A_REQ(d, c + d + DEPTH)
                        }
                    }
                }
#undef CFFM_A_REQ
#pragma clang loop unroll(full)
                for (int u = 0; u < UPW; ++u) {
                    const int ml = wave * UPW + u;
                    if (!okU[u]) accA[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) Tg[ml * TGP + (kk * 4 + j) * 16 + r] = accA[u][j];
                }
                // The filter fragments of phase E: the first half of the wave's units is requested here (phases C and B hide the
                // latency), the second half after phase C (all of them here: 64 live registers across phase C, spills; before
                // phase A: more spills).
                float4 we[UPW][4];
                unsigned emask[UPW];
#define CFFM_E_REQ(U0, U1)                                                                                                    \
    _Pragma("clang loop unroll(full)") for (int u = (U0); u < (U1); ++u) {                                                    \
        const float4* wep = WE + __builtin_amdgcn_readfirstlane((mU[u] * QT + qt) * 256);   /* wave-uniform */                 \
        unsigned mk = 0;                                                                                                      \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                                       \
            /* tile t holds rows n = 2j + dw = 16t .. 16t+15, i.e. fields 8t .. 8t+7: empty when none of them is > i */       \
            const int any = okU[u] & (t < NTE ? 1 : 0) & (min(F - 1, 8 * t + 7) > iU[u] ? 1 : 0);                             \
            mk |= (unsigned)any << t;                                                                                         \
        }                                                                                                                     \
        emask[u] = mk;                                                                                                        \
        /* UNCONDITIONAL loads (an empty tile re-reads the unit's last tile: an L1 hit): behind conditional loads hipcc     \
           waits with vmcnt(0) in phase E - in the last row group that is a wait for the NEXT dC tile */                        \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) we[u][t] = (wep + (((mk >> t) & 1u) ? t : 3) * 64)[ulane];               \
    }
                CFFM_E_REQ(0, (UPW + 1) / 2)
                // ---- C: dT planes of the group, wave's columns UPW*wave .. UPW*wave + UPW-1; K = y in halves of 16: the 4 + 4*UPW
                //         operands of a half are requested together, then its 4*UPW MFMAs run ---------------------------------------
                {
                    f32x4 acc[UPW];
#pragma clang loop unroll(full)
                    for (int xl = 0; xl < UPW; ++xl) acc[xl] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    const float* ep = Es + iC * Dp + 2 * kk + dhC;
                    const float* dp = dCt + kk * DCP + UPW * wave * 16 + r;
                    // (hipcc moves every LDS read back in front of its MFMA - one exposed LDS round trip per two MFMAs - unless a
                    //  scheduling barrier separates the requests from the arithmetic)
#pragma clang loop unroll(disable)
                    for (int h = 0; h < YT; ++h) {
                        float av0[4], bv0[4][UPW];
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            av0[s] = ep[32 * h + 8 * s];
#pragma clang loop unroll(full)
                            for (int xl = 0; xl < UPW; ++xl) bv0[s][xl] = dp[(16 * h + 4 * s) * DCP + xl * 16];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma clang loop unroll(full)
                            for (int xl = 0; xl < UPW; ++xl) acc[xl] = mfma16(av0[s], bv0[s][xl], acc[xl]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma clang loop unroll(full)
                    for (int xl = 0; xl < UPW; ++xl)
#pragma unroll
                        for (int j = 0; j < 4; ++j) dTg[((kk * 4 + j) * 16 + UPW * wave + xl) * 16 + r] = acc[xl][j];
                }
                CFFM_E_REQ((UPW + 1) / 2, UPW)
#undef CFFM_E_REQ
                __syncthreads();                               // Tg and dTg written
                // ---- B: dEi of the group: rows y, K = (x,q) of this tile (64 k-steps, KPW per wave), cols m; operands of four
                //         k-steps requested together (k index kf = 4*(wave*KPW + ks) + kk = x*16 + q) -----------------------------
#define CFFM_PHASE_B_LOAD(K4, BV, A0, A1)                                                                                     \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                           \
        BV[s] = tp[4 * ((K4) + s)]; A0[s] = dp[4 * ((K4) + s)];                                                               \
        if (YT > 1) A1[s] = dp[16 * DCP + 4 * ((K4) + s)];                                                                    \
    }                                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);
#define CFFM_PHASE_B_MMA(GG, BV, A0, A1)                                                                                      \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                           \
        accB[GG][0] = mfma16(A0[s], BV[s], accB[GG][0]);                                                                      \
        if (YT > 1) accB[GG][1] = mfma16(A1[s], BV[s], accB[GG][1]);                                                          \
    }                                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);
#define CFFM_PHASE_B(GG)                                                                                                      \
    {   /* (two operand sets - chunk n+1 requested before chunk n is multiplied - spill 59 registers at NW = 4) */            \
        const float* tp = Tg + r * TGP + 4 * wave * KPW + kk;                                                                 \
        const float* dp = dCt + r * DCP + 4 * wave * KPW + kk;                                                                \
        _Pragma("unroll") for (int k4 = 0; k4 < KPW; k4 += 4) {                                                               \
            float bA[4], aA0[4], aA1[4];                                                                                      \
            CFFM_PHASE_B_LOAD(k4, bA, aA0, aA1)                                                                               \
            CFFM_PHASE_B_MMA(GG, bA, aA0, aA1)                                                                                \
        }                                                                                                                     \
    }
                switch (g) {
                    case 0: CFFM_PHASE_B(0) break;
                    case 1: CFFM_PHASE_B(1) break;
                    case 2: CFFM_PHASE_B(2) break;
                    default: CFFM_PHASE_B(3) break;
                }
#undef CFFM_PHASE_B
#undef CFFM_PHASE_B_LOAD
#undef CFFM_PHASE_B_MMA
                // ---- E: dEj rows (dw,j), K = q, cols x: this wave's units ------------------------------------------------------
#define CFFM_PHASE_E(XT)                                                                                                      \
    _Pragma("clang loop unroll(full)") for (int u = 0; u < UPW; ++u) {                                                        \
        if (emask[u] == 0) continue;                                                                                          \
        const float4 bv = bvE[u];                                                                                             \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                                       \
            if (!((emask[u] >> t) & 1u)) continue;                                                                            \
            const float4 wv = we[u][t];                                                                                       \
            accE[XT][t] = mfma16(wv.x, bv.x, accE[XT][t]);                                                                    \
            accE[XT][t] = mfma16(wv.y, bv.y, accE[XT][t]);                                                                    \
            accE[XT][t] = mfma16(wv.z, bv.z, accE[XT][t]);                                                                    \
            accE[XT][t] = mfma16(wv.w, bv.w, accE[XT][t]);                                                                    \
        }                                                                                                                     \
    }
                // the dT operands of the phase are read BEFORE the next tile is requested: hipcc puts a full vmcnt(0) in front of
                // any LDS read that follows a global_load_lds (it cannot tell dTg from dCt)
                float4 bvE[UPW];
#pragma clang loop unroll(full)
                for (int u = 0; u < UPW; ++u)
                    bvE[u] = *reinterpret_cast<const float4*>(dTg + ((wave * UPW + u) * 16 + r) * 16 + 4 * kk);   // [k = q = 4kk+c][n = x = r]
                if (g == G - 1) {                              // (phase E twice in the source: its waits for the fragments must
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __syncthreads();                           //  not see the tile request as possibly older than they are)
                    const bool nq = qt + 1 < QT, nxt = xt + 1 < RT;   // clamped to the last tile: a harmless re-read at the very end
                    fetch_tile(nq ? x0 : (nxt ? x0 + 16 : x0), nq ? q0 + 16 : (nxt ? 0 : q0));
                    if (xt == 0) { CFFM_PHASE_E(0) } else { CFFM_PHASE_E(1) }
                } else {
                    if (xt == 0) { CFFM_PHASE_E(0) } else { CFFM_PHASE_E(1) }
                }
#undef CFFM_PHASE_E
            }
        }
    }
    // ---- dEi: sum the wavefronts' K slices, group by group ------------------------------------------------------------
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g >= G) continue;
        __syncthreads();
#pragma unroll
        for (int yt = 0; yt < 2; ++yt)
            *reinterpret_cast<f32x4*>(part + ((wave * 4 + yt) * 64 + lane) * 4) = accB[g][yt];
        __syncthreads();
        for (int e = tid; e < YT * 256; e += NTH) {              // e = (yt, lane, j)
            const int j = e & 3, ln = (e >> 2) & 63, yt = e >> 8;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += part[((w * 4 + yt) * 64 + ln) * 4 + j];
            const int y = yt * 16 + (ln >> 4) * 4 + j, mm = g * 16 + (ln & 15);          // D layout: row y, col m
            dEi[mm * SMAX + y] = v;
        }
    }
    // ---- dEj: sum the wavefronts' unit subsets -----------------------------------------------------------------------
#pragma unroll
    for (int xt = 0; xt < 2; ++xt) {
        if (xt >= RT) continue;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(part + ((wave * 4 + t) * 64 + lane) * 4) = accE[xt][t];
        __syncthreads();
        for (int e = tid; e < 4 * 256; e += NTH) {              // e = (t, lane, j)
            const int j = e & 3, ln = (e >> 2) & 63, t = e >> 8;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += part[((w * 4 + t) * 64 + ln) * 4 + j];
            const int n = t * 16 + (ln >> 4) * 4 + j, x = xt * 16 + (ln & 15);           // D layout: row n = 2j + dw, col x
            dEj[n * SMAX + x] = v;
        }
    }
    if (tid < 2 * F) {                                           // row sums and <ds0, E[f]> for the closed-form s0 terms
        const int f = tid % F;
        float sacc = 0.f;
        if (tid < F) { for (int h = 0; h < D; ++h) sacc += Es[f * Dp + h]; }
        else { for (int h = 0; h < D; ++h) sacc += Es[f * Dp + h] * a.dt1[(int64_t)b * a.t1w + h]; }
        rs[tid] = sacc;
    }
    __syncthreads();
    for (int e = tid; e < F * D; e += NTH) {
        const int f = e / D, h = e - f * D, lo = h & 1, hh = h >> 1;
        float R = 0.f, Q = 0.f;
        for (int j = f + 1; j < F; ++j) R += rs[j];
        for (int i = 0; i < f; ++i) Q += rs[F + i];
        a.dprev[(int64_t)b * F * D + e] = (dEi[(lo * F + f) * SMAX + hh] + dEj[(2 * f + lo) * SMAX + hh])
                                          + a.dt1[(int64_t)b * a.t1w + h] * R + Q;
    }
}

template <int NW>
static int launch_conv0_fact_tile_dgrad2(const DgradArgs& a, float* wpack, hipStream_t st) {
    const int QT = a.Pp / 16, CE = (a.F + 3) / 4;
    float* WA = wpack;
    float4* WE = reinterpret_cast<float4*>(wpack + (int64_t)2 * a.F * QT * 1024);
    hipLaunchKernelGGL(pack_w0_tile_kernel, dim3(2 * a.F * QT), dim3(256), 0, st, a.W, WA, WE, a.F, a.Pp);
    CFFM_CHECK_LAUNCH();
    // Es [4*CE + 1][D+1] | dCt [32][260] | Tg [16][258] | dTg [4096]
    const size_t lds = (size_t)(((4 * CE + 1) * (a.D + 1) + 3) / 4 * 4 + 32 * 260 + 16 * 258 + 4096) * 4 + 16;
    int rc = set_lds(conv0_fact_tile_dgrad2_kernel<32, NW>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((conv0_fact_tile_dgrad2_kernel<32, NW>), dim3(a.B), dim3(64 * NW), lds, st, a, (const float*)WA, (const float4*)WE);
    CFFM_CHECK_LAUNCH();
    return 0;
}

static int launch_conv0_fact_tile_dgrad(const DgradArgs& a, hipStream_t st, float* wpack = nullptr) {
    const char* ver = getenv("CFFM_TILE_DGRAD");                // debug: 1 = the round-2 kernel, 8 = packed fragments with 8 wavefronts
    if (wpack != nullptr && a.F <= 32 && a.D / 2 <= 32 && !(ver && ver[0] == '1')) {
        if (ver && ver[0] == '8') return launch_conv0_fact_tile_dgrad2<8>(a, wpack, st);
        return launch_conv0_fact_tile_dgrad2<4>(a, wpack, st);
    }

    const int S = a.D / 2;
    if (S > 32 || 2 * a.F > 64) return CFFM_ERR_UNSUPPORTED;
    constexpr int NW = 4;                // measured at F32 D64 B8192: 118 ms with 4 wavefronts, 194 ms with 8 (register spills)
    // Es | dCt [32][260] | Tg [16][258] | dTg [4096] | rs [2F]
    const size_t lds = (size_t)((a.F * (a.D + 1) + 3) / 4 * 4 + 32 * 260 + 16 * 258 + 4096 + 2 * a.F) * 4 + 16;
    int rc = set_lds(conv0_fact_tile_dgrad_kernel<32, NW>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((conv0_fact_tile_dgrad_kernel<32, NW>), dim3(a.B), dim3(64 * NW), lds, st, a);
    CFFM_CHECK_LAUNCH();
    return 0;
}

// Layer-0 forward of the wide shapes, ONE workgroup per example walking the (column tile, channel tile) pairs - the round-2
// kernel above launches a workgroup per (example, tile): 508 K workgroups at F32 D64 B8192, each of which gathers the
// example's 32 embedding rows again, and with every load, MFMA and store switched off it still ran 11.6 of its 29 ms.  Here
// the rows are staged once per example, step 1 reads the filter as the pre-packed fragments WA of the input gradient
// (pack_w0_tile_kernel: one load at (wave-uniform base)[lane] per fragment, zeros where j <= i), its E operands stay in
// registers across the channel tiles of a column tile, and the units are dealt to the wavefronts in balanced pairs of
// passes: wavefront (dh, w') takes fields i = 4w' .. 4w'+3 (k-steps w' .. 7) and i = 28-4w' .. 31-4w' (k-steps 7-w' .. 7):
// 72 MFMAs each, where consecutive units per wavefront gave 128 / 96 / 64 / 32.
template <int NW>
__global__ __launch_bounds__(64 * NW, 4) void conv0_fact_tile_fwd2_kernel(ConvArgs a, const float* __restrict__ WA) {
    static_assert(NW == 8, "unit passes below are laid out for 8 wavefronts: (dh, w') = (wave >> 2, wave & 3)");
    constexpr int NTH = 64 * NW, XQ = 16 / NW, TP = 16 * 16 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int F = a.F, D = a.D, S = D / 2, RT = S / 16, PpT = a.Pp, QT = PpT / 16;
    const int CE = (F + 3) / 4, Fp = 4 * CE, FpZ = Fp + 1, SP = S + 1, K2p = 2 * Fp;
    // The embedding rows are kept DE-INTERLEAVED, E2[(d, f)][z] = E_f[2z + d] (rows f = F .. Fp zero): both steps read this one
    // matrix as an MFMA operand - step 1 as [(dw, j)][x], step 2 as [(dh, i)][y] - at lane-constant offset + scalar.
    float* T = reinterpret_cast<float*>(smem);                 // [K2p][TP]  planes m = dh*Fp + i (planes of i >= F stay zero)
    float* E2 = T + K2p * TP;                                   // [2][FpZ][SP]
    float* PS = E2 + (2 * FpZ * SP + 3) / 4 * 4;                // [NW][S] pool partials of the wavefronts
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, kk = lane >> 4;
    const unsigned ulane = lane;
    const int b = blockIdx.x;
    {
        const float invD = 1.f / (float)D;
        for (int e = tid; e < F * D; e += NTH) {
            const int f = (int)(((float)e + 0.5f) * invD), d = e - f * D;
            const float v = a.idx == nullptr ? a.in[((int64_t)b * F + f) * D + d] : row_ptr(a.in, a.idx, a.idxM, (int64_t)b * F + f, D, a.idxStride)[d];
            E2[((d & 1) * FpZ + f) * SP + (d >> 1)] = v;
        }
        for (int e = tid; e < 2 * (FpZ - F) * SP; e += NTH) {
            const int d = e / ((FpZ - F) * SP), o = e - d * (FpZ - F) * SP;
            E2[(d * FpZ + F) * SP + o] = 0.f;
        }
        for (int e = tid; e < 2 * (Fp - F) * TP; e += NTH) {   // the planes of the padding fields
            const int d = e / ((Fp - F) * TP), o = e - d * (Fp - F) * TP;
            T[(d * Fp + F) * TP + o] = 0.f;
        }
    }
    const int dhW = wave >> 2, wq = wave & 3;
    float* outb = a.out + (int64_t)b * S * S * PpT;
    const float* e2l = E2 + kk * SP + r;                        // lane-constant part of both operand addresses
    const float* tl = T + kk * TP + r;
    lds_barrier();
#pragma clang loop unroll(disable)
    for (int xt = 0; xt < RT; ++xt) {
        const int x0 = xt * 16;
#pragma clang loop unroll(disable)
        for (int qt = 0; qt < QT; ++qt) {
            const int q0 = qt * 16;
            // ---- step 1: T planes of this wavefront's units, two passes of four consecutive fields -------------------------
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int i0 = pass == 0 ? 4 * wq : 28 - 4 * wq;
                if (i0 >= F) continue;                          // wave-uniform
                const int cs = (i0 + 1) >> 2;                   // first k-step with a field j > i0
                f32x4 acc[4];
                const float* wa[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    const int m = dhW * F + min(i0 + u, F - 1);
                    wa[u] = WA + __builtin_amdgcn_readfirstlane((m * QT + qt) * 1024);
                }
                // k-steps cs .. CE-1 in a run-time loop, DEPTH of them in flight (k-step d + DEPTH is requested when k-step d is
                // multiplied).  (All eight unrolled behind wave-uniform tests: every fragment was spilled right after its load.)
                constexpr int DEPTH = 3;
                float w0[DEPTH][4], w1[DEPTH][4], e0[DEPTH], e1[DEPTH];
#define CFFM_F_REQ(SLOT, C)                                                                                                   \
    if ((C) < CE) {                                                                                                           \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                       \
            w0[SLOT][u] = (wa[u] + (C) * 64)[ulane]; w1[SLOT][u] = (wa[u] + ((C) + 8) * 64)[ulane];                           \
        }                                                                                                                     \
        e0[SLOT] = e2l[4 * (C) * SP + x0]; e1[SLOT] = e2l[(FpZ + 4 * (C)) * SP + x0];                                         \
    }
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) { CFFM_F_REQ(d, cs + d) }
                for (int c = cs; c < CE; c += DEPTH) {
#pragma unroll
                    for (int d = 0; d < DEPTH; ++d) {
                        if (c + d < CE) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                acc[u] = mfma16(e0[d], w0[d][u], acc[u]);
                                acc[u] = mfma16(e1[d], w1[d][u], acc[u]);
                            }
                            CFFM_F_REQ(d, c + d + DEPTH)
                        }
                    }
                }
#undef CFFM_F_REQ
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (i0 + u >= F) continue;
                    float* tp = T + (dhW * Fp + i0 + u) * TP + r;
#pragma unroll
                    for (int j = 0; j < 4; ++j) tp[(kk * 4 + j) * 16] = acc[u][j];
                }
            }
            lds_barrier();                                      // T written (and the pool partials of the previous tile read)
            // ---- step 2: C[y][x][q] = relu(b[q] + sum_k E_i[2y+dh] * T[k = (dh,i)][x][q]), this wavefront's XQ columns --------
            const float bias = a.bias[q0 + r];
            const int xg = wave * XQ;
#pragma clang loop unroll(disable)
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 acc[XQ];
#pragma unroll
                for (int q4 = 0; q4 < XQ; ++q4) acc[q4] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma clang loop unroll(disable)
                for (int dh = 0; dh < 2; ++dh) {                // K = (dh, i): per tap the CE k-steps of its Fp fields, requested together
                    float av[8], bv[8][XQ];
#pragma unroll
                    for (int s2 = 0; s2 < 8; ++s2) {
                        const int sc = min(s2, CE - 1);
                        av[s2] = e2l[(dh * FpZ + 4 * sc) * SP + rt * 16];
#pragma unroll
                        for (int q4 = 0; q4 < XQ; ++q4) bv[s2][q4] = tl[(dh * Fp + 4 * sc) * TP + (xg + q4) * 16];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s2 = 0; s2 < 8; ++s2) {
                        if (s2 >= CE) continue;
#pragma unroll
                        for (int q4 = 0; q4 < XQ; ++q4) acc[q4] = mfma16(av[s2], bv[s2][q4], acc[q4]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                float ps[4] = {0.f, 0.f, 0.f, 0.f};
                unsigned long long mine = 0;                    // relu mask: lane (kk, r = 4*q4 + j) keeps the ballot of element (q4, j)
#pragma unroll
                for (int q4 = 0; q4 < XQ; ++q4)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int yy = rt * 16 + kk * 4 + j;
                        const float c = fmaxf(acc[q4][j] + bias, 0.f);
                        __builtin_nontemporal_store(c, &outb[((int64_t)yy * S + x0 + xg + q4) * PpT + q0 + r]);
                        ps[j] += act_pos(c, a.act);
                        const unsigned long long bal = __ballot(c > 0.f);    // bits 16kk .. 16kk+15: the 16 channels of pixel (yy, x)
                        if (r == 4 * q4 + j) mine = bal;
                    }
                if (a.relu != nullptr && r < 4 * XQ) {          // ONE 2-byte store per lane (4*XQ*4 lanes) instead of one per ballot
                    const int q4 = r >> 2, yy = rt * 16 + kk * 4 + (r & 3);
                    a.relu[(((int64_t)b * S + yy) * S + x0 + xg + q4) * QT + qt] = (uint16_t)(mine >> (16 * kk));
                }
                if (a.pool != nullptr) {                        // pool partial of this (column tile, channel tile): over q in the DPP row
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = row_group_sum<true>(ps[j]);
                        if (r == 0) PS[(wave * RT + rt) * 16 + kk * 4 + j] = v;      // [NW][S]
                    }
                }
            }
            lds_barrier();                                      // T consumed, pool partials written
            if (a.pool != nullptr && tid < S) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) v += PS[w * S + tid];                  // wave order: x0 + w * XQ ascending
                a.pool[((int64_t)b * S + tid) * a.pool_np + xt * QT + qt] = v;
            }
        }
    }
}

static int launch_conv0_fact_tile_fwd2(const ConvArgs& a, float* wpack, hipStream_t st) {
    constexpr int NW = 8;
    const int QT = a.Pp / 16, CE = (a.F + 3) / 4, S = a.D / 2;
    float* WA = wpack;
    float4* WE = reinterpret_cast<float4*>(wpack + (int64_t)2 * a.F * QT * 1024);
    hipLaunchKernelGGL(pack_w0_tile_kernel, dim3(2 * a.F * QT), dim3(256), 0, st, a.W, WA, WE, a.F, a.Pp);
    CFFM_CHECK_LAUNCH();
    // T [2*Fp][272] | E2 [2][Fp + 1][S + 1] | PS [NW][S]
    const size_t lds = (size_t)(8 * CE * (16 * 16 + 16) + (2 * (4 * CE + 1) * (S + 1) + 3) / 4 * 4 + NW * S) * 4 + 16;
    int rc = set_lds(conv0_fact_tile_fwd2_kernel<NW>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((conv0_fact_tile_fwd2_kernel<NW>), dim3(a.B), dim3(64 * NW), lds, st, a, (const float*)WA);
    CFFM_CHECK_LAUNCH();
    return 0;
}

static int launch_conv0_fact_tile_fwd(const ConvArgs& a, hipStream_t st, float* wpack = nullptr) {
    {
        const char* ver = getenv("CFFM_TILE_FWD");              // debug: 1 = the round-2 kernel (a workgroup per tile)
        if (wpack != nullptr && a.F <= 32 && a.D / 2 <= 32 && !(ver && ver[0] == '1')) return launch_conv0_fact_tile_fwd2(a, wpack, st);
    }

    constexpr int NW = 8;                // measured at F32 D64 B8192: 38.6 ms with 4 wavefronts, 29.2 with 8, 36.6 with 16 (one workgroup per CU)
    const int S = a.D / 2;
    const size_t lds = (size_t)(2 * a.F * (16 * 16 + 16) + a.F * (a.D + 1) + NW * S) * 4 + 16;
    int rc = set_lds(conv0_fact_tile_fwd_kernel<NW>, lds);
    if (rc) return rc;
    const int64_t grid = (int64_t)a.B * (a.Pp / 16) * (S / 16);
    if (grid > 0x7fffffffll) return CFFM_ERR_UNSUPPORTED;
#ifdef CFFM_TILE_DBG
    ConvArgs b = a;
    b.dbg = getenv("CFFM_DBG") ? atoi(getenv("CFFM_DBG")) : 0;
    hipLaunchKernelGGL((conv0_fact_tile_fwd_kernel<NW>), dim3((unsigned)grid), dim3(64 * NW), lds, st, b);
#else
    hipLaunchKernelGGL((conv0_fact_tile_fwd_kernel<NW>), dim3((unsigned)grid), dim3(64 * NW), lds, st, a);
#endif
    CFFM_CHECK_LAUNCH();
    return 0;
}

template <int NT>
static int launch_conv0_fact_fwd(const ConvArgs& a, hipStream_t st) {
    constexpr int PP = NT * 16;
    const int S = a.D / 2;
    const size_t lds = (size_t)(4 * PP * PP + 2 * a.F * (S * PP + 16) + a.F * (a.D + 1)) * 4 + 16;
    int rc = set_lds(conv0_fact_fwd_kernel<NT>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((conv0_fact_fwd_kernel<NT>), dim3(a.B), dim3(256), lds, st, a);
    CFFM_CHECK_LAUNCH();
    return 0;
}

// the factorised layer-0 kernels keep the filter, the embedding tile and T[2F][S][Pp] of one example in LDS
static inline bool conv0_fact_ok(const Geo& g) {
    const int S = g.D / 2;
    if (g.Pp > 64 || (S != 16 && S != 32)) return false;
    const size_t lds = (size_t)(4 * g.Pp * g.Pp + 2 * g.F * (S * g.Pp + 16) + g.F * (g.D + 1)) * 4 + 16;
    return lds <= 150 * 1024;
}

// LDS of conv0_fact_bwd_body with its 16 wavefronts: filter, T planes, partial tiles, dEi / dEj, bias partials, row sums,
// embedding tile
static inline size_t conv0_fact_bwd_lds(int Pp, int F, int D) {
    const int NW = 16;
    return (size_t)(4 * Pp * Pp + 2 * F * (16 * Pp + 16) + NW * 16 * 32 + 2 * 32 * 16 + NW * Pp + 2 * F + F * (D + 1)) * 4 + 16;
}
static inline bool conv0_fact_bwd_ok(const Geo& g) {
    if (g.Pp > 64 || g.D != 32) return false;
    return conv0_fact_bwd_lds(g.Pp, g.F, g.D) <= 160 * 1024 - 512;
}

template <int NT>
static int launch_conv0_fact_bwd(const DgradArgs& a, float* slabW, float* slabB, int64_t stride, int nsl, hipStream_t st) {
    constexpr int PP = NT * 16;
    constexpr int NW = 16;
    const size_t lds = conv0_fact_bwd_lds(PP, a.F, a.D);
#define CFFM_C0B_LAUNCH(FV, DV)                                                                                  \
    do {                                                                                                         \
        int rc = set_lds(conv0_fact_bwd_kernel<NT, FV, DV, NW>, lds);                                                \
        if (rc) return rc;                                                                                       \
        hipLaunchKernelGGL((conv0_fact_bwd_kernel<NT, FV, DV, NW>), dim3(nsl), dim3(64 * NW), lds, st, a, slabW, slabB, stride); \
    } while (0)
    if (NT == 3 && a.F == 10 && a.D == 32) CFFM_C0B_LAUNCH(10, 32);          // frappe        (README.md:28)
    else if (NT == 1 && a.F == 6 && a.D == 32) CFFM_C0B_LAUNCH(6, 32);       // book-crossing (README.md:20)
    else if (NT == 1 && a.F == 3 && a.D == 32) CFFM_C0B_LAUNCH(3, 32);       // ml-tag        (README.md:24)
    else CFFM_C0B_LAUNCH(0, 0);
#undef CFFM_C0B_LAUNCH
    CFFM_CHECK_LAUNCH();
    return 0;
}

template <int NT, int RM, bool L0, int HALVES>
static int launch_dgrad_taps(const DgradArgs& a, hipStream_t st) {
    constexpr int PP = NT * 16, BM = 16 * RM;
    const int S2 = 1 << (2 * a.lgSo);
    const int rows_per_wg = L0 ? (S2 > BM ? S2 : BM) : BM;
    const int n_ex = L0 ? rows_per_wg / S2 : 0;
    const int So = 1 << a.lgSo;
    const bool fast = L0 && RM == 4 && a.lgSo >= 4 && a.lgSo <= 6;
    const size_t scratch = fast ? (size_t)(4 + 4 * HALVES) * PP * So : (size_t)4 * HALVES * n_ex * a.F * (a.D + 1);
    const size_t lds = (L0 ? (size_t)PP + n_ex * a.F * (a.D + 1) + 2 * n_ex * a.F + scratch : 0) * 4 + 16;
    int rc = set_lds(dgrad_taps_kernel<NT, RM, L0, HALVES>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((dgrad_taps_kernel<NT, RM, L0, HALVES>), dim3((unsigned)((a.Mtot + rows_per_wg - 1) / rows_per_wg)),
                       dim3(256 * HALVES), lds, st, a);
    CFFM_CHECK_LAUNCH();
    return 0;
}

#define DISPATCH_NT4(NTV, CALL)                    \
    switch (NTV) {                                 \
        case 1: { constexpr int NT_ = 1; CALL; } break; \
        case 2: { constexpr int NT_ = 2; CALL; } break; \
        case 3: { constexpr int NT_ = 3; CALL; } break; \
        default: { constexpr int NT_ = 4; CALL; } break; \
    }

template <int NT, bool GEN>
static int launch_wgrad_taps(const WgradArgs& a, int nsl, hipStream_t st) {
    const int64_t rows_per_slab = (a.Mtot + nsl - 1) / nsl;
    const int S2 = 1 << (2 * a.lgSo);
    const int n_ex_max = GEN ? WGT_SUB / S2 + 2 : 0;
    const size_t base = (size_t)(WGT_SUB * NT * 16 + (GEN ? NT * 16 + (n_ex_max * a.F * (a.D + 1) + 7) / 4 * 4 : 0)) * 4 + 16;
    if (rows_per_slab >= 128) {      // enough rows to give two wavefronts per SIMD something to do
        const size_t lds = base + (size_t)4 * NT * NT * 64 * 16 + (size_t)NT * 64 * 4;
        int rc = set_lds(wgrad_taps_kernel<NT, GEN, 2>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((wgrad_taps_kernel<NT, GEN, 2>), dim3(nsl), dim3(512), lds, st, a);
    } else {
        int rc = set_lds(wgrad_taps_kernel<NT, GEN, 1>, base);
        if (rc) return rc;
        hipLaunchKernelGGL((wgrad_taps_kernel<NT, GEN, 1>), dim3(nsl), dim3(256), base, st, a);
    }
    CFFM_CHECK_LAUNCH();
    return 0;
}

template <int NT, int RM>
static int launch_conv_bwd_pair(const DgradArgs& d, const WgradArgs& w, int nsl, const InnerBwdArgs* ib, int n_i, hipStream_t st,
                                const TopWgrad* twp = nullptr) {
    constexpr int BM = 16 * RM;
    TopWgrad tw;
    memset(&tw, 0, sizeof(tw));
    if (twp) tw = *twp;
    const int n_d = (int)((d.Mtot + BM - 1) / BM);
    // a slab of a paired layer has fewer than 128 rows (conv_pair_ok): half of WGT_SUB, so that four workgroups share a CU
    size_t lds = (size_t)(128 * NT * 16) * 4 + 16;
    InnerBwdArgs none;
    memset(&none, 0, sizeof(none));
    if (ib && inner_bwd_lds(ib->g) > lds) lds = inner_bwd_lds(ib->g);
    // XCD alignment of the two roles (see the kernel): one weight-gradient slab per example, whole groups of 8 examples, and
    // both roles starting at a block index that is a multiple of 8
    const int S2 = 1 << (2 * d.lgSo);
    const int first_w = ib ? n_i : 0, first_d = first_w + nsl + tw.n[0] + tw.n[1];
    const int xcd_align = (nsl == d.B && (int64_t)nsl * S2 == d.Mtot && S2 % BM == 0 && d.B % 8 == 0 && first_w % 8 == 0 &&
                           first_d % 8 == 0) ? 1 : 0;
    // activation compiled in for the README commands (see launch_fwd_all)
#define PAIR_GO(ACT_)                                                                                                          \
    do {                                                                                                                        \
        int rc_ = set_lds(conv_bwd_pair_kernel<NT, RM, ACT_>, lds);                                                             \
        if (rc_) return rc_;                                                                                                    \
        hipLaunchKernelGGL((conv_bwd_pair_kernel<NT, RM, ACT_>), dim3(n_d + nsl + (ib ? n_i : 0) + tw.n[0] + tw.n[1]), dim3(256), \
                           lds, st, d, w, n_d, nsl, ib ? *ib : none, ib ? n_i : 0, tw, xcd_align);                              \
    } while (0)
    if (NT == 3 && d.act == CFFM_ACT_SELU) PAIR_GO(CFFM_ACT_SELU);
    else if (NT == 1 && d.act == CFFM_ACT_ELU) PAIR_GO(CFFM_ACT_ELU);
    else if (NT == 1 && d.act == CFFM_ACT_RELU) PAIR_GO(CFFM_ACT_RELU);
    else PAIR_GO(-1);
#undef PAIR_GO
    CFFM_CHECK_LAUNCH();
    return 0;
}

static inline int64_t layer_rows(const Geo& g, int B, int l, int* lgSo) {
    const int So = g.D >> (l + 1);
    *lgSo = ilog2_i(So);
    return (int64_t)B * So * So;
}
// offset inside t1 of the pool taken from the INPUT of layer l (s_l), width D >> l
static inline int t1_offset(const Geo& g, int l) {
    int off = 0;
    for (int i = 0; i < l; ++i) off += g.D >> i;
    return off;
}

static int conv_fwd_any(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, int l, hipStream_t st,
                        const RowSrc* rs = nullptr) {
    cffm_theta_layout_t tl; cffm_ws_layout_t wl;
    cffm_theta_layout(s, &tl); cffm_ws_layout(s, B, &wl);
    const Geo g = make_geo(s);
    if (l < 0 || l >= g.live) return CFFM_ERR_BAD_SHAPE;
    char* w = (char*)ws;
    ConvArgs a;
    a.in = (const float*)(w + (l == 0 ? wl.Eo : wl.C[l - 1]));
    a.W = theta + tl.conv_w[l]; a.bias = theta + tl.conv_b[l];
    a.out = (float*)(w + wl.C[l]);
    a.Mtot = layer_rows(g, B, l, &a.lgSo);
    a.B = B; a.P = g.P; a.Pp = g.Pp; a.F = g.F; a.D = g.D; a.act = g.act;
    if (wl.pool_np[l] > 0) { a.pool = (float*)(w + wl.pool[l]); a.pool_np = wl.pool_np[l]; }   // wide shapes: the epilogue leaves the pool partials
    if (l >= 1 && wl.wb3_bytes > 0) a.wb3 = (void*)(w + wl.wb3);
    if (wl.relu0 > 0 && l + 1 < g.live) a.relu = (uint16_t*)(w + wl.relu0 + relu_mask_off(g, B, l));   // ... and the relu mask of C_l for the input gradient of layer l+1
    int nblk, NT;
    int rc = 0;
    if (g.Pp <= 64) {                       // tap-split path: one wave per filter tap, no K loop
        const int nt4 = g.Pp / 16;
        const int64_t wg16 = (a.Mtot + 15) / 16;
        if (l == 0 && conv0_fact_ok(g)) {        // rank-1 input channels: factorised contraction, one workgroup per example
            DISPATCH_NT4(nt4, rc = (launch_conv0_fact_fwd<NT_>(a, st)));
            return rc;
        }
        if (wg16 >= 4 * 512) {                    // many rows: whole filter in LDS, a wave runs all four taps
            if (l == 0) { DISPATCH_NT4(nt4, rc = (launch_conv_fwd_rows<NT_, 1, true>(a, st))); }
            else { DISPATCH_NT4(nt4, rc = (launch_conv_fwd_rows<NT_, 1, false>(a, st))); }
            return rc;
        }
        if (l == 0) {
            if (wg16 >= 4 * 1024) { DISPATCH_NT4(nt4, rc = (launch_conv_fwd_taps<NT_, 4, true>(a, st))); }
            else if (wg16 >= 2 * 512) { DISPATCH_NT4(nt4, rc = (launch_conv_fwd_taps<NT_, 2, true>(a, st))); }
            else { DISPATCH_NT4(nt4, rc = (launch_conv_fwd_taps<NT_, 1, true>(a, st))); }
        } else {
            if (wg16 >= 4 * 1024) { DISPATCH_NT4(nt4, rc = (launch_conv_fwd_taps<NT_, 4, false>(a, st))); }
            else if (wg16 >= 2 * 512) { DISPATCH_NT4(nt4, rc = (launch_conv_fwd_taps<NT_, 2, false>(a, st))); }
            else { DISPATCH_NT4(nt4, rc = (launch_conv_fwd_taps<NT_, 1, false>(a, st))); }
        }
        return rc;
    }
    // rank-1 input channels: factorised, channel-tiled (its step 1 indexes k over ALL fields: 2 * ceil4(F) <= 64 k values)
    if (l == 0 && conv0_fact_tile_ok(g) && 2 * ((g.F + 3) & ~3) <= 4 * C0T_MAXKS) {
        if (rs) { a.in = rs->base; a.idx = rs->idx; a.idxM = rs->M; a.idxStride = rs->stride; }     // rows straight from the outer table (RowSrc)
        return launch_conv0_fact_tile_fwd(a, st, wl.w0pack_floats > 0 ? (float*)(w + wl.w0pack) : nullptr);
    }
    if (l == 0 && rs) return CFFM_ERR_UNSUPPORTED;                  // cffm_wide_regather_ok() guards the callers
    pick_nt(g.Pp / 16, &nblk, &NT);
    const bool big = a.Mtot >= 128 * 256;
    if (l == 0) {
        if (big) { DISPATCH_NT(NT, rc = (launch_conv_fwd<NT_, 2, true>(a, nblk, st))); }
        else { DISPATCH_NT(NT, rc = (launch_conv_fwd<NT_, 1, true>(a, nblk, st))); }
    } else {
        if (big) { DISPATCH_NT(NT, rc = (launch_conv_fwd<NT_, 2, false>(a, nblk, st))); }
        else { DISPATCH_NT(NT, rc = (launch_conv_fwd<NT_, 1, false>(a, nblk, st))); }
    }
    return rc;
}

// argument blocks of the tap-split backward of layer l >= 1
static void fill_taps_bwd_args(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, int l, DgradArgs* da_, WgradArgs* wa_) {
    cffm_theta_layout_t tl; cffm_ws_layout_t wl;
    cffm_theta_layout(s, &tl); cffm_ws_layout(s, B, &wl);
    const Geo g = make_geo(s);
    char* w = (char*)ws;
    float* gpart = (float*)(w + wl.gpart);
    SlabPlan sp;
    make_slab_plan(s, B, tl, &sp);
    const SlabRange& sr = sp.r[sp.conv0 + l];
    int lg;
    const int64_t Mtot = layer_rows(g, B, l, &lg);
    WgradArgs& wa = *wa_;
    wa.in = (const float*)(w + wl.C[l - 1]);
    wa.dC = (const float*)(w + wl.dC[l]);
    wa.slabW = gpart + sr.base; wa.slabB = wa.slabW + (tl.conv_b[l] - tl.conv_w[l]);
    wa.slab_stride = sr.len; wa.slabB_stride = sr.len;
    wa.Mtot = Mtot; wa.lgSo = lg;
    wa.B = B; wa.P = g.P; wa.Pp = g.Pp; wa.F = g.F; wa.D = g.D; wa.act = g.act; wa.qblocks = 0;
    DgradArgs& da = *da_;
    da.dC = wa.dC;
    da.W = theta + tl.conv_w[l];
    da.Cprev = wa.in;
    da.dt1 = (const float*)(w + wl.dt1);
    da.dprev = (float*)(w + wl.dC[l - 1]);
    da.Mtot = Mtot; da.lgSo = lg;
    da.B = B; da.P = g.P; da.Pp = g.Pp; da.F = g.F; da.D = g.D; da.act = g.act;
    da.t1w = 2 * g.D - 2; da.t1off = t1_offset(g, l);
}

template <int NT>
static int launch_bwd_top(const BwdTopArgs& a, size_t lds, hipStream_t st) {
    const int act = a.hb.g.act;
#define TOP_GO(ACT_)                                                                                          \
    do {                                                                                                       \
        int rc_ = set_lds(bwd_top_kernel<NT, ACT_>, lds);                                                      \
        if (rc_) return rc_;                                                                                   \
        hipLaunchKernelGGL((bwd_top_kernel<NT, ACT_>), dim3(a.n_inner + a.n_keys + 256), dim3(256), lds, st, a); \
    } while (0)
    if (NT == 3 && act == CFFM_ACT_SELU) TOP_GO(CFFM_ACT_SELU);
    else if (NT == 1 && act == CFFM_ACT_ELU) TOP_GO(CFFM_ACT_ELU);
    else if (NT == 1 && act == CFFM_ACT_RELU) TOP_GO(CFFM_ACT_RELU);
    else TOP_GO(-1);
#undef TOP_GO
    CFFM_CHECK_LAUNCH();
    return 0;
}

// head backward + the top conv layers (+ the inner-branch backward) in one launch; *next_layer receives the highest
// conv layer the caller still has to run (layers >= 1 below it, then layer 0)
int cffm_bwd_top_impl(const cffm_shape_t* s, const float* theta, void* ws, const float* y, int32_t B, int64_t B_global,
                      bool local_sum, float* loss_out, bool unscaled, hipStream_t st, int* next_layer, const int32_t* rank_ids) {
    const Geo g = make_geo(s);
    BwdTopArgs a;
    memset(&a, 0, sizeof(a));
    fill_head_bwd_args(s, theta, ws, y, B, B_global, local_sum, loss_out, unscaled, &a.hb);
    const int first = bwd_top_first_layer(s);
    a.wgrad_here = (top_wgrad_deferred(s, B) || bwd_fused01_ok(s, B)) ? 0 : 1;   // 0: a later launch computes them
    a.n_layers = 0;
    for (int l = g.live - 1; l >= first; --l) {
        fill_taps_bwd_args(s, theta, ws, B, l, &a.d[a.n_layers], &a.w[a.n_layers]);
        a.lgSo[a.n_layers] = a.d[a.n_layers].lgSo;
        ++a.n_layers;
    }
    a.n_inner = fill_inner_bwd_args(s, theta, ws, B, &a.ib);
    a.ib.dout = nullptr;                                     // recomputed from (out, y, L): no dependency on the head role
    a.ib.y = y; a.ib.invB = 1.f / (float)B_global;
    if (rank_ids != nullptr && s->F <= RANK_MAXF && (int64_t)B * s->F <= 4096) {
        cffm_ws_layout_t wl;
        cffm_ws_layout(s, B, &wl);
        a.rank_ids = rank_ids; a.n_rows = B * s->F;
        a.keys_sorted = (unsigned long long*)((char*)ws + wl.sort_vals);
        a.n_keys = 256;
    }
    size_t lds = (size_t)(WGT_SUB * g.Pp) * 4 + 16;
    if (inner_bwd_lds(g) > lds) lds = inner_bwd_lds(g);
    int rc = 0;
    DISPATCH_NT4(g.Pp / 16, rc = (launch_bwd_top<NT_>(a, lds, st)));
    *next_layer = first - 1;
    return rc;
}

// which: bit 0 = weight/bias gradient, bit 1 = input gradient (the two only share their inputs, so the fused
// step runs them on different streams)
static int conv_bwd_any(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, int l, hipStream_t st, int which = 3,
                        bool* with_inner = nullptr, bool with_top_wgrad = false, const RowSrc* rs = nullptr) {
    cffm_theta_layout_t tl; cffm_ws_layout_t wl;
    cffm_theta_layout(s, &tl); cffm_ws_layout(s, B, &wl);
    const Geo g = make_geo(s);
    if (l < 0 || l >= g.live) return CFFM_ERR_BAD_SHAPE;
    char* w = (char*)ws;
    float* gpart = (float*)(w + wl.gpart);
    SlabPlan sp;
    make_slab_plan(s, B, tl, &sp);
    const SlabRange& sr = sp.r[sp.conv0 + l];
    int rc = 0;
    if (l == 0 && which == 3 && conv0_fact_bwd_ok(g)) {   // factorised layer 0: weight, bias and input gradient in one kernel
        DgradArgs a;
        a.dC = (const float*)(w + wl.dC[0]);
        a.W = theta + tl.conv_w[0];
        a.Cprev = (const float*)(w + wl.Eo);
        a.dt1 = (const float*)(w + wl.dt1);
        a.dprev = (float*)(w + wl.dEo);
        a.Mtot = layer_rows(g, B, 0, &a.lgSo);
        a.B = B; a.P = g.P; a.Pp = g.Pp; a.F = g.F; a.D = g.D; a.act = g.act;
        a.t1w = 2 * g.D - 2; a.t1off = 0;
        float* slabW = gpart + sr.base;
        DISPATCH_NT4(g.Pp / 16, rc = (launch_conv0_fact_bwd<NT_>(a, slabW, slabW + (tl.conv_b[0] - tl.conv_w[0]), sr.len, sr.nslab, st)));
        return rc;
    }
    if (which == 3 && l >= 1 && g.Pp <= 64) {
        int lg;
        const int64_t Mtot = layer_rows(g, B, l, &lg);
        if ((Mtot + sr.nslab - 1) / sr.nslab < 128) {          // the 256-thread weight-gradient variant: pair it up
            WgradArgs wa;
            wa.in = (const float*)(w + wl.C[l - 1]);
            wa.dC = (const float*)(w + wl.dC[l]);
            wa.slabW = gpart + sr.base; wa.slabB = wa.slabW + (tl.conv_b[l] - tl.conv_w[l]);
            wa.slab_stride = sr.len; wa.slabB_stride = sr.len;
            wa.Mtot = Mtot; wa.lgSo = lg;
            wa.B = B; wa.P = g.P; wa.Pp = g.Pp; wa.F = g.F; wa.D = g.D; wa.act = g.act; wa.qblocks = 0;
            DgradArgs da;
            da.dC = wa.dC;
            da.W = theta + tl.conv_w[l];
            da.Cprev = wa.in;
            da.dt1 = (const float*)(w + wl.dt1);
            da.dprev = (float*)(w + wl.dC[l - 1]);
            da.Mtot = Mtot; da.lgSo = lg;
            da.B = B; da.P = g.P; da.Pp = g.Pp; da.F = g.F; da.D = g.D; da.act = g.act;
            da.t1w = 2 * g.D - 2; da.t1off = t1_offset(g, l);
            const int64_t wg16 = (Mtot + 15) / 16;
            InnerBwdArgs ib;
            int n_i = 0;
            const bool inner = with_inner != nullptr && s->inner_conv;
            if (inner) n_i = fill_inner_bwd_args(s, theta, ws, B, &ib);
            TopWgrad tw;
            memset(&tw, 0, sizeof(tw));
            if (with_top_wgrad) {                               // the fused top left its weight gradients to this launch
                int k = 0;
                for (int lt = g.live - 1; lt > l && k < 2; --lt, ++k) {
                    DgradArgs unused;
                    fill_taps_bwd_args(s, theta, ws, B, lt, &unused, &tw.w[k]);
                    tw.n[k] = sp.r[sp.conv0 + lt].nslab;
                }
            }
            const TopWgrad* twp = with_top_wgrad ? &tw : nullptr;
            if (wg16 >= 2 * 512) { DISPATCH_NT4(g.Pp / 16, rc = (launch_conv_bwd_pair<NT_, 2>(da, wa, sr.nslab, inner ? &ib : nullptr, n_i, st, twp))); }
            else { DISPATCH_NT4(g.Pp / 16, rc = (launch_conv_bwd_pair<NT_, 1>(da, wa, sr.nslab, inner ? &ib : nullptr, n_i, st, twp))); }
            if (inner && !rc) *with_inner = true;
            return rc;
        }
    }
    if (which & 1) {   // weight / bias gradient
        WgradArgs a;
        a.in = (const float*)(w + (l == 0 ? wl.Eo : wl.C[l - 1]));
        a.dC = (const float*)(w + wl.dC[l]);
        a.slabW = gpart + sr.base; a.slabB = a.slabW + (tl.conv_b[l] - tl.conv_w[l]);
        a.slab_stride = sr.len; a.slabB_stride = sr.len;
        a.Mtot = layer_rows(g, B, l, &a.lgSo);
        a.B = B; a.P = g.P; a.Pp = g.Pp; a.F = g.F; a.D = g.D; a.act = g.act;
        int NT;
        if (g.Pp <= 64) {
            const int nt4 = g.Pp / 16;
            const int nsl = sr.nslab;
            if (l == 0) { DISPATCH_NT4(nt4, rc = (launch_wgrad_taps<NT_, true>(a, nsl, st))); }
            else { DISPATCH_NT4(nt4, rc = (launch_wgrad_taps<NT_, false>(a, nsl, st))); }
            if (rc) return rc;
        } else {
        pick_nt(g.Pp / 16, &a.qblocks, &NT);
        if (l == 0 && conv0_fact_tile_ok(g) && g.D / 2 <= 32) {
            if (rs) { a.in = rs->base; a.idx = rs->idx; a.idxM = rs->M; a.idxStride = rs->stride; }
            rc = launch_conv0_fact_tile_wgrad(a, sr.nslab, st);
        }
        else if (l == 0 && rs) { rc = CFFM_ERR_UNSUPPORTED; }
        else if (l == 0) { DISPATCH_NT(NT, rc = (launch_wgrad<NT_, true>(a, st))); }
        else if (NT == 8) { rc = launch_wgrad2<8>(a, st); }               // 128 x 128 output tile per workgroup
        else if (NT == 6) { rc = launch_wgrad2<6>(a, st); }               // 128 x 96
        else { DISPATCH_NT(NT, rc = (launch_wgrad<NT_, false>(a, st))); }
        if (rc) return rc;
        }
    }
    if (which & 2) {   // input gradient
        DgradArgs a;
        a.dC = (const float*)(w + wl.dC[l]);
        a.W = theta + tl.conv_w[l];
        a.Cprev = (const float*)(w + (l == 0 ? wl.Eo : wl.C[l - 1]));
        a.dt1 = (const float*)(w + wl.dt1);
        a.dprev = (float*)(w + (l == 0 ? wl.dEo : wl.dC[l - 1]));
        if (l >= 1 && wl.wb3_bytes > 0) a.wb3 = (void*)(w + wl.wb3);
        {   // wide shapes: the relu mask the forward of layer l-1 left (1/32 of the bytes of C_{l-1})
            const char* oldfwd = getenv("CFFM_TILE_FWD");
            if (l >= 1 && wl.relu0 > 0 && !(l == 1 && oldfwd && oldfwd[0] == '1') && !getenv("CFFM_DGRAD_NO_MASK"))
                a.relu = (const uint16_t*)(w + wl.relu0 + relu_mask_off(g, B, l - 1));
        }
        a.Mtot = layer_rows(g, B, l, &a.lgSo);
        a.B = B; a.P = g.P; a.Pp = g.Pp; a.F = g.F; a.D = g.D; a.act = g.act;
        a.t1w = 2 * g.D - 2; a.t1off = t1_offset(g, l);
        int nblk, NT;
        if (g.Pp <= 64) {
            const int nt4 = g.Pp / 16;
            const int64_t wg16 = (a.Mtot + 15) / 16;
            if (l == 0) {
                const int S2 = 1 << (2 * a.lgSo);
                if (S2 >= 128) { DISPATCH_NT4(nt4, rc = (launch_dgrad_taps<NT_, 4, true, 2>(a, st))); }   // >= 2 m tiles
                else { DISPATCH_NT4(nt4, rc = (launch_dgrad_taps<NT_, 4, true, 1>(a, st))); }
            }
            else if (wg16 >= 2 * 512) { DISPATCH_NT4(nt4, rc = (launch_dgrad_taps<NT_, 2, false, 1>(a, st))); }
            else { DISPATCH_NT4(nt4, rc = (launch_dgrad_taps<NT_, 1, false, 1>(a, st))); }
            return rc;
        }
        if (l == 0 && conv0_fact_tile_ok(g) && g.D / 2 <= 32 && 2 * g.F <= 64) {
            if (rs) { a.Cprev = rs->base; a.idx = rs->idx; a.idxM = rs->M; a.idxStride = rs->stride; }
            return launch_conv0_fact_tile_dgrad(a, st, wl.w0pack_floats > 0 ? (float*)(w + wl.w0pack) : nullptr);
        }
        if (l == 0 && rs) return CFFM_ERR_UNSUPPORTED;
        pick_nt(4 * g.Pp / 16, &nblk, &NT);
        const bool big = a.Mtot >= 128 * 256;
        if (l == 0) {
            if (big) { DISPATCH_NT(NT, rc = (launch_dgrad<NT_, 2, true>(a, nblk, st))); }
            else { DISPATCH_NT(NT, rc = (launch_dgrad<NT_, 1, true>(a, nblk, st))); }
        } else {
            if (big) { DISPATCH_NT(NT, rc = (launch_dgrad<NT_, 2, false>(a, nblk, st))); }
            else { DISPATCH_NT(NT, rc = (launch_dgrad<NT_, 1, false>(a, nblk, st))); }
        }
    }
    return rc;
}

int cffm_conv_bwd_part(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, int32_t layer, int which,
                       hipStream_t st) {
    return conv_bwd_any(s, theta, ws, B, layer, st, which);
}

int cffm_conv_bwd_with_inner(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, int32_t layer, hipStream_t st,
                             bool* inner_done) {
    *inner_done = false;
    return conv_bwd_any(s, theta, ws, B, layer, st, 3, inner_done);
}

extern "C" int cffm_outer_conv0_fwd(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, void* stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (B <= 0 || !s->outer_conv) return 0;
    return conv_fwd_any(s, theta, ws, B, 0, (hipStream_t)stream);
}
extern "C" int cffm_conv_fwd(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, int32_t layer, void* stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (B <= 0 || !s->outer_conv) return 0;
    if (layer < 1) return CFFM_ERR_BAD_SHAPE;
    return conv_fwd_any(s, theta, ws, B, layer, (hipStream_t)stream);
}
// Wide shapes whose layer 0 runs the three tiled factorised kernels: those can take the outer rows straight from the table
// (RowSrc), so the fused gather + inner-branch forward never writes Ei / Eo (cffm_gather_inner_fwd_wide, inner.hip)
bool cffm_wide_regather_ok(const cffm_shape_t* s) {
    if (check_shape(s) || !s->inner_conv || !s->outer_conv) return false;
    const Geo g = make_geo(s);
    return conv0_fact_tile_ok(g) && 2 * ((g.F + 3) & ~3) <= 4 * C0T_MAXKS && g.D / 2 <= 32 && 2 * g.F <= 64 &&
           g.K == g.D && (g.K == 32 || g.K == 64) && g.F <= 32 && cffm_giw_lds_ok();
}
int cffm_outer_conv0_fwd_rows(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, const RowSrc* rs, hipStream_t st) {
    if (B <= 0 || !s->outer_conv) return 0;
    return conv_fwd_any(s, theta, ws, B, 0, st, rs);
}
int cffm_outer_conv0_bwd_rows(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, const RowSrc* rs, hipStream_t st) {
    if (B <= 0 || !s->outer_conv) return 0;
    return conv_bwd_any(s, theta, ws, B, 0, st, 3, nullptr, false, rs);
}
extern "C" int cffm_outer_conv0_bwd(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, void* stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (B <= 0 || !s->outer_conv) return 0;
    return conv_bwd_any(s, theta, ws, B, 0, (hipStream_t)stream);
}
// the layer right below the fused top, carrying the top layers' weight gradients (top_wgrad_deferred)
int cffm_conv_bwd_below_top(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, int32_t layer, hipStream_t st) {
    return conv_bwd_any(s, theta, ws, B, layer, st, 3, nullptr, true);
}

// bwd_fused01_ok: layers 3..0 below the fused top in one launch (conv01_bwd_kernel)
int cffm_conv01_bwd_impl(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, hipStream_t st) {
    cffm_theta_layout_t tl; cffm_ws_layout_t wl;
    cffm_theta_layout(s, &tl); cffm_ws_layout(s, B, &wl);
    const Geo g = make_geo(s);
    if (!bwd_fused01_ok(s, B) || g.live != 4 || !conv0_fact_bwd_ok(g)) return CFFM_ERR_UNSUPPORTED;
    char* w = (char*)ws;
    float* gpart = (float*)(w + wl.gpart);
    SlabPlan sp;
    make_slab_plan(s, B, tl, &sp);
    Conv01Args a;
    memset(&a, 0, sizeof(a));
    const SlabRange& sr = sp.r[sp.conv0];
    a.d0.dC = (const float*)(w + wl.dC[0]);
    a.d0.W = theta + tl.conv_w[0];
    a.d0.Cprev = (const float*)(w + wl.Eo);
    a.d0.dt1 = (const float*)(w + wl.dt1);
    a.d0.dprev = (float*)(w + wl.dEo);
    a.d0.Mtot = layer_rows(g, B, 0, &a.d0.lgSo);
    a.d0.B = B; a.d0.P = g.P; a.d0.Pp = g.Pp; a.d0.F = g.F; a.d0.D = g.D; a.d0.act = g.act;
    a.d0.t1w = 2 * g.D - 2; a.d0.t1off = 0;
    a.slabW0 = gpart + sr.base; a.slabB0 = a.slabW0 + (tl.conv_b[0] - tl.conv_w[0]); a.stride0 = sr.len;
    DgradArgs unused;
    fill_taps_bwd_args(s, theta, ws, B, 1, &a.d1, &a.w1);
    fill_taps_bwd_args(s, theta, ws, B, 2, &unused, &a.w2);
    fill_taps_bwd_args(s, theta, ws, B, 3, &unused, &a.w3);
    a.n2 = sp.r[sp.conv0 + 2].nslab; a.n3 = sp.r[sp.conv0 + 3].nslab;
    if (sp.r[sp.conv0].nslab != 256 || sp.r[sp.conv0 + 1].nslab != 256 || a.n2 > 256 || a.n3 > 256 ||
        (int64_t)a.n2 * CFFM_TOP_SLAB_ROWS < a.w2.Mtot || (int64_t)a.n3 * CFFM_TOP_SLAB_ROWS < a.w3.Mtot)
        return CFFM_ERR_UNSUPPORTED;             // one layer-0/1 slab per workgroup, the 64-row slabs cover layers 2 and 3
    // Host-side guard (round-2 review, weak #6): the argument block starts zero-filled, and a role that dereferenced a member
    // nobody assigned would fault the GPU at address 0.  Every pointer ANY group of the kernel can dereference is checked here,
    // together with the row counts its clamped addresses are derived from (m = Mtot - 1 for rows past the end).
    {
        const void* must[] = {a.d0.dC, a.d0.W, a.d0.Cprev, a.d0.dt1, a.d0.dprev, a.slabW0, a.slabB0, a.d1.dC, a.d1.W, a.d1.Cprev,
                              a.d1.dt1, a.d1.dprev, a.w1.in, a.w1.dC, a.w1.slabW, a.w1.slabB, a.w2.in, a.w2.dC, a.w2.slabW,
                              a.w2.slabB, a.w3.in, a.w3.dC, a.w3.slabW, a.w3.slabB};
        for (const void* q : must)
            if (q == nullptr) return CFFM_ERR_BAD_SHAPE;
        if (a.d0.Mtot <= 0 || a.d1.Mtot <= 0 || a.w1.Mtot <= 0 || a.w2.Mtot <= 0 || a.w3.Mtot <= 0 || a.n2 < 1 || a.n3 < 1 ||
            a.stride0 <= 0 || a.w1.slab_stride <= 0 || a.w2.slab_stride <= 0 || a.w3.slab_stride <= 0)
            return CFFM_ERR_BAD_SHAPE;
    }
    const int NW = 16, PP = g.Pp;
    const size_t lds = conv0_fact_bwd_lds(PP, g.F, g.D);       // the layer-0 phases; the weight gradients need 3 x 64*PP*4
#define CFFM_C01_LAUNCH(NTV, FV, DV, ACTV)                                                                       \
    do {                                                                                                         \
        int rc = set_lds(conv01_bwd_kernel<NTV, FV, DV, ACTV>, lds);                                             \
        if (rc) return rc;                                                                                       \
        hipLaunchKernelGGL((conv01_bwd_kernel<NTV, FV, DV, ACTV>), dim3(256), dim3(1024), lds, st, a);          \
    } while (0)
    const int nt = PP / 16;
    if (nt == 3 && g.F == 10 && g.act == CFFM_ACT_SELU) CFFM_C01_LAUNCH(3, 10, 32, CFFM_ACT_SELU);     // frappe (README.md:28)
    else if (nt == 3) CFFM_C01_LAUNCH(3, 0, 0, -1);
    else if (nt == 2) CFFM_C01_LAUNCH(2, 0, 0, -1);
    else CFFM_C01_LAUNCH(1, 0, 0, -1);
#undef CFFM_C01_LAUNCH
    CFFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int cffm_conv_bwd(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, int32_t layer, void* stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (B <= 0 || !s->outer_conv) return 0;
    if (layer < 1) return CFFM_ERR_BAD_SHAPE;
    return conv_bwd_any(s, theta, ws, B, layer, (hipStream_t)stream);
}

// ---- fused forward (one launch) --------------------------------------------------------------------------------------
bool cffm_fwd_all_ok(const cffm_shape_t* s, int32_t B) {
    const Geo g = make_geo(s);
    return s->inner_conv && s->outer_conv && g.Pp <= 64 && conv0_fact_ok(g) && (int64_t)B * s->F <= 4096 && s->F <= RANK_MAXF;
}

template <int NT>
static int launch_fwd_all(const FwdAllArgs& fa, size_t lds, hipStream_t st) {
    constexpr int NW = 8;
    // the three README commands get their activation compiled in: frappe = selu at Pp 48, ml-tag = elu and book-crossing =
    // relu at Pp 16 (README.md:20-28)
    const int act = fa.inner.g.act;
#define FWD_ALL_GO(ACT_)                                                                                              \
    do {                                                                                                               \
        int rc_ = set_lds(fwd_all_kernel<NT, NW, ACT_>, lds);                                                          \
        if (rc_) return rc_;                                                                                           \
        hipLaunchKernelGGL((fwd_all_kernel<NT, NW, ACT_>), dim3(fa.B < 256 ? fa.B : 256), dim3(64 * NW), lds, st, fa); \
    } while (0)
    if (NT == 3 && act == CFFM_ACT_SELU) FWD_ALL_GO(CFFM_ACT_SELU);
    else if (NT == 1 && act == CFFM_ACT_ELU) FWD_ALL_GO(CFFM_ACT_ELU);
    else if (NT == 1 && act == CFFM_ACT_RELU) FWD_ALL_GO(CFFM_ACT_RELU);
    else FWD_ALL_GO(-1);
#undef FWD_ALL_GO
    CFFM_CHECK_LAUNCH();
    return 0;
}

int cffm_fwd_all_impl(const cffm_shape_t* s, const cffm_tables_t* tab, const float* theta, const int32_t* ids,
                      const float* y, int32_t B, void* ws, hipStream_t st, bool rank_keys) {
    cffm_theta_layout_t tl; cffm_ws_layout_t wl;
    cffm_theta_layout(s, &tl); cffm_ws_layout(s, B, &wl);
    const Geo g = make_geo(s);
    char* w = (char*)ws;
    FwdAllArgs fa;
    // inner branch + gather
    fa.inner.g = g; fa.inner.Ei = (const float*)(w + wl.Ei);
    fa.inner.cw = theta + tl.inner_cw; fa.inner.cb = theta + tl.inner_cb; fa.inner.wd = theta + tl.inner_dw;
    fa.inner.bd = theta + tl.inner_db; fa.inner.inner_out = (float*)(w + wl.inner_out);
    fa.inner.fg.ids = ids; fa.inner.fg.inner = tab->inner_emb; fa.inner.fg.outer = tab->outer_emb;
    fa.inner.fg.fbias = tab->feat_bias; fa.inner.fg.Ei = (float*)(w + wl.Ei); fa.inner.fg.Eo = (float*)(w + wl.Eo);
    fa.inner.fg.fb = (float*)(w + wl.fb); fa.inner.fg.keys = (unsigned long long*)(w + wl.sort_keys);
    fa.inner.fg.M = s->M; fa.inner.fg.D = s->D;
    // conv stack
    for (int l = 0; l < g.live; ++l) {
        ConvArgs& a = fa.conv[l];
        a.in = (const float*)(w + (l == 0 ? wl.Eo : wl.C[l - 1]));
        a.W = theta + tl.conv_w[l]; a.bias = theta + tl.conv_b[l];
        a.out = (float*)(w + wl.C[l]);
        a.Mtot = layer_rows(g, B, l, &a.lgSo);
        a.B = B; a.P = g.P; a.Pp = g.Pp; a.F = g.F; a.D = g.D; a.act = g.act;
    }
    // head
    HeadArgs& h = fa.head;
    h.g = g; h.B = B;
    h.Eo = (const float*)(w + wl.Eo); h.fb = (const float*)(w + wl.fb); h.inner_out = (const float*)(w + wl.inner_out);
    for (int l = 0; l < CFFM_MAX_LAYERS; ++l) h.C[l] = (const float*)(w + wl.C[l]);
    h.d1_w = theta + tl.d1_w; h.d1_b = theta + tl.d1_b; h.d2_w = theta + tl.d2_w; h.d2_b = theta + tl.d2_b;
    h.att_W = theta + tl.att_W; h.att_b = theta + tl.att_b; h.lin_w = theta + tl.lin_w; h.lin_b = theta + tl.lin_b;
    h.bias = theta + tl.bias;
    h.y = y;
    h.t1 = (float*)(w + wl.t1); h.h1 = (float*)(w + wl.h1); h.att = (float*)(w + wl.att);
    h.out = (float*)(w + wl.out); h.sqerr = (float*)(w + wl.sqerr);
    h.loss = s->loss; h.inner_conv = s->inner_conv; h.outer_conv = s->outer_conv;
    // sort workgroup
    fa.ids = ids; fa.keys_sorted = (unsigned long long*)(w + wl.sort_vals);
    fa.live = g.live; fa.n_rows = B * s->F; fa.B = B;
    fa.rank_keys = rank_keys ? 1 : 0;
    int bits = 1;
    while ((1ll << bits) <= (long long)s->M && bits < 31) ++bits;
    fa.id_bits = bits;
    const int S = g.D / 2, PP = g.Pp;
    size_t lds = inner_fwd_lds(g);
    const size_t l_fact = (size_t)(4 * PP * PP + 2 * g.F * (S * PP + 16) + g.F * (g.D + 1)) * 4 + 16;
    size_t l_taps = (size_t)4 * PP * (PP + 4) * 4;
    const size_t l_red = (size_t)4 * 4 * (PP / 16) * 64 * 16;
    if (l_red > l_taps) l_taps = l_red;
    l_taps += 64;
    if (l_fact > lds) lds = l_fact;
    if (l_taps > lds) lds = l_taps;
    if (head_fwd_lds(g) > lds) lds = head_fwd_lds(g);
    // LDS-resident activations: C_0 above every phase's scratch, C_1.. above the scratch of the tap kernels / the head
    {
        const size_t c0 = (size_t)S * S * PP * 4;
        size_t small = 0;
        for (int l = 1; l < g.live; ++l) small += (size_t)(g.D >> (l + 1)) * (g.D >> (l + 1)) * PP * 4;
        size_t low = l_taps > head_fwd_lds(g) ? l_taps : head_fwd_lds(g);       // scratch of the phases that run while C_1.. live
        low = (low + 255) / 256 * 256;
        size_t base0 = lds > low + small ? lds : low + small;
        base0 = (base0 + 255) / 256 * 256;
        fa.c0_off = -1; fa.c1_off = 0;
        if (base0 + c0 <= 160 * 1024 - 512) {
            fa.c0_off = (int)base0;
            fa.c1_off = (int)low;
            lds = base0 + c0;
        }
    }
    {   // the inner branch (and the key placement) must fit into the T planes, below the embedding tile
        const size_t t_off = (size_t)4 * PP * PP * 4, t_bytes = (size_t)2 * g.F * (S * PP + 16) * 4;
        size_t need = inner_fwd_lds(g);
        if (need < (size_t)8 * RANK_MAXF * 4) need = (size_t)8 * RANK_MAXF * 4;
        fa.early_off = (need <= t_bytes && ids != nullptr) ? (int)t_off : 0;
        fa.es_off = (int)(t_off + t_bytes);
    }
    int rc = 0;
    DISPATCH_NT4(PP / 16, rc = (launch_fwd_all<NT_>(fa, lds, st)));
    return rc;
}
