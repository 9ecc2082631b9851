/*
 * cffm_hip.h - C ABI of libcffm_hip.so: the CFFM convolutional feature-interaction hot path on
 * MI355X (gfx950).
 *
 * The reference (Anony-CFFM/CFFM) has no FFI: its boundary between host loop and tensor runtime is
 * the pair of TensorFlow session calls
 *     loss, opt = sess.run((self.loss, self.optimizer), feed_dict)      CFFM.py:200   (train step)
 *     batch_out = sess.run((self.out), feed_dict)                       CFFM.py:596   (predict)
 * over the graph built by create_inference_convolutional_feature_interaction_FM (CFFM.py:296-453),
 * create_loss (CFFM.py:486-514) and create_optimizer (CFFM.py:517-529).  The entry points below are
 * what a binding for that seam would call instead; each one cites the graph lines it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed from the caller (torch tensors in this repo) unless
 *     the name ends in _host; nothing is allocated or freed inside the library;
 *   - every call is asynchronous on the given hipStream_t (passed as void*; NULL = default stream);
 *   - return value: 0 on success, otherwise a hipError_t (or CFFM_ERR_*) - cffm_error_string() gives
 *     the text; no global mutable state, safe from one host thread per device;
 *   - all arithmetic is fp32, ids are int32 (CFFM.py:232-235).
 */
#ifndef CFFM_HIP_H
#define CFFM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFFM_ABI_VERSION 9
#define CFFM_MAX_LAYERS 8          /* live conv layers = log2(D) - 1 <= 8  (D <= 512)          */
#define CFFM_MAX_FIELDS 64         /* linear-attention softmax runs inside one 64-lane wavefront */
#define CFFM_HEAD_UNITS 32         /* tf.layers.dense(units=32), CFFM.py:409                    */
#define CFFM_NSLAB 64              /* split-K partial slabs of every dense-gradient producer     */

#define CFFM_ERR_BAD_SHAPE 10001
#define CFFM_ERR_UNSUPPORTED 10002

/* activation ids, CFFM.py:132-141 */
enum { CFFM_ACT_RELU = 0, CFFM_ACT_PRELU = 1, CFFM_ACT_ELU = 2, CFFM_ACT_SELU = 3, CFFM_ACT_GELU = 4 };
/* loss ids, CFFM.py:486-514 (square_loss with lamda == 0 is the README default) */
enum { CFFM_LOSS_SQUARE_RMSE = 0, CFFM_LOSS_MSE = 1, CFFM_LOSS_MAE = 2, CFFM_LOSS_LOG = 3,
       CFFM_LOSS_SQUARE_L2 = 4 /* square_loss with lamda > 0: l2_loss + table regularisers, CFFM.py:489-491 */,
       CFFM_LOSS_HYBRID = 5 /* 0.5*l2_loss(y-out) + 0.5*log_loss(out, y) on the RAW out (NaN once out+1e-7 <= 0), CFFM.py:510-513 */ };

/* optimizer ids, CFFM.py:517-529 */
enum { CFFM_OPT_ADAGRAD = 0, CFFM_OPT_SGD = 1, CFFM_OPT_MOMENTUM = 2, CFFM_OPT_ADAM = 3 };

typedef struct cffm_shape {
    int32_t M;            /* features_M                                   CFFM.py:110            */
    int32_t F;            /* num_field                                    CFFM.py:119            */
    int32_t K;            /* inner_dims (even)                            CFFM.py:105            */
    int32_t D;            /* outer_dims (power of two >= 4)               CFFM.py:106            */
    int32_t act;          /* CFFM_ACT_*                                   CFFM.py:132-141        */
    int32_t linear_att;   /* 1: attention first-order term, 0: plain sum  CFFM.py:423-446        */
    int32_t inner_conv;   /* CFFM.py:301                                                         */
    int32_t outer_conv;   /* CFFM.py:348                                                         */
    int32_t loss;         /* CFFM_LOSS_*                                                         */
    float lamda_att;      /* CFFM.py:434                                                         */
    float beta_outer;     /* CFFM.py:414                                                         */
    float lr;             /* CFFM.py:523                                                         */
    float lamda;          /* lamda_bilinear, only used by CFFM_LOSS_SQUARE_L2    CFFM.py:489-491          */
    int32_t optimizer;    /* CFFM_OPT_*                                          CFFM.py:517-529          */
} cffm_shape_t;

/* Offsets (in floats) of the trained dense parameters inside ONE flat fp32 buffer "theta".  The same
 * layout is used for the Adagrad accumulators and for the gradient buffer.  Untrained variables of
 * the reference (outer_W/outer_b, CFFM.py:271-272; the dead last conv layer, CFFM.py:394-396) are not
 * part of theta - the host keeps them only for the '#params' log line. */
typedef struct cffm_theta_layout {
    int64_t n;                              /* total floats                                        */
    int64_t att_W, att_b;                   /* bias_W [F,F], bias_b [F]          CFFM.py:281-282   */
    int64_t bias;                           /* scalar                            CFFM.py:284       */
    int64_t inner_cw, inner_cb;             /* [2 taps][2 ch], [2]               CFFM.py:323       */
    int64_t inner_dw, inner_db;             /* dense [P*K], [1]                  CFFM.py:339       */
    int64_t conv_w[CFFM_MAX_LAYERS];        /* HWIO [2,2,P,P] stored [4][Pp][Pp], zero pads  CFFM.py:375-377 */
    int64_t conv_b[CFFM_MAX_LAYERS];        /* [Pp], zero pads                                     */
    int64_t d1_w, d1_b;                     /* [2D-2][32], [32]                  CFFM.py:409       */
    int64_t d2_w, d2_b;                     /* [32], [1]                         CFFM.py:410       */
    int64_t lin_w, lin_b;                   /* [F], [1]                          CFFM.py:441       */
    int32_t P, Pp, Lc, live;                /* pairs, pairs padded to 16, log2(D), Lc-1            */
} cffm_theta_layout_t;

/* Byte offsets of every intermediate inside the caller-provided workspace for a batch of B rows.
 * Exposed so that the parity tests can read each tensor back and compare it with the oracle. */
typedef struct cffm_ws_layout {
    int64_t bytes;
    int64_t Ei, Eo, fb;                     /* gathered rows [B,F,K] [B,F,D] [B,F]                 */
    int64_t inner_out;                      /* [B]                                                 */
    int64_t C[CFFM_MAX_LAYERS];             /* relu(conv_l + b_l)  [B,S_l,S_l,Pp], S_l = D>>(l+1)  */
    int64_t t1, h1, att, out;               /* [B,2D-2] [B,32] [B,F] [B]                           */
    int64_t sqerr;                          /* per-example loss term [B]                           */
    int64_t scalars;                        /* [0]=sum of loss terms (local) [1]=loss [2]=dscale [3]=sum used */
    int64_t dout, dt1;                      /* [B], [B,2D-2]                                       */
    int64_t dC[CFFM_MAX_LAYERS];            /* grad wrt C_l, same shape as C_l                     */
    int64_t dEi, dEo, dfb;                  /* IndexedSlices values [B,F,K] [B,F,D] [B,F]          */
    int64_t gpart;                          /* split-K partial gradients, per theta range [nslab][len] */
    int64_t gpart_floats;
    int64_t sort_keys, sort_vals;           /* int32 [B*F] each (sorted ids, source slots)         */
    int64_t sort_tmp;                       /* radix sort scratch                                  */
    int64_t sort_tmp_bytes;
    int64_t Gi, Go, Gfb;                    /* dense table gradients [M,K], [M,D], [M] (CFFM_LOSS_SQUARE_L2, CFFM_OPT_ADAM) */
    int64_t pool[CFFM_MAX_LAYERS];          /* wide shapes (Pp > 64): partial sum pools of act(C_l) left by the conv epilogues,
                                               [B][S_l][pool_np[l]] floats (0 = not used); the head adds the partials up       */
    int32_t pool_np[CFFM_MAX_LAYERS];       /* partials per (example, row y) of layer l                                          */
    int64_t w0pack;                         /* wide shapes with the tiled layer 0: the layer-0 filter re-laid out as MFMA operand
                                               fragments for the input-gradient kernel (rebuilt from theta by every backward pass;
                                               0 = not used)                                                                        */
    int64_t w0pack_floats;
    int64_t relu0;                          /* wide shapes with the tiled layer 0: one bit per element of C[0] .. C[live-2], set where
                                               C[l] > 0 (16-bit words, [B*S_l*S_l][Pp/16] per layer, one layer after the other), written
                                               by the forward of layer l and read by the input gradient of layer l+1 instead of C[l]
                                               itself (0 = not used)                                                               */
    int64_t wb3;                            /* wide shapes: scratch for ONE conv filter split into three bf16 pieces in the record order
                                               of the bf16x3 contraction ([k-step][piece][kk][padded column] x 16 bytes; rebuilt in front
                                               of every direct conv forward / input-gradient launch; 0 = not used)                     */
    int64_t wb3_bytes;
} cffm_ws_layout_t;

typedef struct cffm_tables {                /* the three gathered variables and nothing else       */
    float *inner_emb;                       /* [M,K]  inner_embeddings  CFFM.py:257                */
    float *outer_emb;                       /* [M,D]  outer_embeddings  CFFM.py:264                */
    float *feat_bias;                       /* [M]    feature_bias      CFFM.py:276                */
} cffm_tables_t;

/* ---- layout queries (host only, no GPU needed) -------------------------------------------------- */
int cffm_abi_version(void);
const char *cffm_error_string(int err);
int cffm_theta_layout(const cffm_shape_t *s, cffm_theta_layout_t *out);
int cffm_ws_layout(const cffm_shape_t *s, int32_t B, cffm_ws_layout_t *out);

/* ---- stage entry points ------------------------------------------------------------------------ */
/* tf.nn.embedding_lookup x3 (CFFM.py:303, :354, :422): ids int32 [B*F] -> Ei [B,F,K], Eo [B,F,D],
 * fb [B,F].  Any of the three outputs may be NULL to skip that table. */
int cffm_gather(const cffm_shape_t *s, const cffm_tables_t *t, const int32_t *ids, int32_t B,
                float *Ei, float *Eo, float *fb, void *stream);

/* The read-only form of the three lookups for WIDE shapes (F*(F-1)/2 > 64, K == D in {32, 64}, F <= 32: BASELINE configs[3]
 * and [4]): tf.nn.embedding_lookup x3 (CFFM.py:303, :354, :422) fused with every consumer of a whole looked-up example - the
 * inner branch (CFFM.py:304-343), the s0 sum pool of the outer-product map (CFFM.py:381) and the first-order inputs
 * (CFFM.py:422): ids [B,F] -> ws.inner_out [B], ws.t1[:, 0:D] (s0), ws.fb [B,F], ws.sort_keys.  The rows go HBM -> LDS ->
 * registers and are NOT written to ws.Ei / ws.Eo (cffm_predict / cffm_train_step / cffm_dp_local then fetch the rows again
 * from the tables in the few later kernels that need them).  This is the kernel the HBM-gather roofline is quoted on.
 * Returns CFFM_ERR_UNSUPPORTED for other shapes (cffm_gather_inner_fwd_ok(s) == 0): those run cffm_gather / the fused
 * small-shape forward. */
int cffm_gather_inner_fwd_ok(const cffm_shape_t *s);
int cffm_gather_inner_fwd(const cffm_shape_t *s, const cffm_tables_t *t, const float *theta, const int32_t *ids, int32_t B,
                          void *ws, void *stream);

/* inner branch CFFM.py:304-343 on gathered rows: ws.Ei -> ws.inner_out */
int cffm_inner_fwd(const cffm_shape_t *s, const float *theta, void *ws, int32_t B, void *stream);
/* its gradient: ws.dout, ws.Ei -> ws.dEi, gpart slabs of inner_cw/inner_cb/inner_dw/inner_db */
int cffm_inner_bwd(const cffm_shape_t *s, const float *theta, void *ws, int32_t B, void *stream);

/* STAGE CONTRACT FOR WIDE SHAPES (Pp = ceil16(F*(F-1)/2) > 64, every activation but gelu).  The forward stages leave two
 * side products in the workspace and the later stages CONSUME THOSE instead of re-reading the conv outputs:
 *   ws.pool[l]   row-sum partials of C[l] (CFFM.py:381, :390) written by the epilogue of cffm_outer_conv0_fwd (l = 0) and
 *                cffm_conv_fwd (l >= 1); cffm_head_fwd sums THESE and does not read ws.C[*];
 *   ws.relu0     relu bit masks of C[0] .. C[live-2], one 16-bit word per (pixel, 16 channels), written by the same
 *                epilogues; cffm_conv_bwd (l >= 1) reads the mask of C[l-1] in place of ws.C[l-1].
 * A caller that writes ws.C itself, or runs another forward on the same workspace between a forward stage and its consumer,
 * must re-run cffm_outer_conv0_fwd / cffm_conv_fwd for that layer first: stale pools / masks are not detected.  The narrow
 * shapes (Pp <= 64) and gelu keep the contract written at each entry point below (pools and masks are swept from ws.C). */
/* outer product fused with conv layer 0 (CFFM.py:355-367 + :385-386 for i = 0): ws.Eo -> ws.C[0] (wide: + ws.pool[0], ws.relu0) */
int cffm_outer_conv0_fwd(const cffm_shape_t *s, const float *theta, void *ws, int32_t B, void *stream);
/* ws.dC[0], ws.dt1, ws.Eo -> ws.dEo, gpart slabs of conv_w[0]/conv_b[0] */
int cffm_outer_conv0_bwd(const cffm_shape_t *s, const float *theta, void *ws, int32_t B, void *stream);

/* conv layer l >= 1 (CFFM.py:385-387): ws.C[l-1] -> ws.C[l] (wide: + ws.pool[l], and the relu mask of C[l] for l <= live-2).
 * Wide shapes (128-channel tiles, >= 32768 rows) contract on the bf16 MFMA pipe at fp32 accuracy (error-free 3-way bf16 split of both
 * operands, six cross terms, fp32 accumulate; ws.wb3 is their scratch); the environment variable CFFM_CONV_FP32=1 selects the fp32 MFMA
 * loops.  The same holds for cffm_conv_bwd (input gradient; its weight gradient at any row count). */
int cffm_conv_fwd(const cffm_shape_t *s, const float *theta, void *ws, int32_t B, int32_t layer, void *stream);
/* ws.dC[l], ws.C[l-1], ws.dt1 -> ws.dC[l-1], gpart slabs of conv_w[l]/conv_b[l].  Wide shapes: the input gradient takes the
 * relu mask of C[l-1] from ws.relu0 (see the stage contract above); the weight gradient still reads ws.C[l-1]. */
int cffm_conv_bwd(const cffm_shape_t *s, const float *theta, void *ws, int32_t B, int32_t layer, void *stream);

/* sum pooling + concat + dense heads + first-order term + add_n (CFFM.py:381, :390-396, :409-453):
 * ws.C[*], ws.Eo, ws.fb, ws.inner_out -> ws.t1, ws.h1, ws.att, ws.out; when y != NULL also ws.sqerr and
 * scalars[0] = sum of per-example loss terms over the B local rows.  Wide shapes: ws.pool[*] in place of ws.C[*] (see the
 * stage contract above). */
int cffm_head_fwd(const cffm_shape_t *s, const float *theta, void *ws, const float *y, int32_t B, void *stream);
/* loss gradient (CFFM.py:493) and the head's backward: ws.out, y, scalars[3] (the loss-term sum over
 * the GLOBAL batch of B_global rows) -> ws.dout, ws.dt1, ws.dfb, ws.dC[live-1], gpart slabs of the
 * head parameters, scalars[1] = loss. */
int cffm_head_bwd(const cffm_shape_t *s, const float *theta, void *ws, const float *y, int32_t B,
                  int64_t B_global, void *stream);

/* sum the split-K slabs of ws.gpart (and the finer x-slabs of conv layer 0) in slab order -> grad [theta.n];
 * bitwise reproducible.  B = the batch size the workspace was laid out for. */
int cffm_reduce_slabs(const cffm_shape_t *s, void *ws, int32_t B, float *grad, void *stream);

/* tf.train.AdagradOptimizer (CFFM.py:523-524), dense: acc += g*g; v -= lr*g/sqrt(acc) over n floats */
int cffm_dense_adagrad(float *theta, float *acc, const float *grad, int64_t n, float lr, void *stream);
/* sparse: ids int32 [n_rows] with row gradients dEi [n_rows,K], dEo [n_rows,D], dfb [n_rows]; duplicate
 * ids are summed FIRST (in slot order), then one update per distinct row; other rows untouched. */
int cffm_sparse_adagrad(const cffm_shape_t *s, const cffm_tables_t *tab, const cffm_tables_t *acc,
                        const int32_t *ids, int64_t n_rows, const float *dEi, const float *dEo,
                        const float *dfb, void *ws, int32_t B_ws, void *stream);

/* ---- composites: what CFFM.evaluate / CFFM.train call in place of the two sess.run()s ------------- */
/* sess.run(self.out) CFFM.py:596: ids [B,F] -> out [B] (also left in ws.out) */
int cffm_predict(const cffm_shape_t *s, const cffm_tables_t *tab, const float *theta, const int32_t *ids,
                 int32_t B, void *ws, float *out, void *stream);
/* forward half of a train step, through the per-example loss terms.  tab == NULL: the looked-up rows are already
 * staged in ws.Ei / ws.Eo / ws.fb (row-sharded tables: they arrived from their owner ranks) and ids is not read */
int cffm_forward(const cffm_shape_t *s, const cffm_tables_t *tab, const float *theta, const int32_t *ids,
                 const float *y, int32_t B, void *ws, void *stream);
/* backward half: needs scalars[3] = global loss-term sum (cffm_train_step copies scalars[0] there) */
int cffm_backward(const cffm_shape_t *s, const float *theta, const float *y, int32_t B, int64_t B_global,
                  void *ws, float *grad, void *stream);
/* cffm_forward + cffm_backward_unscaled in one call (what one rank runs before the two collectives of a
 * data-parallel step); uses the single-launch forward where the shape allows.  rows must hold B*F*(1+K+D+1 + 2) floats:
 * the packed rows [B*F][1+K+D+1] followed by this rank's B*F update keys (id << 32 | slot, 64-bit) in sorted order
 * (a "run"; valid when the single-launch forward ran, i.e. Pp <= 64 and B*F <= 4096) */
/* 1 if cffm_dp_local with this per-rank batch leaves a valid sorted run behind its rows (else pass n_runs = 0 rows) */
int cffm_dp_runs_ok(const cffm_shape_t *s, int32_t B);
int cffm_dp_local(const cffm_shape_t *s, const cffm_tables_t *tab, const float *theta, const int32_t *ids, const float *y,
                  int32_t B, int64_t B_global, void *ws, float *grad, float *rows, void *stream);
/* Dense-table exchange for small vocabularies (M*(K+D+1) floats fewer than all ranks' row gradients): the local half
 * leaves ONE buffer flat = [theta.n gradients | loss sum | pad | duplicates-summed table gradients as a dense
 * [M][K | D | 1] image] of cffm_dp_dense_floats(s) floats; the caller all-reduces (sums) it over the ranks and every
 * rank applies it with cffm_dp_apply_dense (rows nobody looked up carry 0 and stay bit-identical).  Needs the
 * single-launch forward (cffm_dp_runs_ok(s, B)).
 * CONTRACT of the image: all zeros on entry of cffm_dp_local_dense (which only writes the rows this rank looked up), all
 * zeros again on return of cffm_dp_apply_dense (it clears every element it reads).  The caller zeroes the buffer ONCE,
 * when it allocates it, and after any step it abandoned between the two calls. */
int64_t cffm_dp_dense_floats(const cffm_shape_t *s);
int cffm_dp_local_dense(const cffm_shape_t *s, const cffm_tables_t *tab, const float *theta, const int32_t *ids,
                        const float *y, int32_t B, int64_t B_global, void *ws, float *flat, void *stream);
int cffm_dp_apply_dense(const cffm_shape_t *s, const cffm_tables_t *tab, const cffm_tables_t *acc, float *theta,
                        float *theta_acc, float *flat_sum, int64_t B_global, float *loss_out, void *stream);
/* Data-parallel halves (cffm_amd/dist.py).  cffm_backward_unscaled = cffm_backward with dL/dout = (out - y) / B_global,
 * i.e. without the 1/L of the RMSE-style loss (CFFM.py:493), which needs the loss-term sum over the GLOBAL batch:
 * grad must hold theta.n + 4 floats, grad[theta.n] receives this rank's loss-term sum so that one all-reduce carries
 * gradients and sum; rows [B*F][1+K+D+1] receives (id bits | dEi | dEo | dfb) for one all-gather.  cffm_dp_apply with
 * n_runs = 0 takes rows [n_rows][1+K+D+1] in any order and sorts the keys itself; with n_runs > 0 rows is the
 * concatenation of n_runs cffm_dp_local buffers (n_rows / n_runs slots each): the sorted runs are merged by rank
 * (binary searches in LDS) instead of a radix sort of all n_rows keys.  cffm_dp_apply takes
 * the all-reduced grad and the all-gathered rows, applies 1/L and the dense + sparse Adagrad updates. */
int cffm_backward_unscaled(const cffm_shape_t *s, const float *theta, const int32_t *ids, const float *y, int32_t B,
                           int64_t B_global, void *ws, float *grad, float *rows, void *stream);
int cffm_dp_apply(const cffm_shape_t *s, const cffm_tables_t *tab, const cffm_tables_t *acc, float *theta,
                  float *theta_acc, const float *grad_sum, int64_t B_global, const float *rows, int64_t n_rows,
                  void *ws, int32_t B_ws, float *loss_out, int32_t n_runs, void *stream);
/* sess.run((self.loss, self.optimizer)) CFFM.py:200: one fused forward + backward + Adagrad update of
 * theta/tables (and their accumulators) in place; loss (device scalar, may be NULL) receives the loss. */
int cffm_train_step(const cffm_shape_t *s, const cffm_tables_t *tab, const cffm_tables_t *tab_acc,
                    float *theta, float *theta_acc, float *grad, const int32_t *ids, const float *y,
                    int32_t B, void *ws, float *loss, void *stream);

/* ---- row-sharded tables (cffm_amd/dist.py ShardedStep) -------------------------------------------------
 * The reference is single-device: these replace nothing in it; they are the device side of the three all-to-alls.
 * A looked-up row travels as ONE packed record of cffm_packed_row_floats(s) = K + D + 4 floats:
 * (inner row | outer row | feature_bias, 0, 0, 0), i.e. tf.nn.embedding_lookup x3 (CFFM.py:303, :354, :422) of one id. */
int32_t cffm_packed_row_floats(const cffm_shape_t *s);
/* owner side: rows int32 [n] (local row indices) -> out [n][K+D+4] */
int cffm_gather_packed(const cffm_shape_t *s, const cffm_tables_t *t, const int32_t *rows, int64_t n, float *out, void *stream);
/* requester side: slot i of the [B,F] batch takes record pos[i] of packed [n_records][K+D+4] (pos == NULL: record i) ->
 * ws.Ei / ws.Eo / ws.fb, ready for cffm_forward(tab = NULL) */
int cffm_stage_packed(const cffm_shape_t *s, const float *packed, const int32_t *pos, int64_t n_records, int32_t B, void *ws,
                      void *stream);
/* The same two halves WITHOUT staging, for the wide shapes (cffm_gather_inner_fwd_ok(s); CFFM_ERR_UNSUPPORTED otherwise - use
 * cffm_stage_packed + cffm_forward(tab = NULL) + cffm_backward_unscaled there): the records are consumed where they lie - slot i reads
 * record pos[i] in the kernel that fetches it (cffm_gather_inner_fwd_wide with a record stride), the later kernels that need a row again
 * read it from the records - so ws.Ei / ws.Eo are never written, exactly as cffm_predict / cffm_train_step run on replicated tables.
 * cffm_backward_unscaled_packed leaves ws.dEi / ws.dEo / ws.dfb for cffm_pack_rows_dedup and this rank's loss-term sum in grad[theta.n]. */
int cffm_forward_packed(const cffm_shape_t *s, const float *theta, const float *packed, const int32_t *pos, int64_t n_records,
                        const float *y, int32_t B, void *ws, void *stream);
int cffm_backward_unscaled_packed(const cffm_shape_t *s, const float *theta, const float *packed, const int32_t *pos,
                                  int64_t n_records, const float *y, int32_t B, int64_t B_global, void *ws, float *grad, void *stream);
/* gradient message with the duplicates of one id summed first, in slot order: order int32 [B*F] = slot at sorted position q
 * (sorted by destination, stable), uniq int32 [B*F] = index of the distinct id at position q (non-decreasing), local_ids
 * int32 [B*F] = the owner's row of every slot; ws.dEi / ws.dEo / ws.dfb -> out [#distinct][1+K+D+1] = (row bits | dEi |
 * dEo | dfb), the row format cffm_dp_apply takes */
int cffm_pack_rows_dedup(const cffm_shape_t *s, const int32_t *local_ids, const int32_t *order, const int32_t *uniq, int32_t B,
                         void *ws, float *out, void *stream);

/* routing plan of one [B,F] batch (n = B*F slots) through tables of M global rows sharded r -> (rank r % world, local row r / world);
 * a function of the ids alone, so ShardedStep issues it one step ahead.  All outputs int32 [n] except counts int64 [world]:
 *   local_ids  owner's local row of every slot                       order  slot at sorted position q ((owner, local row) order,
 *   uniq       index of the distinct (owner, local row) pair at q           stable: slots ascend inside a run of equal ids)
 *   pos        slot -> index of its distinct pair                    send_rows  local row of distinct pair u (first #distinct used)
 *   counts     distinct pairs per owner = the split sizes of the three all-to-alls (summed: #distinct)
 * = what cffm_pack_rows_dedup / cffm_stage_packed take.  scratch: cffm_shard_plan_scratch_bytes(n) bytes.  ids must lie in [0, M). */
int64_t cffm_shard_plan_scratch_bytes(int64_t n);
int cffm_shard_plan(const int32_t *ids, int64_t n, int32_t world, int64_t M, void *scratch, int32_t *local_ids, int32_t *order,
                    int32_t *uniq, int32_t *pos, int32_t *send_rows, int64_t *counts, void *stream);

/* ---- evaluate() (CFFM.py:583-615) without a device-to-host copy of the predictions ------------------------------- */
/* clip + metric sums of CFFM.py:607-614 over n rows: p = min(max(pred, lo), hi) with lo/hi = min/max of the split's labels;
 * sums[0] += sum (y - p)^2, sums[1] += sum y, sums[2] += sum y^2, all float64, in a fixed order (bitwise reproducible).
 * sums is ACCUMULATED so that a split swept in several blocks of rows adds up (zero it first); scratch must hold
 * cffm_eval_scratch_bytes() bytes.  RMSE = sqrt(sums[0]/N), R2 = 1 - sums[0] / (sums[2] - sums[1]^2/N). */
int64_t cffm_eval_scratch_bytes(void);
int cffm_eval_sums(const float *pred, const float *y, int64_t n, float lo, float hi, void *scratch, double *sums,
                   void *stream);

/* ---- peak probes (bench.py prices the kernels against the data-sheet peaks AND these measured ones) ------------- */
/* float4 streaming copy src -> dst (bytes % 16 == 0): 2*bytes of HBM traffic per launch */
int cffm_probe_copy(const void *src, void *dst, int64_t bytes, void *stream);
/* read-only stream of `bytes` (the HBM READ roofline a gather is priced against); sink: >= 2048 floats, practically never written */
int cffm_probe_read(const void *src, void *sink, int64_t bytes, void *stream);
/* dependency-free fp32 MFMA loop over the whole chip; *flops receives the flop count of the launch */
int cffm_probe_mfma(float *out, int32_t iters, int64_t *flops, void *stream);
/* the same on the bf16 pipe (v_mfma_f32_16x16x32_bf16): the denominator of the bf16x3 conv loops */
int cffm_probe_mfma_bf16(float *out, int32_t iters, int64_t *flops, void *stream);

/* The same step for the other optimizers of CFFM.py:517-529 (s->optimizer): state1 = Momentum accumulators / Adam m,
 * state2 = Adam v (NULL otherwise), step = 1-based Adam time step.  Adagrad routes to cffm_train_step(state1). */
int cffm_train_step_opt(const cffm_shape_t *s, const cffm_tables_t *tab, const cffm_tables_t *tab_state1,
                        const cffm_tables_t *tab_state2, float *theta, float *theta_state1, float *theta_state2,
                        float *grad, const int32_t *ids, const float *y, int32_t B, void *ws, float *loss, int64_t step,
                        void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CFFM_HIP_H */
