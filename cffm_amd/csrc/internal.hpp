// Internal (non-ABI) variants used by the fused composites in api.hip.
#pragma once
#include "common.hpp"

// gather that also emits the sort keys of the sparse update: keys[slot] = (id << 32) | slot
int cffm_gather_impl(const cffm_shape_t* s, const cffm_tables_t* t, const int32_t* ids, int32_t B, float* Ei, float* Eo,
                     float* fb, unsigned long long* keys, hipStream_t st);
// do_sum = false skips the separate loss-sum launch (cffm_head_bwd_impl then sums the terms itself)
int cffm_head_fwd_impl(const cffm_shape_t* s, const float* theta, void* ws, const float* y, int32_t B, bool do_sum,
                       hipStream_t st);
int cffm_head_bwd_impl(const cffm_shape_t* s, const float* theta, void* ws, const float* y, int32_t B, int64_t B_global,
                       bool local_sum, float* loss_out, hipStream_t st, bool unscaled = false);
// theta != nullptr: the dense Adagrad update is fused into the slab reduction
int cffm_reduce_slabs_impl(const cffm_shape_t* s, void* ws, int32_t B, float* grad, float* theta, float* acc, float lr,
                           hipStream_t st);
int cffm_sparse_adagrad_impl(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* acc,
                             const int32_t* ids, int64_t n_rows, const float* dEi, const float* dEo, const float* dfb,
                             void* ws, int32_t B_ws, bool prepacked, hipStream_t st);
// tab != nullptr: fused step - the kernel gathers the rows of example b itself (all three tables) and leaves
// Ei/Eo/fb and the packed sort keys in the workspace, so no separate gather launch is needed
int cffm_inner_fwd_impl(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, const cffm_tables_t* tab,
                        const int32_t* ids, hipStream_t st);
// one half of a conv layer's backward: which & 1 = weight/bias gradient, which & 2 = input gradient
int cffm_conv_bwd_part(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, int32_t layer, int which,
                       hipStream_t st);
// fused single-GPU update (both tables branches on): slab reduction + dense Adagrad and the sorted sparse table update
// as two roles of one launch; the keys must already be sorted in ws.sort_vals
int cffm_update_all(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* tab_acc, float* theta,
                    float* theta_acc, float* grad, void* ws, int32_t B, hipStream_t st);
// fused top of the backward (bwd_top_ok(s, B)): head + top conv layers + inner branch in one launch
int cffm_bwd_top_impl(const cffm_shape_t* s, const float* theta, void* ws, const float* y, int32_t B, int64_t B_global,
                      bool local_sum, float* loss_out, bool unscaled, hipStream_t st, int* next_layer,
                      const int32_t* rank_ids = nullptr);
// backward of conv layer `layer`; where the paired launch is available it also carries the inner-branch backward
// (*inner_done = true), which the caller must then not launch again
int cffm_conv01_bwd_impl(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, hipStream_t st);
int cffm_conv_bwd_below_top(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, int32_t layer, hipStream_t st);
int cffm_conv_bwd_with_inner(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, int32_t layer, hipStream_t st,
                             bool* inner_done);
// the two halves of the sparse update: stable sort of the packed keys, then the segment-sum + Adagrad sweep
int cffm_sort_keys_impl(const cffm_shape_t* s, const int32_t* ids, int64_t n_rows, void* ws, int32_t B_ws, bool prepacked,
                        hipStream_t st, int64_t id_stride = 1);
struct LateScale;
int cffm_sparse_apply_strided(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* acc, int64_t n_rows,
                              const float* dEi, int64_t sEi, const float* dEo, int64_t sEo, const float* dfb, int64_t sfb,
                              void* ws, int32_t B_ws, LateScale ls, hipStream_t st);
int cffm_sparse_apply_impl(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* acc, int64_t n_rows,
                           const float* dEi, const float* dEo, const float* dfb, void* ws, int32_t B_ws, hipStream_t st);
// slab reduction (gradients only) and the packing of the rows + local loss sum, two roles of one launch
int cffm_dp_tail(const cffm_shape_t* s, const int32_t* ids, int32_t B, void* ws, float* grad, float* rows, bool with_run,
                 hipStream_t st);
// dense-table variant of cffm_dp_tail: slab reduction ∥ scatter of this rank's summed row gradients into flat
int cffm_dp_tail_dense(const cffm_shape_t* s, int32_t B, void* ws, float* flat, hipStream_t st);
// rows[slot] = (id bits | dEi | dEo | dfb) for the all-gather of the data-parallel step; also copies the local
// loss-term sum (scalars[0]) to *sum_dst
int cffm_pack_rows(const cffm_shape_t* s, const int32_t* ids, int32_t B, const float* dEi, const float* dEo, const float* dfb,
                   const float* scalars, float* sum_dst, float* rows, hipStream_t st);
// whole forward of the fused step in one launch (+ the key sort); only for shapes cffm_fwd_all_ok() accepts
bool cffm_fwd_all_ok(const cffm_shape_t* s, int32_t B);
int cffm_fwd_all_impl(const cffm_shape_t* s, const cffm_tables_t* tab, const float* theta, const int32_t* ids,
                      const float* y, int32_t B, void* ws, hipStream_t st, bool rank_keys = true);
// CFFM_LOSS_SQUARE_L2: tables updated densely with g = scatter(row grads) + lamda * w (feature_bias stays sparse)
int cffm_tables_adagrad_l2(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* acc, const int32_t* ids,
                           int64_t n_rows, void* ws, int32_t B_ws, hipStream_t st);
// SGD / Momentum / Adam updates of theta and the three tables from grad + the row gradients in ws (CFFM.py:519-529)
int cffm_apply_opt(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* st1, const cffm_tables_t* st2,
                   float* theta, float* th1, float* th2, const float* grad, const int32_t* ids, int64_t n_rows, void* ws,
                   int32_t B_ws, int64_t step, hipStream_t st);

// ---- wide shapes (Pp > 64): rows consumed where they are fetched, nothing materialised (RowSrc, common.hpp) ---------------
bool cffm_wide_regather_ok(const cffm_shape_t* s);
int64_t cffm_wb3_bytes(int Pp);     // bytes of the pre-split filter image of the bf16x3 conv loops (conv.hip), forward or input gradient
bool cffm_giw_lds_ok();     // the fused gather's LDS addressing assumption holds for every instance (inner.hip; checked on the host)
// tf.nn.embedding_lookup x3 fused with the inner branch, the s0 pool and the first-order inputs: ids -> ws.inner_out,
// ws.t1[:, 0:D] (s0), ws.fb, ws.sort_keys; Ei / Eo are NOT written
// tab_stride > 0: the three "tables" are views into ONE array of records of tab_stride floats (the packed rows a row-sharded rank
// received: tab->inner_emb = records, outer_emb = records + K, feat_bias = records + K + D) with tab_rows records; ids = slot -> record
int cffm_gather_inner_fwd_wide(const cffm_shape_t* s, const cffm_tables_t* tab, const float* theta, const int32_t* ids, int32_t B,
                               void* ws, hipStream_t st, int tab_stride = 0, int tab_rows = 0);
int cffm_outer_conv0_fwd_rows(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, const RowSrc* rs, hipStream_t st);
int cffm_outer_conv0_bwd_rows(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, const RowSrc* rs, hipStream_t st);
int cffm_inner_bwd_rows(const cffm_shape_t* s, const float* theta, void* ws, int32_t B, const RowSrc* rs, hipStream_t st);
// s0_ready: ws.t1[:, 0:D] already holds the s0 pool (the fused gather computed it): ws.Eo is not read
int cffm_head_fwd_impl2(const cffm_shape_t* s, const float* theta, void* ws, const float* y, int32_t B, bool do_sum, bool s0_ready,
                        hipStream_t st);
