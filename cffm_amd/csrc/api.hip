// Layout queries and the composite entry points (what CFFM.evaluate / CFFM.train call in place of the
// two sess.run()s, CFFM.py:200 and :596).
#include "internal.hpp"

#include <string.h>

extern "C" int cffm_abi_version(void) { return CFFM_ABI_VERSION; }

extern "C" const char* cffm_error_string(int err) {
    switch (err) {
        case 0: return "ok";
        case CFFM_ERR_BAD_SHAPE: return "cffm: bad shape / argument";
        case CFFM_ERR_UNSUPPORTED: return "cffm: unsupported configuration";
        default: return hipGetErrorString((hipError_t)err);
    }
}

extern "C" int cffm_theta_layout(const cffm_shape_t* s, cffm_theta_layout_t* out) {
    int rc = check_shape(s);
    if (rc) return rc;
    memset(out, 0, sizeof(*out));
    const Geo g = make_geo(s);
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t r = o; o += (n + 3) / 4 * 4; return r; };   // 16-byte aligned members
    out->att_W = take((int64_t)g.F * g.F);
    out->att_b = take(g.F);
    out->bias = take(1);
    out->inner_cw = take(4);
    out->inner_cb = take(2);
    out->inner_dw = take((int64_t)g.P * g.K);
    out->inner_db = take(1);
    for (int l = 0; l < g.live; ++l) {
        out->conv_w[l] = take((int64_t)4 * g.Pp * g.Pp);     // [tap][Pp][Pp], zero outside [P][P]
        out->conv_b[l] = take(g.Pp);
    }
    out->d1_w = take((int64_t)(2 * g.D - 2) * CFFM_HEAD_UNITS);
    out->d1_b = take(CFFM_HEAD_UNITS);
    out->d2_w = take(CFFM_HEAD_UNITS);
    out->d2_b = take(1);
    out->lin_w = take(g.F);
    out->lin_b = take(1);
    out->n = o;
    out->P = g.P; out->Pp = g.Pp; out->Lc = g.Lc; out->live = g.live;
    return 0;
}

extern "C" int cffm_ws_layout(const cffm_shape_t* s, int32_t B, cffm_ws_layout_t* out) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (B < 1) return CFFM_ERR_BAD_SHAPE;
    memset(out, 0, sizeof(*out));
    const Geo g = make_geo(s);
    cffm_theta_layout_t tl;
    cffm_theta_layout(s, &tl);
    int64_t o = 0;
    auto take = [&](int64_t bytes) { int64_t r = o; o += (bytes + 255) / 256 * 256; return r; };
    const int64_t b = B;
    SlabPlan sp;
    make_slab_plan(s, B, tl, &sp);
    out->gpart = take(sp.total * 4);                             // first: offset 0
    out->gpart_floats = sp.total;
    out->scalars = take(16 * 4);
    out->Ei = take(b * g.F * g.K * 4);
    out->Eo = take(b * g.F * g.D * 4);
    out->fb = take(b * g.F * 4);
    out->inner_out = take(b * 4);
    for (int l = 0; l < g.live; ++l) {
        const int64_t S = g.D >> (l + 1);
        out->C[l] = take(b * S * S * g.Pp * 4);
    }
    out->t1 = take(b * (2 * g.D - 2) * 4);
    out->h1 = take(b * CFFM_HEAD_UNITS * 4);
    out->att = take(b * g.F * 4);
    out->out = take(b * 4);
    out->sqerr = take(b * 4);
    out->dout = take(b * 4);
    out->dt1 = take(b * (2 * g.D - 2) * 4);
    for (int l = 0; l < g.live; ++l) {
        const int64_t S = g.D >> (l + 1);
        out->dC[l] = take(b * S * S * g.Pp * 4);
    }
    out->dEi = take(b * g.F * g.K * 4);
    out->dEo = take(b * g.F * g.D * 4);
    out->dfb = take(b * g.F * 4);
    const int64_t nrows = b * g.F;
    out->sort_keys = take(nrows * 8);                            // packed (id << 32 | slot), unsorted
    out->sort_vals = take(nrows * 8);                            // the same keys, sorted
    out->sort_tmp_bytes = nrows * 32 + (4 << 20);
    out->sort_tmp = take(out->sort_tmp_bytes);
    if (s->loss == CFFM_LOSS_SQUARE_L2 || s->optimizer == CFFM_OPT_ADAM) {   // dense table gradients
        out->Gi = take((int64_t)s->M * s->K * 4);
        out->Go = take((int64_t)s->M * s->D * 4);
        out->Gfb = take((int64_t)s->M * 4);
    }
    if (s->outer_conv) {
        for (int l = 0; l < g.live; ++l) {
            const int np = pool_partials(g, l);
            out->pool_np[l] = np;
            if (np > 0) out->pool[l] = take(b * (g.D >> (l + 1)) * np * 4);
        }
        if (conv0_tile_dgrad2_ok(g)) {                           // WA + WE of conv0_fact_tile_dgrad2_kernel (conv.hip)
            out->w0pack_floats = 2 * (int64_t)(2 * g.F) * (g.Pp / 16) * 1024;
            out->w0pack = take(out->w0pack_floats * 4);
            if (g.live > 1 && g.act != CFFM_ACT_GELU)        // relu mask of C[0] (gelu's derivative needs the value)
                out->relu0 = take(relu_mask_off(g, b, g.live - 1));       // masks of C_0 .. C_{live-2}
        }
        if (g.Pp > 64 && g.live > 1) {                           // the bf16x3 loops of the direct layers: one pre-split filter image
            out->wb3_bytes = cffm_wb3_bytes(g.Pp);
            out->wb3 = take(out->wb3_bytes);
        }
    }
    out->bytes = o;
    return 0;
}

// no_materialise: the composites that own the whole step (cffm_train_step, cffm_predict, cffm_dp_local) let the wide shapes
// consume the looked-up rows in the kernel that fetches them (cffm_gather_inner_fwd_wide) and re-fetch them from the tables
// where a later kernel needs them (backward_impl with the same tab / ids); ws.Ei / ws.Eo are then never written.
// cffm_forward keeps materialising: its callers (the stage-by-stage parity tests, cffm_backward, the row-sharded step) read
// ws.Ei / ws.Eo afterwards.
static bool wide_rows(const cffm_shape_t* s, const cffm_tables_t* tab, const int32_t* ids) {
    return tab && ids && cffm_wide_regather_ok(s);
}

// rstride / rrows > 0 (wide shapes only): tab is a view into packed records of rstride floats, rrows of them (see RowSrc)
static int forward_impl(const cffm_shape_t* s, const cffm_tables_t* tab, const float* theta, const int32_t* ids,
                        const float* y, int32_t B, void* ws, bool fused_step, hipStream_t stream, bool no_materialise = false,
                        int rstride = 0, int rrows = 0) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (B <= 0) return 0;
    cffm_ws_layout_t wl;
    cffm_ws_layout(s, B, &wl);
    char* w = (char*)ws;
    const Geo g = make_geo(s);
    if (no_materialise && wide_rows(s, tab, ids)) {
        if ((rc = cffm_gather_inner_fwd_wide(s, tab, theta, ids, B, ws, stream, rstride, rrows))) return rc;
        const RowSrc ro = {tab->outer_emb, ids, rrows > 0 ? rrows : s->M, rstride};
        if ((rc = cffm_outer_conv0_fwd_rows(s, theta, ws, B, &ro, stream))) return rc;
        for (int l = 1; l < g.live; ++l)
            if ((rc = cffm_conv_fwd(s, theta, ws, B, l, stream))) return rc;
        return cffm_head_fwd_impl2(s, theta, ws, y, B, !fused_step, true, stream);
    }
    if (!tab) {
        // row-sharded tables: the rows came in over the wire and are already staged in ws.Ei / ws.Eo / ws.fb
        if (fused_step) return CFFM_ERR_UNSUPPORTED;
        if ((rc = cffm_inner_fwd(s, theta, ws, B, stream))) return rc;
    } else if (fused_step && s->inner_conv && s->outer_conv) {
        // the inner-branch kernel gathers the rows of its example itself (one launch less)
        if ((rc = cffm_inner_fwd_impl(s, theta, ws, B, tab, ids, stream))) return rc;
    } else {
        rc = cffm_gather_impl(s, tab, ids, B, s->inner_conv ? (float*)(w + wl.Ei) : nullptr,
                              s->outer_conv ? (float*)(w + wl.Eo) : nullptr, (float*)(w + wl.fb),
                              fused_step ? (unsigned long long*)(w + wl.sort_keys) : nullptr, stream);
        if (rc) return rc;
        if ((rc = cffm_inner_fwd(s, theta, ws, B, stream))) return rc;
    }
    if (s->outer_conv) {
        if ((rc = cffm_outer_conv0_fwd(s, theta, ws, B, stream))) return rc;
        for (int l = 1; l < g.live; ++l)
            if ((rc = cffm_conv_fwd(s, theta, ws, B, l, stream))) return rc;
    }
    return cffm_head_fwd_impl(s, theta, ws, y, B, !fused_step, stream);
}

extern "C" int cffm_forward(const cffm_shape_t* s, const cffm_tables_t* tab, const float* theta, const int32_t* ids,
                            const float* y, int32_t B, void* ws, void* stream) {
    return forward_impl(s, tab, theta, ids, y, B, ws, false, (hipStream_t)stream);
}

extern "C" int cffm_predict(const cffm_shape_t* s, const cffm_tables_t* tab, const float* theta, const int32_t* ids,
                            int32_t B, void* ws, float* out, void* stream) {
    int rc = forward_impl(s, tab, theta, ids, nullptr, B, ws, false, (hipStream_t)stream, true);
    if (rc || B <= 0) return rc;
    if (out) {
        cffm_ws_layout_t wl;
        cffm_ws_layout(s, B, &wl);
        hipError_t e = hipMemcpyAsync(out, (char*)ws + wl.out, (size_t)B * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

// backward through the slab reduction; fused = single-GPU step (local loss sum, Adagrad folded into the reduction)
static int backward_impl(const cffm_shape_t* s, float* theta, float* theta_acc, const float* y, int32_t B,
                         int64_t B_global, void* ws, float* grad, bool fused, float* loss_out, hipStream_t stream,
                         bool unscaled = false, bool skip_reduce = false, const int32_t* rank_ids = nullptr,
                         const cffm_tables_t* rtab = nullptr, const int32_t* rids = nullptr, int rstride = 0, int rrows = 0) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (B <= 0) return 0;
    cffm_ws_layout_t wl; cffm_theta_layout_t tl;
    cffm_ws_layout(s, B, &wl); cffm_theta_layout(s, &tl);
    const Geo g = make_geo(s);
    char* w = (char*)ws;
    if (!s->inner_conv || !s->outer_conv || !s->linear_att) {   // slabs of a disabled branch must read as zeros
        hipError_t e = hipMemsetAsync(w + wl.gpart, 0, (size_t)wl.gpart_floats * 4, stream);
        if (e != hipSuccess) return (int)e;
    }
    bool inner_done = false;
    const bool wide = wide_rows(s, rtab, rids);      // the forward did not materialise Ei / Eo: rows come from the tables (RowSrc)
    if (bwd_top_ok(s, B) && s->loss != CFFM_LOSS_SQUARE_L2) {
        // head + top two conv layers + inner branch: one launch
        int next = 0;
        if ((rc = cffm_bwd_top_impl(s, theta, ws, y, B, B_global, fused || loss_out != nullptr, loss_out, unscaled, stream, &next,
                                    rank_ids)))
            return rc;
        inner_done = true;
        if (bwd_fused01_ok(s, B)) {            // layers 3..0 below the fused top: one launch
            if ((rc = cffm_conv01_bwd_impl(s, theta, ws, B, stream))) return rc;
            next = 0;
        }
        for (int l = next; l >= 1; --l) {
            if (l == next && top_wgrad_deferred(s, B)) rc = cffm_conv_bwd_below_top(s, theta, ws, B, l, stream);
            else rc = cffm_conv_bwd(s, theta, ws, B, l, stream);
            if (rc) return rc;
        }
        if (!bwd_fused01_ok(s, B) && (rc = cffm_outer_conv0_bwd(s, theta, ws, B, stream))) return rc;
    } else {
        if ((rc = cffm_head_bwd_impl(s, theta, ws, y, B, B_global, fused || loss_out != nullptr, loss_out, stream, unscaled))) return rc;
        if (s->outer_conv) {
            for (int l = g.live - 1; l >= 1; --l) {
                if (l == g.live - 1) rc = cffm_conv_bwd_with_inner(s, theta, ws, B, l, stream, &inner_done);
                else rc = cffm_conv_bwd(s, theta, ws, B, l, stream);
                if (rc) return rc;
            }
            if (wide) {
                const RowSrc ro = {rtab->outer_emb, rids, rrows > 0 ? rrows : s->M, rstride};
                rc = cffm_outer_conv0_bwd_rows(s, theta, ws, B, &ro, stream);
            } else {
                rc = cffm_outer_conv0_bwd(s, theta, ws, B, stream);
            }
            if (rc) return rc;
        }
    }
    if (!inner_done) {
        if (wide) {
            const RowSrc ri = {rtab->inner_emb, rids, rrows > 0 ? rrows : s->M, rstride};
            rc = cffm_inner_bwd_rows(s, theta, ws, B, &ri, stream);
        } else {
            rc = cffm_inner_bwd(s, theta, ws, B, stream);
        }
        if (rc) return rc;
    }
    if (skip_reduce) return 0;                  // the caller reduces the slabs together with the table update
    return cffm_reduce_slabs_impl(s, ws, B, grad, fused ? theta : nullptr, fused ? theta_acc : nullptr, s->lr, stream);
}

extern "C" int cffm_backward(const cffm_shape_t* s, const float* theta, const float* y, int32_t B, int64_t B_global,
                             void* ws, float* grad, void* stream) {
    return backward_impl(s, const_cast<float*>(theta), nullptr, y, B, B_global, ws, grad, false, nullptr,
                         (hipStream_t)stream);
}

// Data-parallel backward: dL/dout = (out - y) / B_global WITHOUT the 1/L of the RMSE-style loss; grad must have room
// for theta.n + 4 floats - element theta.n receives this rank's loss-term sum so that ONE all-reduce carries both.
// cffm_dp_apply then applies 1/L to the summed gradients.  The packed rows for the all-gather are written to `rows`
// [B*F][1 + K + D + 1] = (id bits | dEi | dEo | dfb).
extern "C" int cffm_backward_unscaled(const cffm_shape_t* s, const float* theta, const int32_t* ids, const float* y,
                                      int32_t B, int64_t B_global, void* ws, float* grad, float* rows, void* stream) {
    if (s && (s->loss == CFFM_LOSS_HYBRID || s->loss == CFFM_LOSS_SQUARE_L2)) return CFFM_ERR_UNSUPPORTED;   // single-process only
    int rc = backward_impl(s, const_cast<float*>(theta), nullptr, y, B, B_global, ws, grad, false, nullptr,
                           (hipStream_t)stream, true);
    if (rc || B <= 0) return rc;
    cffm_ws_layout_t wl; cffm_theta_layout_t tl;
    cffm_ws_layout(s, B, &wl); cffm_theta_layout(s, &tl);
    char* w = (char*)ws;
    if (!rows) {        // the caller packs the row gradients itself (cffm_pack_rows_dedup): only the loss-term sum is moved
        hipError_t e = hipMemcpyAsync(grad + tl.n, w + wl.scalars, sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
        return e == hipSuccess ? 0 : (int)e;
    }
    return cffm_pack_rows(s, ids, B, s->inner_conv ? (const float*)(w + wl.dEi) : nullptr,
                          s->outer_conv ? (const float*)(w + wl.dEo) : nullptr, (const float*)(w + wl.dfb),
                          (const float*)(w + wl.scalars), grad + tl.n, rows, (hipStream_t)stream);
}

// ---- row-sharded step without staging (cffm_amd/dist.py ShardedStep): the packed records a rank received ARE the tables ------
static bool packed_view(const cffm_shape_t* s, const float* packed, int64_t n_records, cffm_tables_t* view) {
    if (check_shape(s) || !packed || n_records <= 0 || n_records >= (1ll << 31) || !cffm_wide_regather_ok(s)) return false;
    view->inner_emb = const_cast<float*>(packed);
    view->outer_emb = const_cast<float*>(packed) + s->K;
    view->feat_bias = const_cast<float*>(packed) + s->K + s->D;
    return true;
}
extern "C" int cffm_forward_packed(const cffm_shape_t* s, const float* theta, const float* packed, const int32_t* pos,
                                   int64_t n_records, const float* y, int32_t B, void* ws, void* stream) {
    cffm_tables_t view;
    if (B <= 0) return check_shape(s);
    if (!pos || !packed_view(s, packed, n_records, &view)) return CFFM_ERR_UNSUPPORTED;
    return forward_impl(s, &view, theta, pos, y, B, ws, false, (hipStream_t)stream, true, s->K + s->D + 4, (int)n_records);
}
extern "C" int cffm_backward_unscaled_packed(const cffm_shape_t* s, const float* theta, const float* packed, const int32_t* pos,
                                             int64_t n_records, const float* y, int32_t B, int64_t B_global, void* ws, float* grad,
                                             void* stream) {
    if (s && (s->loss == CFFM_LOSS_HYBRID || s->loss == CFFM_LOSS_SQUARE_L2)) return CFFM_ERR_UNSUPPORTED;   // single-process only
    cffm_tables_t view;
    if (B <= 0) return check_shape(s);
    if (!pos || !packed_view(s, packed, n_records, &view)) return CFFM_ERR_UNSUPPORTED;
    int rc = backward_impl(s, const_cast<float*>(theta), nullptr, y, B, B_global, ws, grad, false, nullptr, (hipStream_t)stream, true,
                           false, nullptr, &view, pos, s->K + s->D + 4, (int)n_records);
    if (rc) return rc;
    cffm_ws_layout_t wl; cffm_theta_layout_t tl;
    cffm_ws_layout(s, B, &wl); cffm_theta_layout(s, &tl);
    hipError_t e = hipMemcpyAsync(grad + tl.n, (char*)ws + wl.scalars, sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
}

// the key placement can leave the forward launch when the fused top of the backward runs (and is not the L2 loss path)
static bool defer_rank(const cffm_shape_t* s, int32_t B) {
    return bwd_top_ok(s, B) && s->loss != CFFM_LOSS_SQUARE_L2 && cffm_fwd_all_ok(s, B);
}

extern "C" int cffm_dp_runs_ok(const cffm_shape_t* s, int32_t B) {
    return (check_shape(s) == 0 && B > 0 && cffm_fwd_all_ok(s, B)) ? 1 : 0;
}

// Local half of a data-parallel step in one call: forward (one launch at the README shapes), backward with
// dL/dout = (out - y) / B_global, then slab reduction ∥ row packing.  Same outputs as cffm_forward + cffm_backward_unscaled.
extern "C" int cffm_dp_local(const cffm_shape_t* s, const cffm_tables_t* tab, const float* theta, const int32_t* ids,
                             const float* y, int32_t B, int64_t B_global, void* ws, float* grad, float* rows, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int rc = check_shape(s);
    if (rc) return rc;
    if (B <= 0) return 0;
    if (!y || s->loss == CFFM_LOSS_HYBRID || s->loss == CFFM_LOSS_SQUARE_L2) return CFFM_ERR_UNSUPPORTED;
    const bool run = cffm_fwd_all_ok(s, B);          // false for a disabled branch: plain forward, no sorted run          // the single-launch forward also leaves this rank's keys sorted
    const bool later = run && defer_rank(s, B);
    if (run) rc = cffm_fwd_all_impl(s, tab, theta, ids, y, B, ws, st, !later);
    else rc = forward_impl(s, tab, theta, ids, y, B, ws, false, st, true);
    if (rc) return rc;
    if ((rc = backward_impl(s, const_cast<float*>(theta), nullptr, y, B, B_global, ws, grad, false, nullptr, st, true, true,
                            later ? ids : nullptr, run ? nullptr : tab, run ? nullptr : ids))) return rc;
    return cffm_dp_tail(s, ids, B, ws, grad, rows, run, st);
}

// Same local half for the dense-table exchange (small vocabularies): flat = [theta gradients | loss sum | table gradient
// image], cffm_dp_dense_floats(s) floats, to be summed over the ranks by ONE all-reduce and handed to cffm_dp_apply_dense.
extern "C" int cffm_dp_local_dense(const cffm_shape_t* s, const cffm_tables_t* tab, const float* theta, const int32_t* ids,
                                   const float* y, int32_t B, int64_t B_global, void* ws, float* flat, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int rc = check_shape(s);
    if (rc) return rc;
    if (B <= 0) return 0;
    if (!y || s->loss == CFFM_LOSS_HYBRID || s->loss == CFFM_LOSS_SQUARE_L2 || !cffm_fwd_all_ok(s, B)) return CFFM_ERR_UNSUPPORTED;
    cffm_theta_layout_t tl;
    cffm_theta_layout(s, &tl);
    // The table image of `flat` must be all zeros on entry: cffm_dp_apply_dense leaves it that way (zero on exit), so only
    // the very first step needs a cleared buffer (the round-2 code paid a hipMemsetAsync of the 1.4 MB image, 4.5 us in front of
    // the forward launch, on every step).
    const bool later = defer_rank(s, B);
    if ((rc = cffm_fwd_all_impl(s, tab, theta, ids, y, B, ws, st, !later))) return rc;
    if ((rc = backward_impl(s, const_cast<float*>(theta), nullptr, y, B, B_global, ws, flat, false, nullptr, st, true, true,
                            later ? ids : nullptr))) return rc;
    return cffm_dp_tail_dense(s, B, ws, flat, st);
}

extern "C" int cffm_train_step(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* tab_acc,
                               float* theta, float* theta_acc, float* grad, const int32_t* ids, const float* y,
                               int32_t B, void* ws, float* loss, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int rc = check_shape(s);
    if (rc) return rc;
    if (B <= 0) return 0;
    cffm_ws_layout_t wl;
    cffm_ws_layout(s, B, &wl);
    char* w = (char*)ws;
    if (s->loss == CFFM_LOSS_SQUARE_L2) {       // regularised square loss: dense table gradients and updates
        // the reference cannot build this graph with a disabled branch either: create_loss reads self.weights['inner_embeddings']
        // and ['outer_embeddings'] (CFFM.py:489-491), which initialize_variables only creates for an enabled branch (:255, :262)
        if (!s->inner_conv || !s->outer_conv) return CFFM_ERR_UNSUPPORTED;
        if ((rc = forward_impl(s, tab, theta, ids, y, B, ws, true, st))) return rc;
        if ((rc = backward_impl(s, theta, theta_acc, y, B, (int64_t)B, ws, grad, true, loss, st))) return rc;
        return cffm_tables_adagrad_l2(s, tab, tab_acc, ids, (int64_t)B * s->F, ws, B, st);
    }
    if (cffm_fwd_all_ok(s, B)) {                 // small-channel shapes: the whole forward (and the key sort) in one launch
        const bool later = defer_rank(s, B);
        if ((rc = cffm_fwd_all_impl(s, tab, theta, ids, y, B, ws, st, !later))) return rc;
        if ((rc = backward_impl(s, theta, theta_acc, y, B, (int64_t)B, ws, grad, true, loss, st, false, true, later ? ids : nullptr)))
            return rc;
        return cffm_update_all(s, tab, tab_acc, theta, theta_acc, grad, ws, B, st);
    }
    rc = forward_impl(s, tab, theta, ids, y, B, ws, true, st, true);
    if (rc) return rc;
    if ((rc = backward_impl(s, theta, theta_acc, y, B, (int64_t)B, ws, grad, true, loss, st, false, false, nullptr, tab, ids))) return rc;
    return cffm_sparse_adagrad_impl(s, tab, tab_acc, ids, (int64_t)B * s->F,
                                    s->inner_conv ? (const float*)(w + wl.dEi) : nullptr,
                                    s->outer_conv ? (const float*)(w + wl.dEo) : nullptr, (const float*)(w + wl.dfb),
                                    ws, B, true, st);
}

extern "C" int cffm_train_step_opt(const cffm_shape_t* s, const cffm_tables_t* tab, const cffm_tables_t* tab_state1,
                                   const cffm_tables_t* tab_state2, float* theta, float* theta_state1, float* theta_state2,
                                   float* grad, const int32_t* ids, const float* y, int32_t B, void* ws, float* loss,
                                   int64_t step, void* stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    if (s->optimizer == CFFM_OPT_ADAGRAD)
        return cffm_train_step(s, tab, tab_state1, theta, theta_state1, grad, ids, y, B, ws, loss, stream);
    if (B <= 0) return 0;
    if (s->loss == CFFM_LOSS_SQUARE_L2 && (!s->inner_conv || !s->outer_conv)) return CFFM_ERR_UNSUPPORTED;   // as in cffm_train_step
    hipStream_t st = (hipStream_t)stream;
    const bool nm = s->loss != CFFM_LOSS_SQUARE_L2;      // the regularised loss sweeps the tables densely: keep its path as it was
    if ((rc = forward_impl(s, tab, theta, ids, y, B, ws, true, st, nm))) return rc;
    // gradients only (no fused Adagrad); the loss is written by head_bwd
    if ((rc = backward_impl(s, theta, nullptr, y, B, (int64_t)B, ws, grad, false, loss, st, false, false, nullptr,
                            nm ? tab : nullptr, nm ? ids : nullptr))) return rc;
    return cffm_apply_opt(s, tab, tab_state1, tab_state2, theta, theta_state1, theta_state2, grad, ids, (int64_t)B * s->F, ws,
                          B, step, st);
}
