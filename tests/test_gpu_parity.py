"""GPU parity: every stage of the HIP path, called through the C ABI, against the float64 oracle on
the same seeded inputs.  Tolerance is the north-star one (1e-5 relative fp32), applied element-wise against
1e-5 * (|ref| + rms(ref)) in the three tiers close() documents."""
import numpy as np
import pytest
import torch

from cffm_amd.spec import CFFMConfig, init_params
from oracle import cffm_oracle as orc
from oracle.parity import (REL_FLOOR, REL_TOL, TOL, WORST, adopt_device_kinks, check_backward_stages,  # noqa: F401
                           check_gather_inner_fwd_wide, close, dense_grad_slack, dout_slack, inner_kink_slack,
                           oracle_dense_grads, pad_channels, to64)

pytestmark = pytest.mark.gpu

CASES = {
    'tiny-relu': dict(M=60, F=4, K=8, D=8, act='relu', B=5),
    'tiny-d4-prelu': dict(M=40, F=3, K=4, D=4, act='prelu', B=9),
    'd16-gelu': dict(M=300, F=5, K=16, D=16, act='gelu', B=37),
    'mltag-elu': dict(M=2000, F=3, K=32, D=32, act='elu', B=64),
    'bookx-relu': dict(M=3000, F=6, K=32, D=32, act='relu', B=48),
    'frappe-selu': dict(M=5382, F=10, K=32, D=32, act='selu', B=256),
    # the other two README commands at their OWN batch size and vocabulary (README.md:20, :24 - BASELINE configs[0] and [2]):
    # every forward / backward stage and the full train step, not only forward rows
    'mltag-full': dict(M=90445, F=3, K=32, D=32, act='elu', B=1024),
    'bookx-full': dict(M=226336, F=6, K=32, D=32, act='relu', B=512),
    'f32-d64-relu': dict(M=4000, F=32, K=64, D=64, act='relu', B=3),
    'f12-d32-nolinatt': dict(M=500, F=12, K=16, D=32, act='selu', B=20, linear_att=0),
    # batch-size edges of the one-workgroup-per-example kernels: a single example, and B > 256 (workgroup 0 takes two)
    'b1-elu': dict(M=200, F=6, K=32, D=32, act='elu', B=1),
    'b257-relu': dict(M=900, F=6, K=32, D=32, act='relu', B=257),
    # wide filters (Pp > 64): the tiled factorised layer-0 kernels at other field counts / map sizes, and F = 33 where the
    # input-gradient kernel falls back to the direct contraction
    'f20-d64-elu': dict(M=2000, F=20, K=16, D=64, act='elu', B=2),
    'f33-d32-relu': dict(M=3000, F=33, K=8, D=32, act='relu', B=2),
    # wide filters with SEVERAL examples per gradient slab (the tiled layer-0 weight gradient walks b = slab, slab + 64, ...)
    # and the direct kernels of layers >= 1 at a realistic row count
    'f16-d32-b130': dict(M=3000, F=16, K=8, D=32, act='relu', B=130),
    'f20-d32-b300-selu': dict(M=5000, F=20, K=16, D=32, act='selu', B=300),
    'f32-d64-b160': dict(M=20000, F=32, K=64, D=64, act='relu', B=160, heavy=True),
    # wide filters with K == D: predict / train_step take the non-materialising route (cffm_gather_inner_fwd: rows consumed in
    # the kernel that fetches them, later kernels re-fetch from the tables) - the generic instances of that kernel at K = 32
    # and K = 64, a ragged last phase (B % 4 != 0), a half-filled fetch wave (F = 13: 416 pieces) and B < 4
    'f16-k32-d32-b70-selu': dict(M=3000, F=16, K=32, D=32, act='selu', B=70),
    'f13-k64-d64-b9-prelu': dict(M=900, F=13, K=64, D=64, act='prelu', B=9),
    'f20-k32-d32-b3-relu': dict(M=900, F=20, K=32, D=32, act='relu', B=3),
    # the F = 32, K = D = 64 instance of that kernel (circulant pair assignment, round 4) over several phases with a ragged last
    # one: its generic-activation build (selu) and its relu build
    'f32-k64-d64-b11-selu': dict(M=3000, F=32, K=64, D=64, act='selu', B=11),
    'f32-k64-d64-b9-relu': dict(M=3000, F=32, K=64, D=64, act='relu', B=9),
    # the bf16x3 instances of the direct conv layers (round 4: forward / input gradient of a layer with >= 32768 rows at NT = 8, the
    # weight gradient at any row count) with an activation that is NOT the identity on the stored relu output: selu goes through
    # the A-operand activation of gemm_tile_b3 / wgrad3's staging.  (gelu at this batch size fails on the top layer's dC under BOTH
    # conv loops - 50 of 262144 elements where relu(z) sits just outside the band adopt_device_kinks covers and gelu'(0+) = 0.5 makes
    # the decision an O(1) change - so the gelu arm is covered by the small gelu cases only.)
    'f16-k8-d32-b512-selu': dict(M=4000, F=16, K=8, D=32, act='selu', B=512, heavy=True),
    # more than 4,096 lookups per step: the rocPRIM radix sort + segment walk of the sparse update (the CLI's default
    # --batch_size 1024), with heavy duplication (ids drawn from 150 values per column)
    'frappe-b1024-dups': dict(M=5382, F=10, K=32, D=32, act='selu', B=1024, id_range=150),
    'bookx-b1024-dups': dict(M=226336, F=6, K=32, D=32, act='relu', B=1024, id_range=150),
    # disabled branches (CFFM.py:301, :348), with and without the attention first-order term
    'no-inner': dict(M=700, F=6, K=32, D=32, act='elu', B=33, inner_conv=0),
    'no-outer': dict(M=700, F=6, K=32, D=32, act='selu', B=33, outer_conv=0),
    'no-inner-no-outer': dict(M=700, F=6, K=32, D=32, act='relu', B=33, inner_conv=0, outer_conv=0),
    'no-inner-nolinatt': dict(M=700, F=10, K=32, D=32, act='relu', B=40, inner_conv=0, linear_att=0),
    'no-outer-nolinatt': dict(M=700, F=10, K=32, D=32, act='gelu', B=40, outer_conv=0, linear_att=0),
    'fm-only-nolinatt': dict(M=700, F=5, K=8, D=8, act='relu', B=300, inner_conv=0, outer_conv=0, linear_att=0),
    # conv01_bwd_kernel (D = 32, Pp <= 48, 64 <= B <= 256) outside the frappe instance: the generic instances for Pp = 48 / 32 /
    # 16, workgroups without an example (B < 256), a partial last slab of layers 2 / 3, the B = 64 edge and, one below it, the
    # five-launch path of the same shape
    'f10-d32-b100-elu': dict(M=900, F=10, K=16, D=32, act='elu', B=100),
    'f7-d32-b64-gelu': dict(M=600, F=7, K=8, D=32, act='gelu', B=64),
    'f5-d32-b200-relu': dict(M=400, F=5, K=32, D=32, act='relu', B=200),
    'f7-d32-b63-gelu': dict(M=600, F=7, K=8, D=32, act='gelu', B=63),
    # the fused top followed by conv_bwd_pair_kernel with the deferred top-layer weight gradients (top_wgrad_deferred): 256
    # examples at Pp = 64 (too wide for conv01_bwd_kernel) and at D = 64 (five conv layers)
    'f11-d32-b256-relu': dict(M=800, F=11, K=8, D=32, act='relu', B=256),
    'f6-d64-b256-elu': dict(M=500, F=6, K=8, D=64, act='elu', B=256),
}
HEAVY = [k for k, v in CASES.items() if v.get('heavy')]       # oracle needs several GB and ~a minute per pass
LIGHT = [k for k in CASES if k not in HEAVY]


def make_case(name, seed=0, trained_like=True):
    c = CASES[name]
    cfg = CFFMConfig(M=c['M'], F=c['F'], K=c['K'], D=c['D'], activation=c['act'], lamda_att=1.3,
                     linear_att=c.get('linear_att', 1), loss_type=c.get('loss', 'square_loss'),
                     inner_conv=c.get('inner_conv', 1), outer_conv=c.get('outer_conv', 1))
    p32 = init_params(cfg, seed=seed, dtype=np.float32)
    rng = np.random.default_rng(seed + 7)
    if trained_like:   # feature_bias is exactly 0 at init (CFFM.py:276): make the first-order term non-trivial
        p32['feature_bias'] = (rng.standard_normal(p32['feature_bias'].shape) * 0.3).astype(np.float32)
        p32['outer_embeddings'] = (p32['outer_embeddings'] * 20.0).astype(np.float32)
        p32['inner_embeddings'] = (p32['inner_embeddings'] * 4.0).astype(np.float32)
    X = rng.integers(0, c.get('id_range', cfg.M), size=(c['B'], cfg.F)).astype(np.int32)
    if 'id_range' in c:                                # spread the few values over the table (rows far apart)
        X = (X.astype(np.int64) * (cfg.M // c['id_range'])).astype(np.int32)
    X[0, 0] = X[-1, 0]
    if c['B'] > 2:
        X[1] = X[0]                                   # a fully duplicated row: duplicate ids in every column
    y = rng.choice([-1.0, 1.0], size=(c['B'],)).astype(np.float32)
    return cfg, p32, X, y


def engine_for(cfg, p32):
    from cffm_amd.engine import HipEngine
    return HipEngine(cfg, params=p32)


@pytest.mark.parametrize('name', LIGHT)
def test_forward_stages(name):
    cfg, p32, X, y = make_case(name)
    eng = engine_for(cfg, p32)
    B = X.shape[0]
    out_ref, c = orc.forward(to64(p32), X, cfg)
    ids = torch.from_numpy(X).cuda()
    yt = torch.from_numpy(y).cuda()
    eng.forward(ids, yt)
    torch.cuda.synchronize()
    Pp = eng.tl.Pp
    if cfg.inner_conv:
        np.testing.assert_array_equal(eng.ws_tensor(B, 'Ei', (B, cfg.F, cfg.K)).cpu().numpy(), p32['inner_embeddings'][X])
        close(eng.ws_tensor(B, 'inner_out', (B,)).cpu().numpy(), c['inner_out'], 'inner_out')
    np.testing.assert_array_equal(eng.ws_tensor(B, 'fb', (B, cfg.F)).cpu().numpy(), p32['feature_bias'][X][:, :, 0])
    if cfg.outer_conv:
        np.testing.assert_array_equal(eng.ws_tensor(B, 'Eo', (B, cfg.F, cfg.D)).cpu().numpy(), p32['outer_embeddings'][X])
        for l in range(cfg.live_layers):
            S = cfg.D >> (l + 1)
            got = eng.ws_tensor(B, 'C', (B, S, S, Pp), index=l).cpu().numpy()
            close(got, pad_channels(c['rs'][l], Pp), 'C[%d]' % l)
        close(eng.ws_tensor(B, 't1', (B, 2 * cfg.D - 2)).cpu().numpy(), c['t1'], 't1')
        close(eng.ws_tensor(B, 'h1', (B, 32)).cpu().numpy(), c['h1'], 'h1')
    if cfg.linear_att:
        close(eng.ws_tensor(B, 'att', (B, cfg.F)).cpu().numpy(), c['a'], 'att')
    close(eng.ws_tensor(B, 'out', (B,)).cpu().numpy(), out_ref, 'out')
    sc = eng.ws_tensor(B, 'scalars', (16,)).cpu().numpy()
    close(sc[0:1], [np.sum((y.astype(np.float64) - out_ref) ** 2)], 'sum of loss terms')
    # predict() is the same forward without labels
    close(eng.predict(ids).cpu().numpy(), out_ref, 'predict')


WIDE_NM = ['f32-d64-relu', 'f16-k32-d32-b70-selu', 'f13-k64-d64-b9-prelu', 'f20-k32-d32-b3-relu', 'f32-k64-d64-b11-selu',
           'f32-k64-d64-b9-relu']


@pytest.mark.parametrize('name', WIDE_NM)
def test_gather_inner_fwd_wide(name):
    """cffm_gather_inner_fwd on its own (oracle.parity.check_gather_inner_fwd_wide: rows never materialised, fb exact,
    inner_out / s0 / keys / the whole non-materialising predict against the oracle)."""
    cfg, p32, X, y = make_case(name)
    check_gather_inner_fwd_wide(cfg, p32, X, y, engine_for(cfg, p32))


@pytest.mark.parametrize('name', list(CASES))
def test_backward_stages(name):
    cfg, p32, X, y = make_case(name)
    check_backward_stages(cfg, p32, X, y, engine_for(cfg, p32), name=name, heavy=name in HEAVY)


TRAIN_CASES = ['tiny-relu', 'd16-gelu', 'bookx-relu', 'frappe-selu', 'mltag-full', 'bookx-full', 'f32-d64-relu', 'f12-d32-nolinatt', 'b1-elu',
               'b257-relu', 'f20-d64-elu', 'f33-d32-relu', 'f16-d32-b130', 'f20-d32-b300-selu', 'frappe-b1024-dups',
               'bookx-b1024-dups', 'no-inner', 'no-outer', 'no-inner-no-outer', 'no-inner-nolinatt', 'no-outer-nolinatt',
               'fm-only-nolinatt', 'f10-d32-b100-elu', 'f7-d32-b64-gelu', 'f5-d32-b200-relu', 'f7-d32-b63-gelu',
               'f11-d32-b256-relu', 'f6-d64-b256-elu', 'f16-k32-d32-b70-selu', 'f13-k64-d64-b9-prelu', 'f20-k32-d32-b3-relu',
               'f32-k64-d64-b11-selu']


@pytest.mark.parametrize('name,trained_like', [(n, t) for n in TRAIN_CASES for t in (True, False)] + [(n, True) for n in HEAVY])
def test_train_step_matches_oracle(name, trained_like):
    """One sess.run((loss, optimizer)): post-update parameters AND Adagrad accumulators of every
    variable, including the sparse (duplicates-summed-first) table updates."""
    cfg, p32, X, y = make_case(name, trained_like=trained_like)
    eng = engine_for(cfg, p32)
    step_check(cfg, eng, p32, None, X, y, name)


def step_check(cfg, eng, p32, acc0, X, y, name):
    """One train step of `eng` (whose state is p32 / acc0; acc0 = None: the initial accumulators) against one oracle step from
    the same state: loss, post-update parameters and accumulators within the gradient tolerance propagated through Adagrad."""
    p64 = to64(p32)
    B = X.shape[0]
    eng.forward(torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda())   # same forward the step will redo
    torch.cuda.synchronize()
    hook = (lambda cache: adopt_device_kinks(cfg, eng, B, cache)) if cfg.outer_conv else None
    # ONE oracle pass: gradients in dense form (for the tolerance propagation) and the step itself
    out, cache = orc.forward(p64, X, cfg)
    if hook is not None:
        hook(cache)
    L, dout = orc.loss_and_grad(out, y.astype(np.float64), cfg, p64)
    g = orc.backward(p64, cache, dout, cfg)
    ids_flat = X.reshape(-1)
    grads = {k: np.asarray(v) for k, v in g.items() if not k.startswith('d_') and not k.startswith('_')}
    dgs = {k: 1e-5 * (np.abs(v) + max(float(np.sqrt(np.mean(v * v))), 1e-30)) for k, v in grads.items()}
    slack = inner_kink_slack(p64, cache, dout, cfg)
    for k in ('inner_layer_conv_weight_0', 'inner_layer_conv_bias_0'):
        if k in slack:
            dgs[k] = dgs[k] + slack[k].reshape(dgs[k].shape)
    dsl = dout_slack(out, y, cfg, p64)              # the tolerance of `out` propagated through dL/dout (see the helpers)
    for k, v in dense_grad_slack(p64, cache, dsl, cfg).items():
        dgs[k] = dgs[k] + np.asarray(v).reshape(dgs[k].shape)
    rel_b = dsl / np.maximum(np.abs(dout), 1e-300)
    for tname, key in (('inner_embeddings', 'd_inner_rows'), ('outer_embeddings', 'd_outer_rows'), ('feature_bias', 'd_bias_rows')):
        if key in g:
            rows = g[key].reshape(ids_flat.shape[0], -1)
            t, ta = np.zeros(p64[tname].shape), np.zeros(p64[tname].shape)
            np.add.at(t, ids_flat, rows)
            # every looked-up row gradient is held to 1e-5 * (|row element| + rms of the row-gradient tensor) in
            # test_backward_stages; a table row's gradient is the SUM over its duplicates, so the bounds add up
            np.add.at(ta, ids_flat, np.abs(rows) + float(np.sqrt(np.mean(rows * rows))))
            grads[tname] = t
            dgs[tname] = 1e-5 * ta
            np.add.at(dgs[tname], ids_flat, (rel_b[:, None, None] * np.abs(g[key].reshape(B, cfg.F, -1))).reshape(ids_flat.shape[0], -1))
            if key in slack:
                np.add.at(dgs[tname], ids_flat, slack[key].reshape(ids_flat.shape[0], -1))
    del cache
    acc = orc.init_accumulators(p64)
    if acc0 is not None:
        for k, v in acc0.items():
            acc[k] = np.asarray(v, dtype=np.float64).reshape(np.shape(acc[k])).copy()
    acc_before = {k: np.array(v) for k, v in acc.items()}
    p_before = {k: np.array(v) for k, v in p64.items()}
    for k, gk in grads.items():             # Adagrad (CFFM.py:523-524) from the gradients above: same as orc.train_step
        if k in ('inner_embeddings', 'outer_embeddings', 'feature_bias'):
            continue
        gk = gk.reshape(np.shape(p64[k]))
        acc[k] = acc[k] + gk * gk
        p64[k] = p64[k] - cfg.lr * gk / np.sqrt(acc[k])
    for tname, key in (('inner_embeddings', 'd_inner_rows'), ('outer_embeddings', 'd_outer_rows'), ('feature_bias', 'd_bias_rows')):
        if key in g:
            orc.adagrad_sparse(p64[tname], acc[tname], ids_flat, g[key], cfg.lr)
    if name == 'tiny-relu' and acc0 is None:    # the hand-rolled update above IS orc.train_step (checked once, cheap case)
        q = {k: np.array(v) for k, v in p_before.items()}
        qa = orc.init_accumulators(q)
        orc.train_step(q, qa, X, y.astype(np.float64), cfg, cache_hook=hook)
        for k in q:
            np.testing.assert_array_equal(np.asarray(q[k]), np.asarray(p64[k]), err_msg=k)
            np.testing.assert_array_equal(np.asarray(qa[k]), np.asarray(acc[k]), err_msg=k)
    loss = eng.train_step(torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda())
    torch.cuda.synchronize()
    close(loss.cpu().numpy(), [L], 'loss')
    got, gacc = eng.export_params(), eng.export_accumulators()
    for k, v in got.items():
        # With acc0 = 1e-8 the first Adagrad step is u(g) = lr*g/sqrt(1e-8+g^2): for |g| <~ 1e-4 it amplifies a
        # gradient error dg by u'(g) = lr*1e-8/(1e-8+g^2)^1.5 (up to lr*1e4).  The gradients themselves are held
        # to 1e-5 of their tensor scale in test_backward_stages; here that error is propagated through u.
        extra = acc_extra = None
        if k in grads:
            gk = grads[k].reshape(v.shape)
            dg = dgs[k].reshape(v.shape)
            a0 = acc_before[k].reshape(v.shape)                       # 1e-8 at the first step (CFFM.py:524)
            u = lambda t: cfg.lr * t / np.sqrt(a0 + t * t)            # monotonic: the worst case sits at g +- dg
            extra = np.maximum(np.abs(u(gk + dg) - u(gk)), np.abs(u(gk - dg) - u(gk)))
            acc_extra = 2 * np.abs(gk) * dg + dg * dg
        close(v, p64[k].reshape(v.shape), 'param ' + k, tol=2e-5, extra=extra)
        if k in gacc:
            close(gacc[k], acc[k].reshape(v.shape), 'acc ' + k, tol=2e-5, extra=acc_extra)
    # untouched rows are bit-identical to the initial state
    touched = np.zeros(cfg.M, dtype=bool)
    touched[X.reshape(-1)] = True
    for k in ('inner_embeddings', 'outer_embeddings', 'feature_bias'):
        np.testing.assert_array_equal(got[k][~touched], p32[k][~touched])
        np.testing.assert_array_equal(gacc[k][~touched], acc_before[k].reshape(gacc[k].shape)[~touched].astype(np.float32))
    # a disabled branch leaves its table and every parameter of its own exactly as they were
    if not cfg.inner_conv:
        np.testing.assert_array_equal(got['inner_embeddings'], p32['inner_embeddings'])
        np.testing.assert_array_equal(got['dense_kernel'], p32['dense_kernel'])
    if not cfg.outer_conv:
        np.testing.assert_array_equal(got['outer_embeddings'], p32['outer_embeddings'])
        np.testing.assert_array_equal(got['dense_1_kernel'], p32['dense_1_kernel'])


def adagrad_step_slack(grads, acc_prev, lr, rel=1e-5):
    """Bound on the parameter change of ONE Adagrad step caused by a gradient error of rel * (|g| + rms(g)): the update
    u(g) = lr * g / sqrt(acc_prev + g^2) is monotonic in g, so the worst case sits at g +- dg.  With acc_prev = 1e-8
    (first touch, CFFM.py:524) u amplifies a gradient error near g = 0 by up to lr * 1e4."""
    out = {}
    for k, gk in grads.items():
        a = np.asarray(acc_prev[k], dtype=np.float64).reshape(gk.shape)
        dg = rel * (np.abs(gk) + max(float(np.sqrt(np.mean(gk * gk))), 1e-30))
        u = lambda t: lr * t / np.sqrt(a + t * t)
        out[k] = np.maximum(np.abs(u(gk + dg) - u(gk)), np.abs(u(gk - dg) - u(gk)))
    return out


def test_frappe_trajectory_of_30_steps_follows_the_oracle():
    """The reference runs thousands of steps (CFFM.py:186-200); this follows 30 of them at the headline shape (frappe command,
    batch 256, selu, Adagrad 0.05) on 30 synthetic batches.

    Adagrad with initial_accumulator_value = 1e-8 moves a parameter by ~lr * sign(g) at its first touch, so wherever a gradient
    element is within rounding of 0 the step is decided by the last bit: two free-running trajectories (fp32 device, fp64
    oracle) are 1e-4 apart in the loss after 3 steps and 1.4 % after 5 (measured) - as any two fp32 implementations would
    be.  What CAN be pinned, and is: at EVERY one of the 30 states the device visits, one oracle step from that same state
    (parameters and accumulators read back) agrees with the device step in loss, post-update parameters and accumulators
    within the single-step bounds of step_check(); and the free-running oracle descends like the device does."""
    from cffm_amd import synth
    cfg = CFFMConfig(M=5382, F=10, K=32, D=32, activation='selu', lr=0.05, lamda_att=1.0)
    Xs, ys = synth.batches(cfg.M, cfg.F, 256, 30, seed=11)
    p32 = init_params(cfg, seed=2021, dtype=np.float32)
    eng = engine_for(cfg, p32)
    free = to64(p32)
    free_acc = orc.init_accumulators(free)
    dev_losses, free_losses = [], []
    host_only = {k: v for k, v in p32.items()}
    for i in range(30):
        state = dict(host_only)
        state.update(eng.export_params())
        accs = eng.export_accumulators() if i else None
        step_check(cfg, eng, state, accs, Xs[i], ys[i], 'trajectory')        # runs eng.train_step on batch i
        dev_losses.append(float(eng.loss_buf.cpu()[0]))
        free_losses.append(float(orc.train_step(free, free_acc, Xs[i], ys[i].astype(np.float64), cfg)[0]))
    print('device loss   : ' + ' '.join('%.3f' % v for v in dev_losses))
    print('free fp64 loss: ' + ' '.join('%.3f' % v for v in free_losses))
    assert dev_losses[-1] < 0.2 * dev_losses[0] and abs(np.mean(dev_losses[-5:]) - np.mean(free_losses[-5:])) < 0.25 * np.mean(free_losses[-5:])


def test_second_step_and_reproducibility():
    """Two consecutive steps stay on the oracle's trajectory, and two engines fed the same data end
    bit-identical (fixed reduction orders, no float atomics across workgroups)."""
    cfg, p32, X, y = make_case('bookx-relu', trained_like=False)
    rng = np.random.default_rng(11)
    X2 = rng.integers(0, cfg.M, size=X.shape).astype(np.int32)
    p64 = to64(p32)
    acc = orc.init_accumulators(p64)
    y64 = y.astype(np.float64)
    slack = {}
    for Xs in (X, X2):                     # the gradient tolerance propagated through each of the two updates
        g = oracle_dense_grads(p64, Xs, y, cfg)
        for k, v in adagrad_step_slack(g, acc, cfg.lr, rel=2e-5).items():
            slack[k] = slack.get(k, 0.0) + v
        orc.train_step(p64, acc, Xs, y64, cfg)
    outs = []
    for _ in range(2):
        eng = engine_for(cfg, p32)
        eng.train_step(torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda())
        eng.train_step(torch.from_numpy(X2).cuda(), torch.from_numpy(y).cuda())
        torch.cuda.synchronize()
        outs.append(eng.export_params())
    for k, v in outs[0].items():
        np.testing.assert_array_equal(v, outs[1][k], err_msg=k)
        close(v, p64[k].reshape(v.shape), 'param ' + k, tol=2e-5, extra=None if k not in slack else slack[k].reshape(v.shape))


@pytest.mark.parametrize('name', ['f33-d32-relu', 'f16-d32-b130', 'f20-d64-elu', 'frappe-b1024-dups', 'frappe-selu',
                                  'f16-k32-d32-b70-selu', 'f32-d64-relu'])
def test_two_runs_bit_identical(name):
    """Every kernel family leaves the same bits on two runs of the same two steps: the direct layer-0 kernels (F >= 33,
    whose LDS float atomics only ever target the issuing wave's PRIVATE accumulator plane, so their order is the
    program order of one wave), the tiled layer-0 kernels (F = 16 / 20), the fused small-shape launches (frappe) and
    the sorted sparse update under heavy id duplication."""
    cfg, p32, X, y = make_case(name)
    X2 = np.ascontiguousarray(X[::-1])
    outs = []
    for _ in range(2):
        eng = engine_for(cfg, p32)
        l1 = eng.train_step(torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda())
        l2 = eng.train_step(torch.from_numpy(X2).cuda(), torch.from_numpy(y).cuda())
        torch.cuda.synchronize()
        outs.append((eng.export_params(), eng.export_accumulators(), float(l1), float(l2)))
    assert outs[0][2:] == outs[1][2:]
    for which in (0, 1):
        for k, v in outs[0][which].items():
            np.testing.assert_array_equal(v, outs[1][which][k], err_msg=k)


@pytest.mark.parametrize('loss', ['mse', 'mae', 'log_loss', 'hybrid'])
def test_other_losses(loss):
    CASES['tmp-' + loss] = dict(M=80, F=4, K=8, D=8, act='elu', B=12, loss=loss)
    cfg, p32, X, y = make_case('tmp-' + loss, trained_like=(loss != 'hybrid'))
    if loss == 'log_loss':
        y = (y > 0).astype(np.float32)
    if loss == 'hybrid':               # log_loss on the raw out (CFFM.py:511-513) is finite only for 0 < out < 1
        p32['bias'] = np.float32(0.5)
    eng = engine_for(cfg, p32)
    p64 = to64(p32)
    out_ref, c = orc.forward(p64, X, cfg)
    if loss == 'hybrid':
        assert (out_ref > 0.05).all() and (out_ref < 0.95).all()
    L, dout = orc.loss_and_grad(out_ref, y.astype(np.float64), cfg, p64)
    B = X.shape[0]
    yt = torch.from_numpy(y).cuda()
    eng.forward(torch.from_numpy(X).cuda(), yt)
    eng.backward(yt, B)
    torch.cuda.synchronize()
    close(eng.ws_tensor(B, 'scalars', (16,)).cpu().numpy()[1:2], [L], 'loss')
    close(eng.ws_tensor(B, 'dout', (B,)).cpu().numpy(), dout, 'dout')


def test_gather_edge_cases():
    cfg, p32, X, y = make_case('mltag-elu')
    eng = engine_for(cfg, p32)
    # batch of one, first and last row of the table
    ids = torch.tensor([[0, cfg.M - 1, 5]], dtype=torch.int32).cuda()
    Ei, Eo, fb = eng.gather(ids)
    np.testing.assert_array_equal(Ei.cpu().numpy()[0], p32['inner_embeddings'][[0, cfg.M - 1, 5]])
    np.testing.assert_array_equal(Eo.cpu().numpy()[0], p32['outer_embeddings'][[0, cfg.M - 1, 5]])
    np.testing.assert_array_equal(fb.cpu().numpy()[0], p32['feature_bias'][[0, cfg.M - 1, 5], 0])
    assert eng.predict(torch.zeros((0, 3), dtype=torch.int32).cuda()).shape == (0,)
    # batch of 1 through the whole forward (the reference's tf.squeeze breaks at B == 1, quirk Q10)
    out_ref, _ = orc.forward(to64(p32), X[:1], cfg)
    close(eng.predict(torch.from_numpy(X[:1]).cuda()).cpu().numpy(), out_ref, 'predict B=1')


# ---- BASELINE.json configs at their full sizes (size-independent properties) -----------------------------
FULL = {
    'ml-tag (cfg1 shape)': dict(M=90445, F=3, K=32, D=32, B=1024, act='elu'),
    'book-crossing (cfg3)': dict(M=226336, F=6, K=32, D=32, B=512, act='relu'),
    'synthetic 1M, 32 fields, dim 64, batch 8192 (cfg4)': dict(M=1000000, F=32, K=64, D=64, B=8192, act='relu'),
}


@pytest.mark.parametrize('name', list(FULL))
def test_full_size_configs(name):
    """At full size the oracle cannot materialise the batch (66.6 GB outer map at cfg4), so use what the domain
    offers: examples are independent in the forward pass (any few rows must match the oracle run on just those
    rows), a train step must leave every table row outside the batch bit-identical and its accumulator at
    1e-8, touched rows must all move, and the loss must be finite."""
    from cffm_amd import synth
    c = FULL[name]
    cfg = CFFMConfig(M=c['M'], F=c['F'], K=c['K'], D=c['D'], activation=c['act'])
    p32 = init_params(cfg, seed=5)
    eng = engine_for(cfg, p32)
    X, y = synth.batches(cfg.M, cfg.F, c['B'], 1, seed=9)
    ids, yt = torch.from_numpy(X[0]).cuda(), torch.from_numpy(y[0]).cuda()
    out = eng.predict(ids).cpu().numpy()
    rows = [0, 1, c['B'] // 2, c['B'] - 1]
    ref, _ = orc.forward(to64(p32), X[0][rows], cfg)
    close(out[rows], ref, 'predict rows of ' + name)
    assert np.isfinite(out).all()
    before_inner = eng.inner.clone()
    loss = float(eng.train_step(ids, yt).cpu()[0])
    torch.cuda.synchronize()
    assert np.isfinite(loss) and loss > 0
    L0 = np.sqrt(np.mean((y[0].astype(np.float64) - out) ** 2) + 1e-10)
    assert abs(loss - L0) <= 1e-4 * max(1.0, L0)                 # the loss is that of the pre-update forward
    touched = torch.zeros(cfg.M, dtype=torch.bool, device='cuda')
    touched[ids.reshape(-1).long()] = True
    assert torch.equal(eng.inner[~touched], before_inner[~touched])
    assert bool((eng.inner_acc[~touched] == 1e-8).all()) and bool((eng.inner_acc[touched] > 1e-8).any(dim=1).all())
    assert bool((eng.outer_acc[~touched] == 1e-8).all()) and bool((eng.fbias_acc[~touched] == 1e-8).all())
    assert np.isfinite(eng.predict(ids[:64]).cpu().numpy()).all()


@pytest.mark.parametrize('route,case', [('runs', 'bookx-relu'), ('rows', 'bookx-relu'), ('dense', 'bookx-relu'),
                                        ('rows', 'no-inner'), ('runs', 'no-outer'), ('rows', 'fm-only-nolinatt'),
                                        ('runs', 'frappe-selu'), ('dense', 'frappe-selu')])      # 2 x 128 examples: conv01_bwd_kernel
def test_data_parallel_halves_on_one_gpu(route, case):
    """The N > 1 compute path without a second GPU: two 'ranks' (two engines holding the same replica) run the local half
    of the step on the two halves of a batch, the test plays the role of the two collectives (sum of the flat gradient
    buffers, concatenation of what would be all-gathered), and cffm_dp_apply on each replica must reproduce one oracle
    step on the whole batch - including duplicates of an id that sit on different ranks.
    route 'runs': cffm_dp_local blocks (rows + each rank's sorted key run, merged by rank in cffm_dp_apply);
    route 'rows': cffm_forward + cffm_backward_unscaled rows in any order (sorted inside cffm_dp_apply);
    route 'dense': cffm_dp_local_dense buffers (dense gradients + dense image of the table gradients), ONE sum, then
    cffm_dp_apply_dense."""
    cfg, p32, X, y = make_case(case)
    if X.shape[0] % 2:
        X, y = X[:-1], y[:-1]
    B = X.shape[0]
    h = B // 2
    engines = [engine_for(cfg, p32), engine_for(cfg, p32)]
    ids = [torch.from_numpy(X[:h]).cuda(), torch.from_numpy(X[h:]).cuda()]
    ys = [torch.from_numpy(y[:h]).cuda(), torch.from_numpy(y[h:]).cuda()]
    p64 = to64(p32)
    grads_ref = oracle_dense_grads(p64, X, y, cfg)
    acc = orc.init_accumulators(p64)
    L, _ = orc.train_step(p64, acc, X, y.astype(np.float64), cfg)
    outs = []
    if route == 'dense':
        flats = [e.dp_local_dense(i, t, h, B).clone() for e, i, t in zip(engines, ids, ys)]
        torch.cuda.synchronize()
        fsum = flats[0] + flats[1]                   # the all-reduce
        for e in engines:
            loss = e.dp_apply_dense(fsum.clone(), B)
            torch.cuda.synchronize()
            close(loss.cpu().numpy(), [L], 'loss')
            outs.append(e.export_params())
    grads, rows = [], []
    for e, i, t in zip(engines, ids, ys):
        if route == 'dense':
            break
        if route == 'runs':
            g, r = e.dp_local(i, t, h, B)
        else:
            e.forward(i, t)
            g, r = e.backward_unscaled(i, t, h, B)
        grads.append(g.clone()); rows.append(r.clone())
    torch.cuda.synchronize()
    for e in engines:
        if route == 'dense':
            break
        gsum = grads[0] + grads[1]                       # all-reduce
        rall = torch.cat(rows, dim=0).contiguous()       # all-gather
        loss = e.dp_apply(gsum.clone(), rall, B, 2 if route == 'runs' else 0)
        torch.cuda.synchronize()
        close(loss.cpu().numpy(), [L], 'loss')
        outs.append(e.export_params())
    for k, v in outs[0].items():
        np.testing.assert_array_equal(v, outs[1][k], err_msg=k)          # replicas stay bit-identical (same summed operands)
        extra = None
        if k in grads_ref:
            gk = grads_ref[k].reshape(v.shape)
            dg = 2e-5 * max(np.abs(gk).max(), 1e-30)
            u = lambda t: cfg.lr * t / np.sqrt(1e-8 + t * t)
            extra = np.maximum(np.abs(u(gk + dg) - u(gk)), np.abs(u(gk - dg) - u(gk)))
        close(v, p64[k].reshape(v.shape), 'param ' + k, tol=2e-5, extra=extra)
    touched = np.zeros(cfg.M, dtype=bool)
    touched[X.reshape(-1)] = True
    for k in ('inner_embeddings', 'outer_embeddings', 'feature_bias'):
        np.testing.assert_array_equal(outs[0][k][~touched], p32[k][~touched])


def test_merge_of_sorted_runs_equals_a_global_sort():
    """Eight ranks' worth of runs (the N = 8 case the driver measures): the merged key order cffm_dp_apply builds from the
    per-rank sorted runs is exactly the sorted order of all (id, global slot) keys - checked through its effect: the
    table update equals the one computed from the same rows handed over unsorted (n_runs = 0, radix sort inside)."""
    cfg, p32, X, y = make_case('frappe-selu')
    R, B = 8, 64
    rng = np.random.default_rng(3)
    engines = [engine_for(cfg, p32), engine_for(cfg, p32)]
    W = 1 + cfg.K + cfg.D + 1
    blocks, rows_only = [], []
    e0 = engines[0]
    for r in range(R):
        Xi = rng.integers(0, 40, size=(B, cfg.F)).astype(np.int32)       # few distinct ids: long segments across ranks
        yi = rng.choice([-1.0, 1.0], size=(B,)).astype(np.float32)
        g, blk = e0.dp_local(torch.from_numpy(Xi).cuda(), torch.from_numpy(yi).cuda(), B, R * B)
        blocks.append(blk.clone())
        rows_only.append(blk[:B * cfg.F * W].reshape(B * cfg.F, W).clone())
    grad = g.clone()
    e1 = engines[1]
    e0.load_params(p32); e1.load_params(p32)
    e0.dp_apply(grad.clone(), torch.cat(blocks).contiguous(), R * B, R)
    e1.dp_apply(grad.clone(), torch.cat(rows_only, dim=0).contiguous(), R * B, 0)
    torch.cuda.synchronize()
    a, b = e0.export_params(), e1.export_params()
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert not np.array_equal(a['outer_embeddings'], p32['outer_embeddings'])


@pytest.mark.parametrize('case', ['bookx-relu', 'f10-d32-b100-elu'])     # the second: a conv01_bwd_kernel shape on the stage-by-stage path
def test_regularised_square_loss_step(case):
    """--lamda > 0 (CFFM.py:489-491): l2_loss data term, dense table gradients scatter(row grads) + lamda * w with
    the outer table scaled by lamda_att (Q13), dense Adagrad over both tables, sparse feature_bias."""
    cfg, p32, X, y = make_case(case)
    cfg.lamda_bilinear = 0.02
    eng = engine_for(cfg, p32)
    p64 = to64(p32)
    B = X.shape[0]
    eng.forward(torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda())
    torch.cuda.synchronize()
    hook = lambda cache: adopt_device_kinks(cfg, eng, B, cache)
    out, c = orc.forward(p64, X, cfg)
    hook(c)
    _, dout = orc.loss_and_grad(out, y.astype(np.float64), cfg, p64)
    g = orc.backward(p64, c, dout, cfg)
    ids = X.reshape(-1)
    dense = {}
    for name, key, scale in (('inner_embeddings', 'd_inner_rows', cfg.lamda_bilinear), ('outer_embeddings', 'd_outer_rows', cfg.lamda_att)):
        t = scale * p64[name]
        np.add.at(t, ids, g[key].reshape(ids.shape[0], -1))
        dense[name] = t
    acc = orc.init_accumulators(p64)
    L, _ = orc.train_step(p64, acc, X, y.astype(np.float64), cfg, cache_hook=hook)
    loss = eng.train_step(torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda())
    torch.cuda.synchronize()
    data_term = 0.5 * np.sum((y.astype(np.float64) - out) ** 2)
    close(loss.cpu().numpy(), [data_term], 'loss (data term)')
    got, gacc = eng.export_params(), eng.export_accumulators()
    for k in ('inner_embeddings', 'outer_embeddings'):
        gk = dense[k]
        dg = 1e-5 * np.abs(gk).max()
        u = lambda t: cfg.lr * t / np.sqrt(1e-8 + t * t)
        extra = np.maximum(np.abs(u(gk + dg) - u(gk)), np.abs(u(gk - dg) - u(gk)))
        close(got[k], p64[k], 'param ' + k, tol=2e-5, extra=extra)
        close(gacc[k], acc[k], 'acc ' + k, tol=2e-5, extra=2 * np.abs(gk) * dg + dg * dg)
        assert (gacc[k] > 1e-8).mean() > 0.99           # every row was updated
    touched = np.zeros(cfg.M, dtype=bool)
    touched[ids] = True
    np.testing.assert_array_equal(got['feature_bias'][~touched], p32['feature_bias'][~touched])


@pytest.mark.parametrize('opt', ['GradientDescentOptimizer', 'MomentumOptimizer', 'AdamOptimizer'])
@pytest.mark.parametrize('name', ['bookx-relu', 'frappe-selu', 'no-inner', 'no-outer'])
def test_other_optimizers(name, opt):
    """cffm_train_step_opt: the other create_optimizer branches (CFFM.py:519-529) against the oracle.  Step 1 is held
    to the gradient tolerance propagated through the update rule (linear for SGD/Momentum; Adam's first step is
    lr*g/(|g| + 1e-8/sqrt(1-b2)), a smoothed sign); step 2 (other ids) checks the slot state carried over: Momentum
    leaves unlooked-up rows alone, Adam moves every row that has a non-zero first moment."""
    cfg, p32, X, y = make_case(name)
    cfg.optimizer, cfg.lr = opt, (0.01 if opt == 'AdamOptimizer' else 1e-4)
    eng = engine_for(cfg, p32)
    p64 = to64(p32)
    B = X.shape[0]
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    eng.forward(Xd, yd)
    torch.cuda.synchronize()
    hook = (lambda cache: adopt_device_kinks(cfg, eng, B, cache)) if cfg.outer_conv else None
    grads = oracle_dense_grads(p64, X, y, cfg, hook)
    st = orc.init_opt_state(p64, opt)
    L, _ = orc.train_step_opt(p64, st, X, y.astype(np.float64), cfg, cache_hook=hook)
    loss = eng.train_step(Xd, yd)
    torch.cuda.synchronize()
    close(loss.cpu().numpy(), [L], 'loss')
    got = eng.export_params()
    e1 = 1e-8 / np.sqrt(0.001)
    for k, v in got.items():
        extra = None
        if k in grads:
            gk = grads[k].reshape(v.shape)
            dg = 1e-5 * max(np.abs(gk).max(), 1e-30)
            if opt == 'AdamOptimizer':
                u = lambda t: cfg.lr * t / (np.abs(t) + e1)
                # + 1e-5 of the step itself: w0 - lr*sign-like step can cancel to ~1e-8 (conv biases start at 0.01 = lr)
                extra = np.maximum(np.abs(u(gk + dg) - u(gk)), np.abs(u(gk - dg) - u(gk))) + 1e-5 * cfg.lr
            else:
                extra = cfg.lr * dg * np.ones_like(gk)
        close(v, p64[k].reshape(v.shape), 'param ' + k, tol=2e-5, extra=extra)
    slots = eng.export_accumulators()
    ref_slot = st['m'] if opt == 'AdamOptimizer' else st.get('acc')
    live = [k for k in ('inner_embeddings', 'dense_kernel') if cfg.inner_conv] + \
           [k for k in ('outer_embeddings', 'dense_1_kernel', 'outer_layer_conv_weight_0') if cfg.outer_conv] + ['feature_bias']
    if ref_slot is not None:
        for k in live:
            close(slots[k], ref_slot[k].reshape(slots[k].shape), 'slot ' + k, tol=2e-5)
    if opt == 'AdamOptimizer':
        v2 = eng.export_second_moments()
        for k in live:
            close(v2[k], st['v'][k].reshape(v2[k].shape), 'v ' + k, tol=4e-5)
    # a disabled branch has no variables in the reference graph: its table and slots are exactly as they were
    for flag, k in ((cfg.inner_conv, 'inner_embeddings'), (cfg.outer_conv, 'outer_embeddings')):
        if not flag:
            np.testing.assert_array_equal(got[k], p32[k])
            assert np.all(slots[k] == 0)

    # ---- second step on other ids
    rng = np.random.default_rng(21)
    X2 = rng.integers(0, cfg.M // 2, size=X.shape).astype(np.int32)
    before = got
    p1 = {k: np.array(v) for k, v in p64.items()}
    orc.train_step_opt(p64, st, X2, y.astype(np.float64), cfg)
    eng.train_step(torch.from_numpy(X2).cuda(), yd)
    torch.cuda.synchronize()
    got = eng.export_params()
    only1 = np.setdiff1d(X.reshape(-1), X2.reshape(-1))
    never = np.setdiff1d(np.arange(cfg.M), np.concatenate([X.reshape(-1), X2.reshape(-1)]))
    assert only1.size and never.size
    for k in ('inner_embeddings', 'outer_embeddings', 'feature_bias'):
        np.testing.assert_array_equal(got[k][never], p32[k][never])
        moved = got[k][only1] != before[k][only1]
        if (k == 'inner_embeddings' and not cfg.inner_conv) or (k == 'outer_embeddings' and not cfg.outer_conv):
            assert not (got[k] != p32[k]).any(), k
        elif opt == 'AdamOptimizer':
            assert moved.mean() > 0.9, k                      # m != 0 keeps pushing the row
        else:
            assert not moved.any(), k
    # the step-2 MOVE (slot state carried over from step 1) against the oracle's move
    for k, v in got.items():
        d_dev = v.astype(np.float64) - before[k].astype(np.float64)
        d_ref = p64[k].reshape(v.shape) - p1[k].reshape(v.shape)
        err = np.abs(d_dev - d_ref)
        ok = err <= 1e-3 * max(np.abs(d_ref).max(), 1e-30) + 4e-7 * np.abs(v)
        # Adam's smoothed sign flips on gradients within rounding of 0: allow a vanishing fraction of such elements
        assert ok.mean() > (0.999 if opt == 'AdamOptimizer' else 0.99999), (k, float(ok.mean()), float(err.max()))


@pytest.mark.parametrize('opt', ['GradientDescentOptimizer', 'MomentumOptimizer', 'AdamOptimizer'])
def test_regularised_square_loss_with_other_optimizers(opt):
    """--lamda > 0 with the other create_optimizer branches (CFFM.py:489-491 + :519-529): the two embedding tables get a
    dense gradient scatter(row grads) + lamda * w (outer table: lamda_att, Q13) under every optimizer; feature_bias
    keeps its sparse gradient.  Two steps against the oracle."""
    cfg, p32, X, y = make_case('bookx-relu')
    cfg.lamda_bilinear = 0.02
    cfg.optimizer, cfg.lr = opt, (0.01 if opt == 'AdamOptimizer' else 1e-4)
    eng = engine_for(cfg, p32)
    p64 = to64(p32)
    B = X.shape[0]
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    eng.forward(Xd, yd)
    torch.cuda.synchronize()
    hook = lambda cache: adopt_device_kinks(cfg, eng, B, cache)
    grads = oracle_dense_grads(p64, X, y, cfg, hook)
    for k, scale in (('inner_embeddings', cfg.lamda_bilinear), ('outer_embeddings', cfg.lamda_att)):
        grads[k] = grads[k] + scale * p64[k]
    st = orc.init_opt_state(p64, opt)
    orc.train_step_opt(p64, st, X, y.astype(np.float64), cfg, cache_hook=hook)
    eng.train_step(Xd, yd)
    torch.cuda.synchronize()
    got = eng.export_params()
    e1 = 1e-8 / np.sqrt(0.001)
    for k, v in got.items():
        extra = None
        if k in grads:
            gk = grads[k].reshape(v.shape)
            dg = 1e-5 * (np.abs(gk) + max(float(np.sqrt(np.mean(gk * gk))), 1e-30))
            if opt == 'AdamOptimizer':
                u = lambda t: cfg.lr * t / (np.abs(t) + e1)
                extra = np.maximum(np.abs(u(gk + dg) - u(gk)), np.abs(u(gk - dg) - u(gk))) + 1e-5 * cfg.lr
            else:
                extra = cfg.lr * dg
        close(v, p64[k].reshape(v.shape), 'L2 %s param %s' % (opt, k), tol=2e-5, extra=extra)
    # every row of both tables moved (dense gradient), rows of feature_bias nobody looked up did not (SGD / Momentum)
    assert (got['inner_embeddings'] != p32['inner_embeddings']).mean() > 0.99
    assert (got['outer_embeddings'] != p32['outer_embeddings']).mean() > 0.99
    touched = np.zeros(cfg.M, dtype=bool)
    touched[X.reshape(-1)] = True
    if opt != 'AdamOptimizer':
        np.testing.assert_array_equal(got['feature_bias'][~touched], p32['feature_bias'][~touched])
    # second step: the slot state (momentum / Adam moments over ALL rows) carried over
    before = got
    p1 = {k: np.array(v) for k, v in p64.items()}
    rng = np.random.default_rng(5)
    X2 = rng.integers(0, cfg.M, size=X.shape).astype(np.int32)
    orc.train_step_opt(p64, st, X2, y.astype(np.float64), cfg)
    eng.train_step(torch.from_numpy(X2).cuda(), yd)
    torch.cuda.synchronize()
    got = eng.export_params()
    for k, v in got.items():
        d_dev = v.astype(np.float64) - before[k].astype(np.float64)
        d_ref = p64[k].reshape(v.shape) - p1[k].reshape(v.shape)
        ok = np.abs(d_dev - d_ref) <= 1e-3 * max(np.abs(d_ref).max(), 1e-30) + 4e-7 * np.abs(v)
        assert ok.mean() > (0.999 if opt == 'AdamOptimizer' else 0.99999), (k, float(ok.mean()))


def test_row_sharded_halves_on_one_gpu():
    """cfg5's mode (row-sharded tables) without a second GPU: two engines each own the rows r % 2 == rank of the three
    tables; the test plays the all-to-alls of cffm_amd.dist.ShardedStep (ids to the owner, rows back, packed row
    gradients to the owner) and the all-reduce.  Union of the shards + replicated parameters == one oracle step on
    the whole batch with whole tables."""
    import copy
    from cffm_amd.dist import local_rows_count, shard_params
    from cffm_amd.engine import HipEngine
    cfg, p32, X, y = make_case('bookx-relu')
    B, F, G = X.shape[0], cfg.F, 2
    h = B // 2
    K, D = cfg.K, cfg.D
    engines = []
    for r in range(G):
        lc = copy.copy(cfg)
        lc.M = local_rows_count(cfg.M, r, G)
        engines.append(HipEngine(lc, params=shard_params(p32, r, G)))
    halves = [X[:h], X[h:]]
    ys = [torch.from_numpy(y[:h]).cuda(), torch.from_numpy(y[h:]).cuda()]
    grads, packed = [], []
    for r in range(G):
        flat = torch.from_numpy(halves[r].reshape(-1)).cuda().long()
        staged = torch.empty((flat.numel(), K + D + 4), device='cuda')       # packed records: (inner | outer | bias, 0, 0, 0)
        for o in range(G):                               # "all-to-all": ask owner o for its rows
            m = (flat % G) == o
            staged[m] = engines[o].gather_packed((flat[m] // G).to(torch.int32))
        e = engines[r]
        e.forward_rows(staged[:, :K].contiguous(), staged[:, K:K + D].contiguous(), staged[:, K + D].contiguous(), ys[r], h)
        g, rows = e.backward_unscaled((flat // G).to(torch.int32).reshape(h, F), ys[r], h, B)
        grads.append(g.clone()); packed.append((flat % G, rows.clone()))
    torch.cuda.synchronize()
    gsum = grads[0] + grads[1]
    p64 = to64(p32)
    grads_ref = oracle_dense_grads(p64, X, y, cfg)
    acc = orc.init_accumulators(p64)
    L, _ = orc.train_step(p64, acc, X, y.astype(np.float64), cfg)
    for o in range(G):
        recv = torch.cat([rows[own == o] for own, rows in packed], dim=0).contiguous()
        loss = engines[o].dp_apply(gsum.clone(), recv, B)
        torch.cuda.synchronize()
        close(loss.cpu().numpy(), [L], 'loss')
        got = engines[o].export_params()
        for k, v in got.items():
            ref = p64[k][o::G] if k in ('inner_embeddings', 'outer_embeddings', 'feature_bias') else p64[k]
            extra = None
            if k in grads_ref:
                gk = grads_ref[k][o::G] if ref is not p64[k] else grads_ref[k]
                gk = gk.reshape(v.shape)
                dg = 2e-5 * max(np.abs(grads_ref[k]).max(), 1e-30)
                u = lambda t: cfg.lr * t / np.sqrt(1e-8 + t * t)
                extra = np.maximum(np.abs(u(gk + dg) - u(gk)), np.abs(u(gk - dg) - u(gk)))
            close(v, ref.reshape(v.shape), 'rank %d param %s' % (o, k), tol=2e-5, extra=extra)


def test_empty_batch_is_a_no_op():
    cfg, p32, X, y = make_case('tiny-relu')
    eng = engine_for(cfg, p32)
    before = eng.export_params()
    ids = torch.zeros((0, cfg.F), dtype=torch.int32, device='cuda')
    assert eng.predict(ids).shape == (0,)
    eng.train_step(ids, torch.zeros(0, device='cuda'))
    torch.cuda.synchronize()
    after = eng.export_params()
    for k, v in before.items():
        np.testing.assert_array_equal(v, after[k])


@pytest.mark.parametrize('B', [64, 1024])
def test_out_of_range_ids_do_not_split_a_segment(B):
    """A bad id (outside [0, M)) is read as a clamped row by the gather and skipped by the update.  Its LOW bits may equal
    a valid id: the radix sorts of the sparse update look at ceil(log2(M + 1)) id bits only, so it must not land inside
    that id's run (two wavefronts would then update one row).  Property: the step with bad ids equals, bit for bit and on
    every row but the clamp row M - 1, the step where those slots carry M - 1 instead.  B = 64: single-launch forward with
    keys placed by counting rank; B = 1024: 10,240 lookups, rocPRIM radix sort."""
    cfg, p32, _, _ = make_case('frappe-selu')
    rng = np.random.default_rng(2)
    X = rng.integers(0, cfg.M - 1, size=(B, cfg.F)).astype(np.int32)
    y = rng.choice([-1.0, 1.0], size=(B,)).astype(np.float32)
    X[:, 0] = 37                                       # a long segment of a valid id
    bad_lo = (1 << int(np.ceil(np.log2(cfg.M + 1)))) + 37          # low bits == 37
    Xa, Xb = X.copy(), X.copy()
    slots = [(3, 1), (B // 2, 4), (B - 1, 9), (5, 2)]
    for (b, f), v in zip(slots, (bad_lo, -7, bad_lo + (1 << 20), cfg.M)):
        Xa[b, f] = v
        Xb[b, f] = cfg.M - 1 if v >= 0 else 0          # what the gather reads for it
    ea, eb = engine_for(cfg, p32), engine_for(cfg, p32)
    yt = torch.from_numpy(y).cuda()
    la = ea.train_step(torch.from_numpy(Xa).cuda(), yt).clone()
    lb = eb.train_step(torch.from_numpy(Xb).cuda(), yt).clone()
    torch.cuda.synchronize()
    assert torch.equal(la, lb)
    keep = np.ones(cfg.M, dtype=bool)
    keep[[0, cfg.M - 1]] = False                       # the two clamp rows receive the bad slots' gradients in run B only
    pa, pb = ea.export_params(), eb.export_params()
    aa, ab = ea.export_accumulators(), eb.export_accumulators()
    for k in pa:
        if k in ('inner_embeddings', 'outer_embeddings', 'feature_bias'):
            np.testing.assert_array_equal(pa[k][keep], pb[k][keep], err_msg=k)
            np.testing.assert_array_equal(aa[k][keep], ab[k][keep], err_msg='acc ' + k)
        else:
            np.testing.assert_array_equal(pa[k], pb[k], err_msg=k)
