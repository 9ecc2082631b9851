"""Oracle self-checks (CPU).  The oracle is 'parity unpinned' against TF-1.14; what CAN be
pinned is pinned here: analytic known answers from the reference source, and agreement with an
independent torch-autograd reading of the same graph (tests/twin_torch.py)."""
import math

import numpy as np
import pytest
import torch

from cffm_amd.spec import CFFMConfig, init_params, logged_param_count, param_shapes
from oracle import cffm_oracle as orc
from tests import twin_torch as twin

CASES = [
    dict(F=3, K=8, D=8, act='relu', B=5),
    dict(F=4, K=8, D=8, act='selu', B=3),
    dict(F=4, K=32, D=16, act='elu', B=4),
    dict(F=5, K=16, D=32, act='gelu', B=2),
    dict(F=6, K=8, D=4, act='prelu', B=7),
    dict(F=3, K=8, D=8, act='relu', B=1),
]


def _setup(case, seed=0, linear_att=1, loss_type='square_loss', emb_scale=1.0):
    cfg = CFFMConfig(M=40, F=case['F'], K=case['K'], D=case['D'], activation=case['act'],
                     lamda_att=1.7, beta_outer=1.0, linear_att=linear_att, loss_type=loss_type)
    p = init_params(cfg, seed=seed, dtype=np.float64)
    rng = np.random.default_rng(seed + 1)
    # make every branch non-trivial: feature_bias starts at exactly 0 in the reference
    p['feature_bias'] = rng.standard_normal(p['feature_bias'].shape) * 0.3
    p['outer_embeddings'] = p['outer_embeddings'] * 30.0 * emb_scale
    p['inner_embeddings'] = p['inner_embeddings'] * 5.0 * emb_scale
    X = rng.integers(0, cfg.M, size=(case['B'], cfg.F))
    X[0, 0] = X[-1, 0]                       # duplicate ids inside the batch
    y = rng.choice([-1.0, 1.0], size=(case['B'],))
    return cfg, p, X, y


def _torch_params(p):
    return {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('linear_att', [1, 0])
def test_forward_and_grads_match_torch_twin(case, linear_att):
    cfg, p, X, y = _setup(case, linear_att=linear_att)
    out, cache = orc.forward(p, X, cfg)
    L, dout = orc.loss_and_grad(out, y, cfg, p)
    g = orc.backward(p, cache, dout, cfg)

    tp = _torch_params(p)
    tout = twin.forward(tp, torch.tensor(X), cfg)
    tL = twin.loss(tout, torch.tensor(y), cfg)
    tL.backward()

    np.testing.assert_allclose(out, tout.detach().numpy(), rtol=1e-11, atol=1e-12)
    assert abs(L - tL.item()) < 1e-12
    ids = X.reshape(-1)
    for name, v in tp.items():
        if name in ('inner_embeddings', 'outer_embeddings', 'feature_bias'):
            key = {'inner_embeddings': 'd_inner_rows', 'outer_embeddings': 'd_outer_rows',
                   'feature_bias': 'd_bias_rows'}[name]
            dense = np.zeros(p[name].shape)
            np.add.at(dense, ids, g[key].reshape(ids.shape[0], -1))
            np.testing.assert_allclose(dense, v.grad.numpy(), rtol=1e-9, atol=1e-12, err_msg=name)
        elif name in g:
            np.testing.assert_allclose(np.asarray(g[name]).reshape(p[name].shape), v.grad.numpy(),
                                       rtol=1e-9, atol=1e-12, err_msg=name)
        else:
            # outer_W / outer_b are never used; the last conv layer is dead code (CFFM.py:394-396)
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, name
    dead = 'outer_layer_conv_weight_%d' % (cfg.Lc - 1)
    assert dead not in g and tp[dead].grad is None


@pytest.mark.parametrize('loss_type', ['mse', 'mae', 'log_loss', 'hybrid'])
def test_other_losses_match_twin(loss_type):
    cfg, p, X, y = _setup(CASES[1], loss_type=loss_type)
    if loss_type == 'log_loss':
        y = (y > 0).astype(np.float64)
    out, _ = orc.forward(p, X, cfg)
    if loss_type == 'hybrid':          # finite only while 0 < out < 1 (the reference feeds the raw out to log_loss)
        out = 0.5 + 0.4 * np.tanh(out)
    L, dout = orc.loss_and_grad(out, y, cfg, p)
    to = torch.tensor(out, requires_grad=True)
    tL = twin.loss(to, torch.tensor(y), cfg)
    tL.backward()
    assert abs(L - tL.item()) < 1e-12
    np.testing.assert_allclose(dout, to.grad.numpy(), rtol=1e-10, atol=1e-14)


def test_adagrad_step_matches_twin():
    """TF semantics: duplicates summed first, acc0 = 1e-8, no epsilon; untouched rows unchanged."""
    cfg, p, X, y = _setup(CASES[0])
    p0 = {k: v.copy() for k, v in p.items()}
    acc = orc.init_accumulators(p)
    orc.train_step(p, acc, X, y, cfg)

    tp = _torch_params(p0)
    tL = twin.loss(twin.forward(tp, torch.tensor(X), cfg), torch.tensor(y), cfg)
    tL.backward()
    for name, v in tp.items():
        if v.grad is None:
            np.testing.assert_array_equal(p[name], p0[name])
            np.testing.assert_array_equal(acc[name], np.full_like(p0[name], 1e-8))
            continue
        gr = v.grad.numpy()
        a = 1e-8 + gr * gr
        want = p0[name] - cfg.lr * gr / np.sqrt(a)
        if name in ('inner_embeddings', 'outer_embeddings', 'feature_bias'):
            touched = np.zeros(cfg.M, dtype=bool)
            touched[X.reshape(-1)] = True
            np.testing.assert_allclose(p[name][touched], want[touched], rtol=1e-9, atol=1e-12)
            np.testing.assert_array_equal(p[name][~touched], p0[name][~touched])
            np.testing.assert_array_equal(acc[name][~touched], 1e-8)
            np.testing.assert_allclose(acc[name][touched], a[touched], rtol=1e-9)
        else:
            np.testing.assert_allclose(p[name], want, rtol=1e-9, atol=1e-12, err_msg=name)


def test_sparse_adagrad_sums_duplicates_before_squaring():
    table = np.zeros((4, 2))
    acc = np.full((4, 2), 1e-8)
    ids = np.array([1, 1, 3])
    rows = np.array([[1.0, 2.0], [3.0, -2.0], [0.5, 0.5]])
    orc.adagrad_sparse(table, acc, ids, rows, lr=0.1)
    np.testing.assert_allclose(acc[1], [1e-8 + 16.0, 1e-8 + 0.0])
    np.testing.assert_allclose(table[1], [-0.1 * 4 / math.sqrt(1e-8 + 16.0), 0.0])
    np.testing.assert_allclose(acc[3], 1e-8 + 0.25)
    assert (table[[0, 2]] == 0).all() and (acc[[0, 2]] == 1e-8).all()


# ---- analytic known answers from the reference source -------------------------
def test_logged_param_counts():
    # '#params' formula of calculate_parameters (CFFM.py:543-553) on README.md:35-40 dataset sizes
    assert logged_param_count(CFFMConfig(M=5382, F=10)) == 390718
    assert logged_param_count(CFFMConfig(M=90445, F=3)) == 5879143
    assert logged_param_count(CFFMConfig(M=226336, F=6)) == 14716480
    assert orc.count_logged_params(5382, 10, 32, 32) == 390718
    assert orc.count_logged_params(1000000, 32, 64, 64) == 134908920


def test_shapes():
    cfg = CFFMConfig(M=100, F=10, K=32, D=32)
    s = param_shapes(cfg)
    assert s == orc.param_shapes(100, 10, 32, 32)
    assert s['dense_kernel'] == (45 * 32, 1)         # P*K flatten (CFFM.py:333, 16 -> K/2)
    assert s['dense_1_kernel'] == (62, 32)           # t1 width 2D-2 (CFFM.py:394-396)
    assert cfg.Lc == 5 and cfg.live_layers == 4


def test_init_state_linear_term_is_zero():
    # feature_bias starts at exactly 0 (CFFM.py:276-277) -> lin == dense_3 bias == 0
    cfg = CFFMConfig(M=50, F=4, K=8, D=8)
    p = init_params(cfg, dtype=np.float64)
    X = np.random.default_rng(0).integers(0, 50, size=(6, 4))
    _, cache = orc.forward(p, X, cfg)
    assert (cache['lin'] == 0).all()


def test_eva_termination_truth_table():
    assert not orc.eva_termination([1, 2, 3, 4, 5])              # len must exceed 5
    assert orc.eva_termination([9, 1, 2, 3, 4, 5])
    assert not orc.eva_termination([9, 1, 2, 3, 3, 5])           # strict
    assert not orc.eva_termination([1, 2, 3, 4, 5, 4])


def test_clipped_metrics_match_sklearn():
    from sklearn.metrics import mean_squared_error, r2_score
    rng = np.random.default_rng(3)
    yt = rng.choice([-1.0, 1.0], size=200)
    yp = rng.standard_normal(200) * 2
    rmse, r2 = orc.clipped_rmse_r2(yp, yt)
    b = np.minimum(np.maximum(yp, -1), 1)
    assert abs(rmse - math.sqrt(mean_squared_error(yt, b))) < 1e-14
    assert abs(r2 - r2_score(yt, b)) < 1e-12


def test_conv0_rank1_factorisation():
    """The identity the fused HIP conv0 relies on: with rank-1 input channels
    Y[y,x,q] = sum_{dh,dw} sum_i u_i[y] * (sum_{j>i} W[dh,dw,(i,j),q] * v_j[x])."""
    rng = np.random.default_rng(5)
    F, D = 5, 8
    P = F * (F - 1) // 2
    E = rng.standard_normal((F, D))
    W = rng.standard_normal((2, 2, P, P))
    ii, jj = orc.pair_index(F)
    A = (E[ii, :, None] * E[jj, None, :]).transpose(1, 2, 0)[None]
    direct = (orc._im2col_2x2(A) @ W.reshape(4 * P, P))[0]
    fact = np.zeros_like(direct)
    for dh in range(2):
        for dw in range(2):
            for pidx, (i, j) in enumerate(zip(ii, jj)):
                fact += (E[i, dh::2][:, None, None] * E[j, dw::2][None, :, None]) * W[dh, dw, pidx][None, None, :]
    np.testing.assert_allclose(direct, fact, rtol=1e-12, atol=1e-13)


def test_regularised_square_loss_step_matches_twin():
    """lamda > 0 (CFFM.py:489-491): the l2 terms make the table gradients dense, so every row and accumulator of
    the inner/outer tables moves; the outer table is scaled by lamda_att (quirk Q13)."""
    cfg, p, X, y = _setup(CASES[1])
    cfg.lamda_bilinear = 0.03
    p0 = {k: v.copy() for k, v in p.items()}
    acc = orc.init_accumulators(p)
    L, _ = orc.train_step(p, acc, X, y, cfg)
    tp = _torch_params(p0)
    tL = twin.loss(twin.forward(tp, torch.tensor(X), cfg), torch.tensor(y), cfg, tp)
    tL.backward()
    assert abs(L - tL.item()) < 1e-10
    for name in ('inner_embeddings', 'outer_embeddings', 'feature_bias', 'dense_1_kernel', 'outer_layer_conv_weight_0', 'bias_W'):
        gr = tp[name].grad.numpy()
        want = p0[name] - cfg.lr * gr / np.sqrt(1e-8 + gr * gr)
        np.testing.assert_allclose(p[name], want, rtol=1e-9, atol=1e-12, err_msg=name)
    untouched = np.ones(cfg.M, dtype=bool)
    untouched[X.reshape(-1)] = False
    assert (p['inner_embeddings'][untouched] != p0['inner_embeddings'][untouched]).any()      # dense update
    np.testing.assert_array_equal(p['feature_bias'][untouched], p0['feature_bias'][untouched])  # still sparse


@pytest.mark.parametrize('opt', ['GradientDescentOptimizer', 'MomentumOptimizer', 'AdamOptimizer'])
def test_other_optimizers_match_torch(opt):
    """Two steps of the other create_optimizer branches (CFFM.py:519-529) against torch autograd + the TF-1.14 update
    formulas written out here (torch.optim.SGD for SGD / momentum 0.95; TF's Adam puts epsilon OUTSIDE the bias
    correction, so that one is spelled out).  Sparse semantics: Momentum touches only the looked-up rows, Adam moves
    every row."""
    cfg, p, X, y = _setup(CASES[1])
    cfg.optimizer, cfg.lr = opt, 0.01
    p0 = {k: v.copy() for k, v in p.items()}
    st = orc.init_opt_state(p, opt)
    rng = np.random.default_rng(9)
    X2 = rng.integers(0, cfg.M, size=X.shape)
    for Xs in (X, X2):
        orc.train_step_opt(p, st, Xs, y, cfg)

    tp = _torch_params(p0)
    live = [v for k, v in tp.items() if k not in ('outer_W', 'outer_b', 'outer_layer_conv_weight_%d' % (cfg.Lc - 1),
                                                  'outer_layer_conv_bias_%d' % (cfg.Lc - 1))]
    tables = ('inner_embeddings', 'outer_embeddings', 'feature_bias')
    state = {k: [torch.zeros_like(v), torch.zeros_like(v)] for k, v in tp.items()}
    for t, Xs in enumerate((X, X2), start=1):
        for v in tp.values():
            v.grad = None
        twin.loss(twin.forward(tp, torch.tensor(Xs), cfg), torch.tensor(y), cfg).backward()
        touched = torch.zeros(cfg.M, dtype=torch.bool)
        touched[torch.tensor(Xs).reshape(-1)] = True
        with torch.no_grad():
            for k, v in tp.items():
                if v.grad is None:
                    continue
                g = v.grad
                rows = touched.reshape(-1, *([1] * (v.dim() - 1))) if k in tables else torch.ones_like(v, dtype=torch.bool)
                if opt == 'GradientDescentOptimizer':
                    v -= cfg.lr * g
                elif opt == 'MomentumOptimizer':
                    a = torch.where(rows, 0.95 * state[k][0] + g, state[k][0])
                    state[k][0] = a
                    v -= torch.where(rows, cfg.lr * a, torch.zeros_like(a))
                else:
                    m = 0.9 * state[k][0] + 0.1 * g
                    s2 = 0.999 * state[k][1] + 0.001 * g * g
                    state[k] = [m, s2]
                    v -= cfg.lr * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (torch.sqrt(s2) + 1e-8)
    assert len(live) > 10
    for k, v in tp.items():
        np.testing.assert_allclose(p[k], v.detach().numpy(), rtol=1e-9, atol=1e-12, err_msg=k)
    if opt == 'AdamOptimizer':        # rows seen in step 1 but not in step 2 still move in step 2
        only1 = np.setdiff1d(X.reshape(-1), X2.reshape(-1))
        assert only1.size and st['t'] == 2


def test_twin_train_step_equals_oracle_train_step():
    """The threaded torch CPU step bench.py times as cpu_baseline is the same step as the numpy oracle's."""
    cfg, p, X, y = _setup(CASES[2])
    tp = _torch_params(p)
    tacc = {k: torch.full_like(v, 1e-8) for k, v in tp.items()}
    acc = orc.init_accumulators(p)
    for _ in range(2):
        L = orc.train_step(p, acc, X, y, cfg)[0]
        tL = twin.train_step(tp, tacc, torch.tensor(X), torch.tensor(y), cfg)
        assert abs(L - tL) < 1e-10
    for k, v in tp.items():
        np.testing.assert_allclose(v.detach().numpy(), p[k], rtol=1e-8, atol=1e-11, err_msg=k)
