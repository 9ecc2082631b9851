"""GPU parity at the FULL sizes of BASELINE.json configs[3] (32 fields, dim 64, 1 M features, batch 8192) and of one
GPU's share of configs[4] (10 M features, row-sharded tables), where the oracle cannot materialise the batch (the outer
product map alone is 66.6 GB).  What the domain offers instead:

* examples are independent in the forward pass, and every gradient is LINEAR in dL/dout: the per-example quantities of any
  few rows of the batch (out, dC, dt1, dEi, dEo, dfb) must equal the oracle run on just those rows with the batch's loss
  normaliser L;
* a batch built from n distinct examples repeated B/n times (labels drawn per row) has the dense gradients of those n
  examples with dL/dout summed over the copies - so ONE oracle pass over n = 128 rows pins every dense gradient, the
  duplicates-summed table gradients and the whole Adagrad step of the full-size launch (64-bit offsets, all 8192 rows
  walking every gradient slab, the radix-sorted sparse update with 64-fold duplicates).
"""
import os

import numpy as np
import pytest
import torch

from cffm_amd import synth
from cffm_amd.spec import CFFMConfig, init_params
from oracle import cffm_oracle as orc
from tests.test_gpu_parity import close, pad_channels, to64

pytestmark = pytest.mark.gpu


def _adopt_kinks_rows(cfg, eng, B, cache, dev_rows):
    """adopt_device_kinks of test_gpu_parity for an oracle cache that holds only the rows dev_rows of the device batch."""
    n = 0
    idx = torch.as_tensor(dev_rows, device='cuda')
    for l in range(cfg.live_layers):
        z = cache['zs'][l]
        kink = np.abs(z) < 1e-5 * np.abs(z).max()
        assert kink.sum() <= max(2, 1e-3 * kink.size)
        if kink.any():
            S = cfg.D >> (l + 1)
            Cg = eng.ws_tensor(B, 'C', (B, S, S, eng.tl.Pp), index=l)[idx].cpu().numpy()[..., :cfg.P].astype(np.float64)
            r = cache['rs'][l]
            r[kink] = np.where(Cg[kink] > 0, np.maximum(r[kink], 1e-30), 0.0)
            n += int(kink.sum())
    return n


def _cfg4():
    cfg = CFFMConfig(M=1000000, F=32, K=64, D=64, activation='relu')
    p32 = init_params(cfg, seed=5)
    rng = np.random.default_rng(3)
    # trained-like magnitudes (feature_bias is exactly 0 at init, CFFM.py:276)
    p32['feature_bias'] = (rng.standard_normal(p32['feature_bias'].shape) * 0.3).astype(np.float32)
    p32['outer_embeddings'] = (p32['outer_embeddings'] * 20.0).astype(np.float32)
    p32['inner_embeddings'] = (p32['inner_embeddings'] * 4.0).astype(np.float32)
    return cfg, p32


def test_cfg4_backward_rows_at_full_size():
    """B = 8192 uniform ids: out, L, and the per-example gradients of rows {0, 1, B/2, B-1} against the oracle run on those
    rows with the device batch's L."""
    from cffm_amd.engine import HipEngine
    cfg, p32 = _cfg4()
    B = 8192
    eng = HipEngine(cfg, params=p32)
    X, y = synth.batches(cfg.M, cfg.F, B, 1, seed=9)
    X, y = X[0], y[0]
    ids, yt = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    eng.forward(ids, yt)
    torch.cuda.synchronize()
    rows = [0, 1, B // 2, B - 1]
    p64 = to64(p32)
    out_ref, c = orc.forward(p64, X[rows], cfg)
    out_dev = eng.ws_tensor(B, 'out', (B,)).cpu().numpy().astype(np.float64)
    close(out_dev[rows], out_ref, 'out rows')
    n_kink = _adopt_kinks_rows(cfg, eng, B, c, rows)
    print('cfg4 rows: %d relu decisions adopted from the device' % n_kink)
    eng.backward(yt, B)
    torch.cuda.synchronize()
    L_dev = float(eng.ws_tensor(B, 'scalars', (16,)).cpu().numpy()[1])
    L_ref = np.sqrt(np.mean((y.astype(np.float64) - out_dev) ** 2) + 1e-10)     # CFFM.py:493 on the device's outputs
    assert abs(L_dev - L_ref) <= 1e-5 * L_ref, (L_dev, L_ref)
    dout = (out_ref - y[rows].astype(np.float64)) / (B * L_ref)
    g = orc.backward(p64, c, dout, cfg)
    idx = torch.as_tensor(rows, device='cuda')
    Pp = eng.tl.Pp
    close(eng.ws_tensor(B, 'dout', (B,))[idx].cpu().numpy(), dout, 'dout rows')
    close(eng.ws_tensor(B, 'dt1', (B, 2 * cfg.D - 2))[idx].cpu().numpy(), g['_dt1'], 'dt1 rows')
    for l in range(cfg.live_layers - 1, -1, -1):
        S = cfg.D >> (l + 1)
        got = eng.ws_tensor(B, 'dC', (B, S, S, Pp), index=l)[idx].cpu().numpy()
        close(got, pad_channels(g['_dC'][l], Pp), 'dC[%d] rows' % l)
    close(eng.ws_tensor(B, 'dEo', (B, cfg.F, cfg.D))[idx].cpu().numpy(), g['d_outer_rows'], 'dEo rows')
    close(eng.ws_tensor(B, 'dEi', (B, cfg.F, cfg.K))[idx].cpu().numpy(), g['d_inner_rows'], 'dEi rows')
    close(eng.ws_tensor(B, 'dfb', (B, cfg.F))[idx].cpu().numpy(), g['d_bias_rows'], 'dfb rows')


def test_cfg4_train_step_at_full_size_by_linearity():
    """B = 8192 built from 128 distinct examples x 64 copies in shuffled order, labels per row: every dense gradient, the
    loss, and the complete Adagrad step (dense parameters, the three tables and all accumulators) against ONE oracle pass
    over the 128 distinct rows with dL/dout summed over the copies (gradients are linear in dL/dout)."""
    from cffm_amd.engine import HipEngine
    cfg, p32 = _cfg4()
    B, n = 8192, 128
    eng = HipEngine(cfg, params=p32)
    rng = np.random.default_rng(17)
    Xd = synth.sample_ids(rng, cfg.M, cfg.F, n)
    src = rng.permutation(np.repeat(np.arange(n), B // n))
    X = Xd[src]
    y = synth.sample_labels(rng, B)
    ids, yt = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    eng.forward(ids, yt)
    torch.cuda.synchronize()
    p64 = to64(p32)
    out_d, c = orc.forward(p64, Xd, cfg)
    out_dev = eng.ws_tensor(B, 'out', (B,)).cpu().numpy().astype(np.float64)
    close(out_dev, out_d[src], 'out (all copies)')
    first = np.array([int(np.nonzero(src == e)[0][0]) for e in range(n)])
    # copies of one example are computed identically wherever they sit in the batch
    Pp = eng.tl.Pp
    C0 = eng.ws_tensor(B, 'C', (B, cfg.D // 2, cfg.D // 2, Pp), index=0)
    last = np.array([int(np.nonzero(src == e)[0][-1]) for e in range(n)])
    assert torch.equal(C0[torch.as_tensor(first[:8], device='cuda')], C0[torch.as_tensor(last[:8], device='cuda')])
    print('cfg4 tiled: %d relu decisions adopted from the device' % _adopt_kinks_rows(cfg, eng, B, c, first))
    yd = y.astype(np.float64)
    L = np.sqrt(np.mean((yd - out_d[src]) ** 2) + 1e-10)                  # CFFM.py:493 over the 8192 rows
    dout_full = (out_d[src] - yd) / (B * L)
    dout_eff = np.zeros(n)
    np.add.at(dout_eff, src, dout_full)
    g = orc.backward(p64, c, dout_eff, cfg)
    del c
    eng.backward(yt, B)
    torch.cuda.synchronize()
    close(eng.ws_tensor(B, 'scalars', (16,)).cpu().numpy()[1:2], [L], 'loss')
    got = eng.export_grad()
    for k, v in got.items():
        close(v, np.asarray(g[k]).reshape(v.shape), 'grad ' + k)
    # ---- the step: Adagrad over dense parameters, duplicates-summed-first over the tables (CFFM.py:523-524)
    loss = eng.train_step(ids, yt)
    torch.cuda.synchronize()
    close(loss.cpu().numpy(), [L], 'loss (train_step)')
    acc = orc.init_accumulators(p64)
    idsd = Xd.reshape(-1)
    u = lambda t: cfg.lr * t / np.sqrt(1e-8 + t * t)
    gp, ga = eng.export_params(), eng.export_accumulators()
    for k, gk in g.items():
        if k.startswith('d_') or k.startswith('_'):
            continue
        gk = np.asarray(gk).reshape(np.shape(p64[k]))
        a = acc[k] + gk * gk
        ref = p64[k] - cfg.lr * gk / np.sqrt(a)
        dg = 1e-5 * (np.abs(gk) + max(float(np.sqrt(np.mean(gk * gk))), 1e-30))
        extra = np.maximum(np.abs(u(gk + dg) - u(gk)), np.abs(u(gk - dg) - u(gk)))
        close(gp[k], ref.reshape(gp[k].shape), 'param ' + k, tol=2e-5, extra=extra.reshape(gp[k].shape))
        close(ga[k], a.reshape(ga[k].shape), 'acc ' + k, tol=2e-5, extra=(2 * np.abs(gk) * dg + dg * dg).reshape(ga[k].shape))
    uniq = np.unique(idsd)
    for tname, key in (('inner_embeddings', 'd_inner_rows'), ('outer_embeddings', 'd_outer_rows'), ('feature_bias', 'd_bias_rows')):
        rows = g[key].reshape(idsd.shape[0], -1)
        t, ta = np.zeros(p64[tname].shape), np.zeros(p64[tname].shape)
        np.add.at(t, idsd, rows)
        np.add.at(ta, idsd, np.abs(rows) + float(np.sqrt(np.mean(rows * rows))))
        orc.adagrad_sparse(p64[tname], acc[tname], idsd, g[key], cfg.lr)
        gk, dg = t[uniq], 1e-5 * ta[uniq]
        extra = np.maximum(np.abs(u(gk + dg) - u(gk)), np.abs(u(gk - dg) - u(gk)))
        close(gp[tname][uniq], p64[tname][uniq], 'rows of ' + tname, tol=2e-5, extra=extra)
        close(ga[tname][uniq], acc[tname][uniq], 'acc rows of ' + tname, tol=2e-5, extra=2 * np.abs(gk) * dg + dg * dg)
        mask = np.ones(cfg.M, dtype=bool)
        mask[uniq] = False
        np.testing.assert_array_equal(gp[tname][mask], p32[tname][mask])
        assert np.all(ga[tname][mask] == np.float32(1e-8))


@pytest.fixture(scope='module')
def nccl_world1():
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29541')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    yield
    if created:
        dist.destroy_process_group()


@pytest.mark.parametrize('dedup', [True, False])
def test_sharded_step_world1_equals_oracle_step(nccl_world1, dedup):
    """ShardedStep end to end on one GPU (RCCL loopback): distinct-id exchange, packed records, duplicates re-expanded by
    cffm_stage_packed and summed again by cffm_pack_rows_dedup, against ONE oracle step - heavy duplication (ids drawn
    from 40 values per column), two steps with the second batch's plan issued one step ahead."""
    from cffm_amd.dist import ShardedStep
    from cffm_amd.engine import HipEngine
    from tests.test_gpu_parity import adagrad_step_slack, make_case, oracle_dense_grads
    cfg, p32, X, y = make_case('bookx-relu')
    rng = np.random.default_rng(4)
    X = rng.integers(0, 40, size=X.shape).astype(np.int32) * 70
    X2 = rng.integers(0, 40, size=X.shape).astype(np.int32) * 70
    eng = HipEngine(cfg, params=p32)
    sh = ShardedStep(eng, dedup=dedup)
    a, b, yt = torch.from_numpy(X).cuda(), torch.from_numpy(X2).cuda(), torch.from_numpy(y).cuda()
    l1 = float(sh.train_step(a, yt, next_ids=b).cpu()[0])
    assert sh._ahead is not None
    l2 = float(sh.train_step(b, yt).cpu()[0])
    torch.cuda.synchronize()
    p64 = to64(p32)
    acc = orc.init_accumulators(p64)
    slack = {}
    Ls = []
    for Xs in (X, X2):
        g = oracle_dense_grads(p64, Xs, y, cfg)
        for k, v in adagrad_step_slack(g, acc, cfg.lr, rel=2e-5).items():
            slack[k] = slack.get(k, 0.0) + v
        Ls.append(orc.train_step(p64, acc, Xs, y.astype(np.float64), cfg)[0])
    assert abs(l1 - Ls[0]) <= 1e-5 * Ls[0] and abs(l2 - Ls[1]) <= 2e-5 * Ls[1]
    got = eng.export_params()
    for k, v in got.items():
        close(v, p64[k].reshape(v.shape), 'sharded world-1 param ' + k, tol=2e-5,
              extra=None if k not in slack else slack[k].reshape(v.shape))
    touched = np.zeros(cfg.M, dtype=bool)
    touched[X.reshape(-1)] = True
    touched[X2.reshape(-1)] = True
    for k in ('inner_embeddings', 'outer_embeddings', 'feature_bias'):
        np.testing.assert_array_equal(got[k][~touched], p32[k][~touched])


def test_cfg5_share_row_sharded_world1(nccl_world1):
    """One GPU's share of BASELINE.json configs[4]: 10 M features, 32 fields, dim 64, 8192 rows per GPU through
    ShardedStep (row-sharded tables, the three all-to-alls and the all-reduce over RCCL loopback at world size 1).
    Forward outputs and per-example row gradients of 4 examples against the oracle (on the table rows those examples
    touch, read before the step), and the invariants of the owner-side update."""
    from cffm_amd.dist import ShardedStep
    from cffm_amd.engine import HipEngine
    if True:
        cfg = CFFMConfig(M=10000000, F=32, K=64, D=64, activation='relu')
        B = 8192
        eng = HipEngine(cfg, params='device', seed=2021)
        eng.fbias.normal_(0.0, 0.3, generator=torch.Generator(device='cuda').manual_seed(1))
        eng.outer.mul_(20.0)
        eng.inner.mul_(4.0)
        X, y = synth.batches(cfg.M, cfg.F, B, 1, seed=11)
        X, y = X[0], y[0]
        ids, yt = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
        rows = [0, 1, B // 2, B - 1]
        # compact parameter set for the oracle: the table rows of the 4 examples, read BEFORE the step
        uniq, inv = np.unique(X[rows].reshape(-1), return_inverse=True)
        ut = torch.from_numpy(uniq).cuda().long()
        dense = eng.export_params_dense()
        p64 = to64(dense)
        p64['inner_embeddings'] = eng.inner[ut].cpu().numpy().astype(np.float64)
        p64['outer_embeddings'] = eng.outer[ut].cpu().numpy().astype(np.float64)
        p64['feature_bias'] = eng.fbias[ut].cpu().numpy().astype(np.float64).reshape(-1, 1)
        Xs = inv.reshape(len(rows), cfg.F)
        touched = torch.zeros(cfg.M, dtype=torch.bool, device='cuda')
        touched[ids.reshape(-1).long()] = True
        probe = torch.randint(0, cfg.M, (4096,), device='cuda')
        probe = probe[~touched[probe]]
        before = (eng.inner[probe].clone(), eng.outer[probe].clone(), eng.fbias[probe].clone())
        hit = ids.reshape(-1).long()[:4096]
        hit_before = eng.outer[hit].clone()

        sh = ShardedStep(eng)
        loss = float(sh.train_step(ids, yt).cpu()[0])
        torch.cuda.synchronize()

        out_ref, c = orc.forward(p64, Xs, cfg)
        out_dev = eng.ws_tensor(B, 'out', (B,)).cpu().numpy().astype(np.float64)
        close(out_dev[rows], out_ref, 'out rows (cfg5 share)')
        L_ref = np.sqrt(np.mean((y.astype(np.float64) - out_dev) ** 2) + 1e-10)
        assert abs(loss - L_ref) <= 1e-5 * L_ref, (loss, L_ref)
        _adopt_kinks_rows(cfg, eng, B, c, rows)
        # ShardedStep runs the backward WITHOUT 1/L (it is applied to the summed gradients in the update): dL/dout = (out-y)/Bg
        dout = (out_ref - y[rows].astype(np.float64)) / B
        g = orc.backward(p64, c, dout, cfg)
        idx = torch.as_tensor(rows, device='cuda')
        close(eng.ws_tensor(B, 'dEo', (B, cfg.F, cfg.D))[idx].cpu().numpy(), g['d_outer_rows'], 'dEo rows (cfg5 share)')
        close(eng.ws_tensor(B, 'dEi', (B, cfg.F, cfg.K))[idx].cpu().numpy(), g['d_inner_rows'], 'dEi rows (cfg5 share)')
        close(eng.ws_tensor(B, 'dfb', (B, cfg.F))[idx].cpu().numpy(), g['d_bias_rows'], 'dfb rows (cfg5 share)')
        # update invariants: rows nobody looked up are bit-identical with accumulators at 1e-8, looked-up rows all moved
        assert torch.equal(eng.inner[probe], before[0]) and torch.equal(eng.outer[probe], before[1])
        assert torch.equal(eng.fbias[probe], before[2])
        assert bool((eng.inner_acc[probe] == 1e-8).all()) and bool((eng.outer_acc[probe] == 1e-8).all())
        assert bool((eng.outer_acc[hit] > 1e-8).any(dim=1).all())
        assert bool((eng.outer[hit] != hit_before).any(dim=1).all())
        assert np.isfinite(eng.predict(ids[:64]).cpu().numpy()).all()


def test_sharded_checkpoint_round_trip(tmp_path):
    """N3, sharded-table aware: two 'ranks' (row shards r % 2) save their shard + slots, fresh engines restore them."""
    import copy
    from cffm_amd.dist import load_sharded, local_rows_count, save_sharded, shard_params
    from cffm_amd.engine import HipEngine
    from tests.test_gpu_parity import make_case
    cfg, p32, X, y = make_case('bookx-relu')
    G = 2
    path = str(tmp_path / 'ckpt')
    engines = []
    for r in range(G):
        lc = copy.copy(cfg)
        lc.M = local_rows_count(cfg.M, r, G)
        e = HipEngine(lc, params=shard_params(p32, r, G))
        Xl = torch.from_numpy((X // G).astype(np.int32) % lc.M).cuda()
        e.train_step(Xl, torch.from_numpy(y).cuda())             # non-trivial accumulators
        torch.cuda.synchronize()
        save_sharded(e, path, r, G, opt_step=1)
        engines.append((lc, e))
    for r, (lc, e) in enumerate(engines):
        f = HipEngine(lc, seed=99)                                 # different random state, then restored
        assert load_sharded(f, path, r, G) == 1
        # tables and their slots come from the rank's own shard; the replicated dense parameters from rank 0's file (the
        # two stand-in ranks of this test trained on their own, so their dense parameters differ - a real run keeps them
        # bit-identical)
        tables = ('inner_embeddings', 'outer_embeddings', 'feature_bias')
        e0 = engines[0][1]
        for exp in ('export_params', 'export_accumulators'):
            a, a0, b = getattr(e, exp)(), getattr(e0, exp)(), getattr(f, exp)()
            for k in a:
                np.testing.assert_array_equal((a if k in tables else a0)[k], b[k], err_msg='rank %d %s %s' % (r, exp, k))
    with pytest.raises(ValueError):
        load_sharded(engines[0][1], path, 1, G) if engines[0][0].M != engines[1][0].M else (_ for _ in ()).throw(ValueError('same size'))


def test_graph_replay_survives_a_workspace_regrowth(nccl_world1):
    """DataParallelStep(use_graph=True) bakes the workspace address into its captured kernels.  A later, larger request
    (evaluate()'s 8192-row blocks) re-allocates the workspace: the engine must keep the outgrown buffer alive while a graph
    is pinned to it, and the step must re-capture against the new one - an eager twin that never used a graph stays
    bit-identical through the whole sequence."""
    from cffm_amd.dist import DataParallelStep
    from cffm_amd.engine import HipEngine
    from tests.test_gpu_parity import make_case
    cfg, p32, X, y = make_case('bookx-relu')
    rng = np.random.default_rng(9)
    batches = [rng.integers(0, cfg.M, size=X.shape).astype(np.int32) for _ in range(7)]
    big = torch.from_numpy(rng.integers(0, cfg.M, size=(8192, cfg.F)).astype(np.int32)).cuda()
    yt = torch.from_numpy(y).cuda()
    eng_g, eng_e = HipEngine(cfg, params=p32), HipEngine(cfg, params=p32)
    dp_g, dp_e = DataParallelStep(eng_g, use_graph=True), DataParallelStep(eng_e, use_graph=False)
    for i, Xb in enumerate(batches):
        ids = torch.from_numpy(Xb).cuda()
        lg, le = dp_g.train_step(ids, yt), dp_e.train_step(ids, yt)
        torch.cuda.synchronize()
        assert float(lg) == float(le), i
        if i == 3:                                    # a graph exists by now (captured at the third call)
            st = next(iter(dp_g._graphs.values()))
            assert st['graph'] is not None and eng_g._pool.pins == {eng_g.ws_generation: 1}
            gen = eng_g.ws_generation
            old = eng_g._pool.buf
            pg, pe = eng_g.predict(big), eng_e.predict(big)          # outgrows the captured workspace
            torch.cuda.synchronize()
            assert eng_g.ws_generation == gen + 1 and eng_g._pool.retired[gen] is old
            assert torch.equal(pg, pe)
    st = next(iter(dp_g._graphs.values()))
    assert st['graph'] is not None and st['ws_gen'] == eng_g.ws_generation      # re-captured against the new buffer
    assert not eng_g._pool.retired and eng_g._pool.pins == {eng_g.ws_generation: 1}  # and the outgrown buffer is freed
    a, b = eng_g.export_params(), eng_e.export_params()
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
