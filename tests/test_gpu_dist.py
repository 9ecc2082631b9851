"""N > 1 path ON THE GPU: two ranks on cuda:0 over gloo (a one-GPU box cannot give RCCL two devices, and gloo moves device
tensors by itself), so that the multi-rank code of cffm_amd.dist runs against the real HIP kernels before an 8-GPU node sees
it: DataParallelStep (dense-image route and all-gather route) and ShardedStep at world size 2 must leave every rank with
what ONE HipEngine gets from the whole batch.  The collectives differ from the product's only in the backend name.

What is compared: the state after each of two consecutive train steps.  In the FIRST step every example is computed by the same
kernels from the same parameters in both runs, so the forward is bit-identical and the gradients differ by summation order
only (per-rank loss sums, slab grouping, duplicates of an id across ranks: a few 1e-7 relative): 1e-4 relative / 2e-6 absolute,
four orders of magnitude below what a wrong 1/B, a lost rank or a mis-routed row would produce.  The SECOND step starts from
states that already differ in the last bits, and Adagrad from 1e-8 accumulators amplifies that (free-running
trajectories are not comparable element by element): 90 % of a tensor's elements within 2e-2 relative / 2e-4 absolute, which
shows that the step runs on updated accumulators and re-used plans and still excludes a wrong scale."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cffm_amd.spec import CFFMConfig, init_params  # noqa: E402
from tests import test_dist_cpu as H  # noqa: E402   (the spawn / gloo harness)

pytestmark = pytest.mark.gpu
TABLES = ('inner_embeddings', 'outer_embeddings', 'feature_bias')


def _case(wide=False):
    cfg = CFFMConfig(M=800, F=10, K=32, D=32, activation='selu')            # the frappe command's shape, small vocabulary
    if wide:      # F = 16: 120 pairs > 64 channels - the wide-filter kernels, whose row-sharded step consumes the received records in place
        cfg = CFFMConfig(M=900, F=16, K=32, D=32, activation='relu')
    rng = np.random.default_rng(11)
    X = rng.integers(0, cfg.M, size=(2, 64 if wide else 128, cfg.F)).astype(np.int32)
    X[:, -20] = X[:, 3]                                                      # the same ids on both ranks: cross-rank duplicates
    y = rng.choice([-1.0, 1.0], size=X.shape[:2]).astype(np.float32)
    return cfg, X, y


def _dp_worker(rank, world, mode):
    from cffm_amd.dist import DataParallelStep
    from cffm_amd.engine import HipEngine
    cfg, X, y = _case()
    eng = HipEngine(cfg, params=init_params(cfg, seed=7 + rank), device='cuda:0')   # different draws: rank 0's must win
    dp = DataParallelStep(eng, mode=mode)
    per = X.shape[1] // world
    sl = slice(rank * per, rank * per + per)
    out = []
    for s in range(X.shape[0]):
        loss = dp.train_step(torch.from_numpy(X[s, sl].copy()).cuda(), torch.from_numpy(y[s, sl].copy()).cuda())
        torch.cuda.synchronize()
        out.append((float(loss.cpu().reshape(-1)[0]), eng.export_params(), eng.export_accumulators()))
    return out


def _sharded_worker(rank, world, ahead, wide=False):
    from cffm_amd.dist import ShardedStep, local_rows_count, shard_params
    from cffm_amd.engine import HipEngine
    cfg, X, y = _case(wide)
    lcfg = copy.copy(cfg)
    lcfg.M = local_rows_count(cfg.M, rank, world)
    eng = HipEngine(lcfg, params=shard_params(init_params(cfg, seed=7), rank, world), device='cuda:0')
    sh = ShardedStep(eng)
    per = X.shape[1] // world
    sl = slice(rank * per, rank * per + per)
    assert eng.packed_ok() == wide
    staged, stage = [], eng.stage_packed           # wide: the step must not stage the rows (they are read out of the records)
    eng.stage_packed = lambda *a: (staged.append(1), stage(*a))[1]
    ids = [torch.from_numpy(X[s, sl].copy()).cuda() for s in range(X.shape[0])]
    ys = [torch.from_numpy(y[s, sl].copy()).cuda() for s in range(X.shape[0])]
    out = []
    for s in range(len(ids)):
        nxt = ids[s + 1] if ahead and s + 1 < len(ids) else None
        loss = sh.train_step(ids[s], ys[s], next_ids=nxt)
        torch.cuda.synchronize()
        out.append((float(loss.cpu().reshape(-1)[0]), eng.export_params(), None))
        assert bool(staged) != wide
    return out, (sh.plans_built, sh.plans_reused)


def _single(wide=False):
    from cffm_amd.engine import HipEngine
    cfg, X, y = _case(wide)
    eng = HipEngine(cfg, params=init_params(cfg, seed=7), device='cuda:0')
    out = []
    for s in range(X.shape[0]):
        loss = eng.train_step(torch.from_numpy(X[s]).cuda(), torch.from_numpy(y[s]).cuda())
        torch.cuda.synchronize()
        out.append((float(loss.cpu().reshape(-1)[0]), eng.export_params(), eng.export_accumulators()))
    return out


def _same(a, b, what, step):
    """Step 0: 99.8 % of the elements within 1e-4 relative / 2e-6 absolute and EVERY element within 5e-4 * (1 + |ref|).  (The handful
    beyond the first bound are parameters whose gradient nearly cancels: the first Adagrad step from a 1e-8 accumulator,
    u(g) = lr*g/sqrt(1e-8 + g^2), has slope lr*1e4 at g = 0, so one ulp of a 0.5-sized partial sum taken in another order
    moves the parameter by 3e-5.  A wrong scale moves EVERY updated element by a fraction of lr = 0.05.)
    Step 1 (a sanity check of a free-running second step, see the header): 90 % within 2e-2 / 2e-4, every element within
    0.2 * (1 + |ref|)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    rtol, atol, frac, hard = [(1e-4, 2e-6, 0.998, 5e-4), (2e-2, 2e-4, 0.90, 0.2)][step]
    err = np.abs(a - b)
    ok = err <= atol + rtol * np.abs(b)
    worst = float((err / (1.0 + np.abs(b))).max())            # the hard bound is relative for large values (accumulators)
    assert ok.mean() >= frac and worst <= hard, 'step %d: %s: %.4f %% within tolerance, worst error/(1+|ref|) %.3g' % (
        step, what, 100 * ok.mean(), worst)


@pytest.mark.parametrize('mode', ['dense', 'gather'])
def test_data_parallel_world2_on_the_gpu_equals_one_engine(mode):
    res = H._run(_dp_worker, 2, mode)
    ref = _single()
    for step, (L, p, acc) in enumerate(ref):
        for rank in (0, 1):
            loss, got, gacc = res[rank][step]
            _same(loss, L, 'rank %d loss' % rank, step)
            for k in acc:        # the TRAINED variables (outer_W / outer_b and the dead last conv layer stay on the host, untouched
                _same(got[k], p[k], 'rank %d %s' % (rank, k), step)                      # by the step and by the broadcast)
                _same(gacc[k], acc[k], 'rank %d accumulator of %s' % (rank, k), step)
        for k in acc:                                                        # the replicas stay bit-identical
            np.testing.assert_array_equal(np.asarray(res[0][step][1][k]), np.asarray(res[1][step][1][k]), err_msg=k)


@pytest.mark.parametrize('ahead,wide', [(False, False), (True, False), (True, True)])
def test_row_sharded_world2_on_the_gpu_equals_one_engine(ahead, wide):
    """wide: the wide-filter shapes run the row-sharded step WITHOUT staging (cffm_forward_packed / cffm_backward_unscaled_packed:
    the received records are read in place, by record index) - the same non-materialising kernels as the replicated tables."""
    res = H._run(_sharded_worker, 2, ahead, wide)
    ref = _single(wide)
    for rank in (0, 1):
        steps, (built, reused) = res[rank]
        assert reused == (1 if ahead else 0) and built == 2
        for step, (L, p, acc) in enumerate(ref):
            loss, got, _ = steps[step]
            _same(loss, L, 'rank %d loss' % rank, step)
            for k in acc:
                want = np.asarray(p[k])[rank::2] if k in TABLES else p[k]
                _same(got[k], want, 'rank %d %s' % (rank, k), step)


def _cffm_class_worker(rank, world, tmp):
    """The drop-in class with its real HipEngine under a process group the caller initialised (tests/test_dist_cpu.py runs the
    same flow with the oracle standing in for the engine)."""
    from cffm_amd import CFFM as M
    from cffm_amd import synth

    class Split(dict):
        pass

    rng = np.random.default_rng(11)
    Mf, F = 40, 4

    def split(n):
        return Split(X=synth.sample_ids(rng, Mf, F, n).tolist(), Y=synth.sample_labels(rng, n).tolist())

    class Data(object):
        pass
    data = Data()
    data.Train_data, data.Validation_data, data.Test_data = split(37), split(11), split(9)
    m = M.CFFM(Mf, 0, os.path.join(tmp, 'g%d_w%d' % (rank, world)), 8, 8, 'square_loss', 2, 8, 0.05, 0, [1.0, 1.0],
               'AdagradOptimizer', 0, 0, 0, F, 1, 0, 1.0, 1, 1.0, 1, 1.0, 'relu')
    np.random.seed(77)                                       # the reference's batch starts are unseeded: pin them for the test
    m.train(data)
    assert m.world == world and (m._dp is not None) == (world > 1)
    return (m.train_rmse, m.valid_rmse, m.test_rmse, m.train_r2), m.engine.export_params()


def test_cffm_class_trains_data_parallel_on_the_gpu(tmp_path):
    one = H._run(_cffm_class_worker, 1, str(tmp_path))[0]
    two = H._run(_cffm_class_worker, 2, str(tmp_path))
    for a, b in zip(two[0][0], two[1][0]):                   # per-epoch metrics: the SAME numbers on both ranks ...
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
    for k in two[0][1]:                                      # ... from bit-identical replicas
        np.testing.assert_array_equal(np.asarray(two[0][1][k]), np.asarray(two[1][1][k]), err_msg=k)
    for a, b in zip(one[0], two[0][0]):                      # and the single-process run's, up to fp32 summation order over two
        np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=2e-2, atol=2e-3)   # free-running epochs (header)


@pytest.mark.parametrize('n_rows,F,M,world,id_range', [(8192, 32, 10_000_000, 8, None), (300, 10, 5382, 2, 40), (1, 3, 17, 4, None),
                                                      (4096, 6, 226336, 3, 500), (70, 16, 3000, 1, None)])
def test_device_shard_plan_equals_the_torch_plan(n_rows, F, M, world, id_range):
    """cffm_shard_plan (pack -> one radix sort -> head flags -> scan -> scatter, csrc/plan.hip) against ShardedStep.plan_torch, the ~15
    torch operations it replaces and its specification: every output identical - local rows, the stable (owner, local row) order
    with slots ascending inside a run, the distinct-pair index of every sorted position and of every slot, the request list and the
    per-owner counts - at cfg5's per-GPU share (8192 x 32 lookups, 10 M features, 8 owners), with heavy duplication, with a single
    row, with a world size that is not a power of two and at world size 1."""
    from cffm_amd.dist import ShardedStep
    from cffm_amd.engine import HipEngine
    rng = np.random.default_rng(n_rows + world)
    X = rng.integers(0, id_range or M, size=(n_rows, F)).astype(np.int32)
    if id_range:
        X = (X.astype(np.int64) * (M // id_range)).astype(np.int32)        # few distinct values, spread over the owners
    X[-1, 0] = M - 1                                                        # the last row of the vocabulary
    ids = torch.from_numpy(X).cuda()
    eng = HipEngine(CFFMConfig(M=64, F=F, K=8, D=8), device='cuda:0')       # the plan depends on the ids only
    st = ShardedStep.__new__(ShardedStep)
    st.world, st.dedup = world, True
    want = st.plan_torch(ids)
    got = eng.shard_plan(ids, world, M)
    torch.cuda.synchronize()
    n_distinct = int(want[5].sum())
    names = ('local_ids', 'order', 'uniq', 'pos', 'send_rows', 'counts')
    for name, a, b in zip(names, got, want):
        assert a.dtype == b.dtype and a.shape == b.shape, name
        if name == 'send_rows':                                             # capacity n, the first #distinct entries are defined
            a, b = a[:n_distinct], b[:n_distinct]
        assert torch.equal(a, b), name
    assert n_distinct == int(got[2].max()) + 1 and int(got[5].sum()) == n_distinct
    # and twice in a row on the same scratch (the counts are re-zeroed by the call itself)
    again = eng.shard_plan(ids, world, M)
    assert torch.equal(again[5], want[5]) and torch.equal(again[3], want[3])
