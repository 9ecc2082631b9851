"""N > 1 path on CPU: world_size-2 gloo runs of cffm_amd.dist.  The collectives, routing and dedup logic are
the product's; the per-rank compute is stood in by the float64 oracle (test infrastructure), so the check
is exact: a 2-rank data-parallel step on two half batches == one oracle step on the whole batch."""
import os
import socket
import sys
import traceback

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cffm_amd.spec import CFFMConfig, init_params  # noqa: E402
from oracle import cffm_oracle as orc  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(fn, world, *args):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_entry, args=(fn, r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == 'ok', r[2]
    return {r[0]: r[2] for r in res}


def _entry(fn, rank, world, port, q, *args):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.set_num_threads(1)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        out = fn(rank, world, *args)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, 'ok', out))
    except Exception:
        q.put((rank, 'fail', traceback.format_exc()))


class OracleCompute(object):
    """HipEngine's step-half interface on CPU tensors, computed by the float64 oracle."""

    def __init__(self, cfg, p):
        self.cfg, self.p = cfg, p
        self.acc = orc.init_accumulators(p)
        self.names = [k for k in orc.DENSE_TRAINED] + \
                     ['outer_layer_conv_%s_%d' % (w, l) for l in range(cfg.Lc - 1) for w in ('weight', 'bias')]
        self.sizes = [int(np.prod(p[k].shape)) if p[k].shape else 1 for k in self.names]
        self.grad = torch.zeros(sum(self.sizes), dtype=torch.float64)
        self.sc = torch.zeros(16, dtype=torch.float64)

    def forward(self, ids, y):
        self.X = ids.numpy()
        self.out, self.cache = orc.forward(self.p, self.X, self.cfg)
        self.sc[0] = float(np.sum((y.numpy() - self.out) ** 2))

    def backward_unscaled(self, ids, y, B, Bg):
        dout = (self.out - y.numpy()) / Bg                       # no 1/L yet
        self.g = orc.backward(self.p, self.cache, dout, self.cfg)
        flat = np.concatenate([np.asarray(self.g[k]).reshape(-1) for k in self.names] + [[float(self.sc[0])], [0.0] * 3])
        X = self.X.reshape(-1, 1).astype(np.float64)            # id column (the product packs the int32 bits)
        rows = np.concatenate([X, self.g['d_inner_rows'].reshape(len(X), -1), self.g['d_outer_rows'].reshape(len(X), -1),
                               self.g['d_bias_rows'].reshape(len(X), 1)], axis=1)
        return torch.from_numpy(flat), torch.from_numpy(np.ascontiguousarray(rows))

    def dp_apply(self, grad, rows_all, Bg):
        g = grad.numpy()
        n = sum(self.sizes)
        L = np.sqrt(g[n] / Bg + 1e-10)
        flat, o = g[:n] / L, 0
        for k, sz in zip(self.names, self.sizes):
            gk = flat[o:o + sz].reshape(self.p[k].shape)
            o += sz
            if self.p[k].shape == ():
                a = self.acc[k] + gk * gk
                self.acc[k] = a
                self.p[k] = self.p[k] - self.cfg.lr * gk / np.sqrt(a)
            else:
                orc.adagrad_dense(self.p[k], self.acc[k], gk, self.cfg.lr)
        r = rows_all.numpy()
        ids = r[:, 0].astype(np.int64)
        K, D = self.cfg.K, self.cfg.D
        orc.adagrad_sparse(self.p['inner_embeddings'], self.acc['inner_embeddings'], ids, r[:, 1:1 + K] / L, self.cfg.lr)
        orc.adagrad_sparse(self.p['outer_embeddings'], self.acc['outer_embeddings'], ids, r[:, 1 + K:1 + K + D] / L, self.cfg.lr)
        orc.adagrad_sparse(self.p['feature_bias'], self.acc['feature_bias'], ids, r[:, 1 + K + D:] / L, self.cfg.lr)
        return torch.tensor([L])


class DenseOracleCompute(OracleCompute):
    """The dense-image route of DataParallelStep (small vocabularies): one flat buffer, one all-reduce."""

    def dp_dense_ok(self, B, world):
        return True

    def dp_local_dense(self, ids, y, B, Bg):
        self.forward(ids, y)
        dout = (self.out - y.numpy()) / Bg
        g = orc.backward(self.p, self.cache, dout, self.cfg)
        X = self.X.reshape(-1)
        parts = [np.asarray(g[k]).reshape(-1) for k in self.names] + [[float(self.sc[0])]]
        for name, key in (('inner_embeddings', 'd_inner_rows'), ('outer_embeddings', 'd_outer_rows'), ('feature_bias', 'd_bias_rows')):
            t = np.zeros(self.p[name].shape)
            np.add.at(t, X, g[key].reshape(len(X), -1).reshape((len(X),) + self.p[name].shape[1:]))
            parts.append(t.reshape(-1))
        return torch.from_numpy(np.concatenate(parts))

    def dp_apply_dense(self, flat, Bg):
        f = flat.numpy()
        n = sum(self.sizes)
        L = np.sqrt(f[n] / Bg + 1e-10)
        o = 0
        for k, sz in zip(self.names, self.sizes):
            gk = (f[o:o + sz] / L).reshape(self.p[k].shape)
            o += sz
            a = self.acc[k] + gk * gk
            self.acc[k] = a
            self.p[k] = self.p[k] - self.cfg.lr * gk / np.sqrt(a)
        o = n + 1
        for name in ('inner_embeddings', 'outer_embeddings', 'feature_bias'):
            sz = self.p[name].size
            gk = (f[o:o + sz] / L).reshape(self.p[name].shape)
            o += sz
            a = self.acc[name] + gk * gk                       # rows with gk == 0 keep acc and value
            self.acc[name] = a
            self.p[name] = self.p[name] - self.cfg.lr * gk / np.sqrt(a)
        return torch.tensor([L])


class ShardedOracleCompute(OracleCompute):
    """The same stand-in over a LOCAL table shard (rows rank, rank+G, ...) for ShardedStep."""

    def gather_packed(self, local_rows):
        r = local_rows.numpy().astype(np.int64)
        pad = np.zeros((r.shape[0], 3))
        return torch.from_numpy(np.concatenate([self.p['inner_embeddings'][r], self.p['outer_embeddings'][r],
                                                self.p['feature_bias'][r].reshape(-1, 1), pad], axis=1))

    def stage_packed(self, packed, pos, B):
        K, D = self.cfg.K, self.cfg.D
        rec = packed.numpy()[pos.numpy().astype(np.int64)]           # duplicates of an id share one record
        assert rec.shape == (B * self.cfg.F, K + D + 4) and np.all(rec[:, K + D + 1:] == 0)
        self._staged = (rec[:, :K].copy(), rec[:, K:K + D].copy(), rec[:, K + D].copy())

    def forward_staged(self, y, B):
        Ei, Eo, fb = self._staged
        F = self.cfg.F
        q = dict(self.p)                       # the looked-up rows act as a private B*F-row table
        q['inner_embeddings'], q['outer_embeddings'] = Ei, Eo
        q['feature_bias'] = fb.reshape(-1, 1)
        self.X = np.arange(B * F).reshape(B, F)
        self.out, self.cache = orc.forward(q, self.X, self.cfg)
        self._q = q
        self.sc[0] = float(np.sum((y.numpy() - self.out) ** 2))

    def backward_unscaled(self, ids, y, B, Bg, pack=True):
        p, self.p = self.p, self._q
        try:
            self.X_save = self.X
            grad, rows = OracleCompute.backward_unscaled(self, ids, y, B, Bg)
        finally:
            self.p = p
        rows[:, 0] = ids.reshape(-1).to(torch.float64)          # keyed by the owner's local row
        self._rows = rows
        return grad, (rows if pack else None)

    def pack_rows_dedup(self, local_ids, order, uniq, B):
        rows = self._rows.numpy()
        o, u = order.numpy().astype(np.int64), uniq.numpy().astype(np.int64)
        out = np.zeros_like(rows)
        np.add.at(out, u, rows[o])                                  # duplicates summed before they travel
        heads = np.r_[True, u[1:] != u[:-1]]
        out[u[heads], 0] = local_ids.reshape(-1).numpy()[o[heads]]
        return torch.from_numpy(out)


def _case():
    cfg = CFFMConfig(M=50, F=4, K=8, D=8, activation='selu', lamda_att=1.5)
    p = init_params(cfg, seed=3, dtype=np.float64)
    rng = np.random.default_rng(5)
    p['feature_bias'] = rng.standard_normal(p['feature_bias'].shape) * 0.3
    p['outer_embeddings'] *= 20
    X = rng.integers(0, cfg.M, size=(8, cfg.F))
    X[5] = X[0]                                  # the same ids on both ranks: cross-rank duplicates
    y = rng.choice([-1.0, 1.0], size=8)
    return cfg, p, X, y


def _dp_worker(rank, world, dense=False):
    from cffm_amd.dist import DataParallelStep
    cfg, p, X, y = _case()
    comp = (DenseOracleCompute if dense else OracleCompute)(cfg, p)
    step = DataParallelStep(comp)
    per = X.shape[0] // world
    sl = slice(rank * per, rank * per + per)
    loss = step.train_step(torch.from_numpy(X[sl]), torch.from_numpy(y[sl]))
    return float(loss[0]), {k: np.asarray(v) for k, v in comp.p.items()}


@pytest.mark.parametrize('world,dense', [(2, False), (4, False), (2, True)])
def test_data_parallel_step_equals_single_process_step(world, dense):
    res = _run(_dp_worker, world, dense)
    cfg, p, X, y = _case()
    acc = orc.init_accumulators(p)
    L, _ = orc.train_step(p, acc, X, y, cfg)
    for rank in range(world):
        loss, got = res[rank]
        assert abs(loss - L) < 1e-12
        for k, v in got.items():
            np.testing.assert_allclose(v, p[k], rtol=1e-10, atol=1e-12, err_msg='rank %d %s' % (rank, k))
    for k in res[0][1]:                           # replicas stay bit-identical
        for rank in range(1, world):
            np.testing.assert_array_equal(res[0][1][k], res[rank][1][k])


def _sharded_step_worker(rank, world):
    from cffm_amd.dist import ShardedStep, local_rows_count, shard_params
    cfg, p, X, y = _case()
    import copy
    lcfg = copy.copy(cfg)
    lcfg.M = local_rows_count(cfg.M, rank, world)
    comp = ShardedOracleCompute(lcfg, shard_params(p, rank, world))
    assert comp.p['inner_embeddings'].shape[0] == lcfg.M
    step = ShardedStep(comp)
    per = X.shape[0] // world
    sl = slice(rank * per, rank * per + per)
    loss = step.train_step(torch.from_numpy(X[sl]), torch.from_numpy(y[sl]))
    return float(loss[0]), {k: np.asarray(v) for k, v in comp.p.items()}, {k: np.asarray(v) for k, v in comp.acc.items()}


def _sharded_two_steps_worker(rank, world, dedup, ahead):
    """Two steps on two different batches; with ahead=True the second batch's routing plan is issued during the first
    step (the per-owner counts reach the host one step early)."""
    from cffm_amd.dist import ShardedStep, local_rows_count, shard_params
    cfg, p, X, y = _case()
    import copy
    lcfg = copy.copy(cfg)
    lcfg.M = local_rows_count(cfg.M, rank, world)
    comp = ShardedOracleCompute(lcfg, shard_params(p, rank, world))
    step = ShardedStep(comp, dedup=dedup)
    per = X.shape[0] // world
    sl = slice(rank * per, rank * per + per)
    X2 = (X[::-1] * 7 + 3) % cfg.M
    a, b = torch.from_numpy(X[sl].copy()), torch.from_numpy(X2[sl].copy())
    l1 = step.train_step(a, torch.from_numpy(y[sl]), next_ids=b if ahead else None)
    assert (step._ahead is not None) == ahead
    l2 = step.train_step(b, torch.from_numpy(y[sl]))
    return (float(l1[0]), float(l2[0])), {k: np.asarray(v) for k, v in comp.p.items()}


@pytest.mark.parametrize('dedup,ahead', [(True, True), (False, False)])
def test_row_sharded_two_steps_dedup_and_plan_ahead(dedup, ahead):
    world = 2
    res = _run(_sharded_two_steps_worker, world, dedup, ahead)
    cfg, p, X, y = _case()
    X2 = (X[::-1] * 7 + 3) % cfg.M
    acc = orc.init_accumulators(p)
    L1, _ = orc.train_step(p, acc, X, y, cfg)
    L2, _ = orc.train_step(p, acc, X2, y, cfg)
    for rank in range(world):
        (l1, l2), got = res[rank]
        assert abs(l1 - L1) < 1e-12 and abs(l2 - L2) < 1e-10
        for k, v in got.items():
            ref = p[k][rank::world] if k in ('inner_embeddings', 'outer_embeddings', 'feature_bias') else p[k]
            np.testing.assert_allclose(v, ref, rtol=1e-9, atol=1e-11, err_msg='rank %d %s' % (rank, k))


@pytest.mark.parametrize('world', [2, 4])
def test_row_sharded_step_equals_single_process_step(world):
    """cfg5's mode: tables row-sharded r -> rank r % G, batch split in G parts; afterwards the union of the shards and
    every replicated parameter equal ONE oracle step on the whole batch with whole tables."""
    res = _run(_sharded_step_worker, world)
    cfg, p, X, y = _case()
    acc = orc.init_accumulators(p)
    L, _ = orc.train_step(p, acc, X, y, cfg)
    for rank in range(world):
        loss, got, gacc = res[rank]
        assert abs(loss - L) < 1e-12
        for k, v in got.items():
            if k in ('inner_embeddings', 'outer_embeddings', 'feature_bias'):
                np.testing.assert_allclose(v, p[k][rank::world], rtol=1e-10, atol=1e-12, err_msg='rank %d %s' % (rank, k))
                np.testing.assert_allclose(gacc[k], acc[k][rank::world], rtol=1e-10, atol=1e-14)
            else:
                np.testing.assert_allclose(v, p[k], rtol=1e-10, atol=1e-12, err_msg='rank %d %s' % (rank, k))


def _shard_worker(rank, world):
    from cffm_amd.dist import ShardedTables
    M, dim = 37, 5
    full = torch.arange(M * dim, dtype=torch.float64).reshape(M, dim)
    bias = torch.arange(M, dtype=torch.float64).reshape(M, 1) * 0.5
    local = {'emb': full[rank::world].clone(), 'bias': bias[rank::world].clone()}
    st = ShardedTables(local)
    rng = np.random.default_rng(100 + rank)
    ids = torch.from_numpy(rng.integers(0, M, size=23))
    rows = st.lookup(ids)
    assert torch.equal(rows['emb'], full[ids]) and torch.equal(rows['bias'], bias[ids])
    grads = {'emb': torch.from_numpy(rng.standard_normal((23, dim))), 'bias': torch.from_numpy(rng.standard_normal((23, 1)))}
    local_rows, recv = st.push_grads(grads)
    dense = torch.zeros(M, dim, dtype=torch.float64)
    dense.index_add_(0, ids, grads['emb'])
    owned = torch.zeros_like(local['emb'])
    owned.index_add_(0, local_rows, recv['emb'])
    return dense.numpy(), owned.numpy()


def test_row_sharded_lookup_and_gradient_routing():
    res = _run(_shard_worker, 2)
    total = res[0][0] + res[1][0]                 # what an unsharded table would accumulate over both ranks
    for rank in (0, 1):
        np.testing.assert_allclose(res[rank][1], total[rank::2], rtol=1e-12, atol=1e-12)


def test_route_ids_is_a_stable_grouping():
    from cffm_amd.dist import route_ids, shard_of
    ids = torch.tensor([7, 2, 9, 4, 4, 1, 8])
    perm, counts = route_ids(ids, 3)
    owner = ids[perm] % 3
    assert counts.tolist() == [1, 4, 2] and torch.equal(owner, torch.sort(owner)[0])
    assert ids[perm].tolist() == [9, 7, 4, 4, 1, 2, 8]       # stable inside each owner group
    o, l = shard_of(ids, 3)
    assert torch.equal(o * 1 + l * 3, ids)


# ---- round 3: replicas built from DIFFERENT seeds, content-matched plan prefetch, the CFFM class over a process group --------
def _state_tensors(comp, tables):
    """replicated_state() of the oracle stand-ins: in-place torch views of the numpy parameters and accumulators."""
    out = []
    for d in (comp.p, comp.acc):
        for k in sorted(d):
            if not tables and k in ('inner_embeddings', 'outer_embeddings', 'feature_bias'):
                continue
            if not (isinstance(d[k], np.ndarray) and d[k].flags['C_CONTIGUOUS'] and d[k].flags['WRITEABLE']):
                d[k] = np.array(d[k], dtype=np.float64)
            out.append(torch.from_numpy(d[k].reshape(-1)))
    return out


OracleCompute.replicated_state = lambda self, tables=True: _state_tensors(self, tables)


def _per_rank_seed_worker(rank, world, sharded):
    """Every rank draws its dense parameters from ITS OWN seed (what bench.py did for the device-drawn shards in round 2):
    the step classes must broadcast rank 0's replicated state at construction, or the replicas never agree."""
    from cffm_amd.dist import DataParallelStep, ShardedStep, local_rows_count, replicas_agree, shard_params
    import copy
    cfg, p0, X, y = _case()
    p = init_params(cfg, seed=3 + 10 * rank, dtype=np.float64)           # rank 0: the seed of _case()
    for k in ('feature_bias', 'outer_embeddings', 'inner_embeddings'):   # the tables as _case() prepares them (shared / sharded)
        p[k] = p0[k].copy()
    if sharded:
        lcfg = copy.copy(cfg)
        lcfg.M = local_rows_count(cfg.M, rank, world)
        comp = ShardedOracleCompute(lcfg, shard_params(p, rank, world))
        assert not replicas_agree(comp, tables=False)
        step = ShardedStep(comp)
    else:
        p['inner_embeddings'] = p['inner_embeddings'] * (1.0 + rank)      # replicated tables that start different as well
        comp = OracleCompute(cfg, p)
        assert not replicas_agree(comp, tables=True)
        step = DataParallelStep(comp)
    assert replicas_agree(comp, tables=not sharded)
    per = X.shape[0] // world
    sl = slice(rank * per, rank * per + per)
    X2 = (X[::-1] * 7 + 3) % cfg.M
    l1 = step.train_step(torch.from_numpy(X[sl].copy()), torch.from_numpy(y[sl]))
    l2 = step.train_step(torch.from_numpy(X2[sl].copy()), torch.from_numpy(y[sl]))
    assert replicas_agree(comp, tables=not sharded)
    return (float(l1[0]), float(l2[0])), {k: np.asarray(v).copy() for k, v in comp.p.items()}


@pytest.mark.parametrize('sharded', [False, True])
def test_replicas_built_from_per_rank_seeds_are_synchronised(sharded):
    world = 2
    res = _run(_per_rank_seed_worker, world, sharded)
    cfg, p, X, y = _case()
    X2 = (X[::-1] * 7 + 3) % cfg.M
    acc = orc.init_accumulators(p)
    L1, _ = orc.train_step(p, acc, X, y, cfg)
    L2, _ = orc.train_step(p, acc, X2, y, cfg)
    tables = ('inner_embeddings', 'outer_embeddings', 'feature_bias')
    for rank in range(world):
        (l1, l2), got = res[rank]
        assert abs(l1 - L1) < 1e-12 and abs(l2 - L2) < 1e-10
        for k, v in got.items():
            ref = p[k][rank::world] if (sharded and k in tables) else p[k]
            np.testing.assert_allclose(v, ref, rtol=1e-9, atol=1e-11, err_msg='rank %d %s' % (rank, k))
    for k in res[0][1]:                                   # bit-identical replicas after two steps
        if not (sharded and k in tables):
            np.testing.assert_array_equal(res[0][1][k], res[1][1][k], err_msg=k)


def _plan_token_worker(rank, world):
    """bench.py hands ShardedStep X[i] / X[i + 1]: indexing a tensor builds a NEW Python object every time, so the
    prefetched plan must be matched on content (storage, geometry, version), not on object identity."""
    from cffm_amd.dist import ShardedStep, batch_token, local_rows_count, shard_params
    import copy
    cfg, p, X, y = _case()
    lcfg = copy.copy(cfg)
    lcfg.M = local_rows_count(cfg.M, rank, world)
    step = ShardedStep(ShardedOracleCompute(lcfg, shard_params(p, rank, world)))
    per = X.shape[0] // world
    pool = torch.from_numpy(np.stack([X[rank * per:(rank + 1) * per], ((X[::-1] * 7 + 3) % cfg.M)[rank * per:(rank + 1) * per]]))
    yt = torch.from_numpy(y[rank * per:(rank + 1) * per])
    assert pool[1] is not pool[1] and batch_token(pool[1]) == batch_token(pool[1])
    step.train_step(pool[0], yt, next_ids=pool[1])
    step.train_step(pool[1], yt, next_ids=pool[0])           # fresh view objects of the same rows: the plan must be found
    pool[0][0, 0] = (int(pool[0][0, 0]) + 1) % cfg.M         # an in-place write bumps the version: the stale plan is dropped
    step.train_step(pool[0], yt)
    return step.plans_built, step.plans_reused


def test_prefetched_plan_is_matched_on_content_not_identity():
    res = _run(_plan_token_worker, 2)
    for rank in (0, 1):
        built, reused = res[rank]
        assert reused == 1 and built == 4, (built, reused)    # step 1: 2 plans; step 2: reuse + 1 ahead; step 3: stale -> 1 new


class OracleEngine(OracleCompute):
    """What cffm_amd.CFFM needs from an engine (HipEngine's surface), computed by the float64 oracle on CPU tensors."""
    device = torch.device('cpu')
    opt_step = 0

    def __init__(self, cfg, seed):
        OracleCompute.__init__(self, cfg, init_params(cfg, seed=seed, dtype=np.float64))

    def train_step(self, ids, y):
        L, _ = orc.train_step(self.p, self.acc, ids.numpy(), y.numpy().astype(np.float64), self.cfg)
        return torch.tensor([L])

    def eval_sums(self, ids, y, lo, hi, block=8192):
        if ids.shape[0] == 0:
            return torch.zeros(3, dtype=torch.float64)
        out, _ = orc.forward(self.p, ids.numpy(), self.cfg)
        yt = y.numpy().astype(np.float64)
        pr = np.minimum(np.maximum(out, lo), hi)
        return torch.tensor([np.sum((yt - pr) ** 2), yt.sum(), np.sum(yt * yt)], dtype=torch.float64)

    def export_params(self):
        return {k: np.asarray(v).copy() for k, v in self.p.items()}


def _cffm_class_worker(rank, world, tmp):
    """The drop-in class under a process group: CFFM.train() runs DataParallelStep on every rank's slice of the SAME global
    batches (rank 0's random starts are broadcast), evaluate() splits the rows and all-reduces the metric sums."""
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
    M.CFFM.engine_factory = OracleEngine
    try:
        m = M.CFFM(Mf, 0, os.path.join(tmp, 'r%d_w%d' % (rank, world)), 8, 8, 'square_loss', 2, 8, 0.05, 0, [1.0, 1.0],
                   'AdagradOptimizer', 0, 0, 0, F, 1, 0, 1.0, 1, 1.0, 1, 1.0, 'relu')
        np.random.seed(77)                                   # the reference's batch starts are unseeded: pin them for the test
        m.train(data)
    finally:
        M.CFFM.engine_factory = None
    assert m.world == world and (m._dp is not None) == (world > 1)
    return (m.train_rmse, m.valid_rmse, m.test_rmse, m.train_r2), m.engine.export_params()


def test_cffm_class_trains_data_parallel_under_a_process_group(tmp_path):
    one = _run(_cffm_class_worker, 1, str(tmp_path))[0]
    two = _run(_cffm_class_worker, 2, str(tmp_path))
    for rank in (0, 1):
        for a, b in zip(one[0], two[rank][0]):              # per-epoch metrics: identical on every rank, equal to the single run
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12)
        for k, v in one[1].items():
            np.testing.assert_allclose(two[rank][1][k], v, rtol=1e-8, atol=1e-11, err_msg='rank %d %s' % (rank, k))
    for k in two[0][1]:
        np.testing.assert_array_equal(two[0][1][k], two[1][1][k], err_msg=k)


def test_plan_torch_against_a_plain_numpy_reading():
    """ShardedStep.plan_torch is the specification the device kernel (cffm_shard_plan) is tested against on the GPU: pin it here
    against a loop-level numpy reading of the same definition."""
    from cffm_amd.dist import ShardedStep
    rng = np.random.default_rng(5)
    B, F, M, G = 37, 5, 97, 3
    X = rng.integers(0, 30, size=(B, F)).astype(np.int32) * 3 + 1
    X[X >= M] = M - 1
    st = ShardedStep.__new__(ShardedStep)
    st.world, st.dedup = G, True
    local_ids, order, uniq, pos, send_rows, counts = [t.numpy() for t in st.plan_torch(torch.from_numpy(X))]
    flat = X.reshape(-1).astype(np.int64)
    pairs = [(int(v % G), int(v // G)) for v in flat]
    want_order = sorted(range(len(flat)), key=lambda i: (pairs[i], i))          # stable: slots ascend inside a run
    distinct = sorted(set(pairs))
    index = {p: u for u, p in enumerate(distinct)}
    assert local_ids.reshape(-1).tolist() == [p[1] for p in pairs]
    assert order.tolist() == want_order
    assert uniq.tolist() == [index[pairs[i]] for i in want_order]
    assert pos.tolist() == [index[p] for p in pairs]
    assert send_rows[:len(distinct)].tolist() == [p[1] for p in distinct]
    assert counts.tolist() == [sum(1 for p in distinct if p[0] == o) for o in range(G)]
