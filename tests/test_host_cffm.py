"""Host-side mirror of the reference's CFFM.py surface: flags, constructor, batchers, early stop (CPU), and
an end-to-end train()/evaluate() run through the HIP engine on the committed frappe slices (GPU)."""
import contextlib
import io
import os

import numpy as np
import pytest

from cffm_amd import CFFM as M
from cffm_amd.LoadData import LoadData

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, 'golden', 'frappe_slice') + '/'

REF_DEFAULTS = {   # CFFM.py:24-78
    'path': 'data/', 'dataset': 'frappe', 'epoch': 50, 'pretrain': 0, 'batch_size': 1024, 'inner_dims': 32,
    'outer_dims': 32, 'lamda': 0, 'keep': '[1.0,1.0]', 'lr': 0.05, 'loss_type': 'square_loss',
    'optimizer': 'AdagradOptimizer', 'verbose': 1, 'batch_norm': 0, 'tensorboard': 0, 'num_field': 3,
    'linear_att': 1, 'att_dim': 0, 'lamda_att': 1.0, 'inner_conv': 1, 'gamma_inner': 1.0, 'outer_conv': 1,
    'beta_outer': 1.0, 'activation': 'relu',
}


def make(tmp_path, **kw):
    a = dict(features_M=590, pretrain_flag=0, save_file=str(tmp_path / 'pre' / 'm'), inner_dims=32, outer_dims=32,
             loss_type='square_loss', epoch=2, batch_size=32, learning_rate=0.05, lamda_bilinear=0, keep=[1.0, 1.0],
             optimizer_type='AdagradOptimizer', batch_norm=0, verbose=1, tensorboard=0, num_field=10, linear_att=1,
             att_dim=0, lamda_att=1.0, inner_conv=1, gamma_inner=1.0, outer_conv=1, beta_outer=1.0,
             activation_function='selu')
    a.update(kw)
    return M.CFFM(*[a[k] for k in ('features_M', 'pretrain_flag', 'save_file', 'inner_dims', 'outer_dims', 'loss_type',
                                   'epoch', 'batch_size', 'learning_rate', 'lamda_bilinear', 'keep', 'optimizer_type',
                                   'batch_norm', 'verbose', 'tensorboard', 'num_field', 'linear_att', 'att_dim',
                                   'lamda_att', 'inner_conv', 'gamma_inner', 'outer_conv', 'beta_outer',
                                   'activation_function')])


def test_flag_set_and_defaults_match_reference():
    args = vars(M.parse_args([]))
    assert args == REF_DEFAULTS
    a = M.parse_args('--dataset frappe --epoch 100 --pretrain -1 --batch_size 256 --inner_dims 32 --outer_dims 32 '
                     '--lr 0.05 --num_field 10 --activation selu --lamda 0'.split())      # README.md:28
    assert (a.batch_size, a.num_field, a.activation, a.pretrain) == (256, 10, 'selu', -1)


def test_constructor_binds_like_the_reference(tmp_path):
    m = make(tmp_path)
    assert m.att_dim == 10 and m.num_interactions == 45 and m.random_seed == 2021
    assert os.path.isdir(m.save_file)                          # create_save_folder (CFFM.py:637-639)
    assert m.train_rmse == [] and m.valid_r2 == []
    assert m.calculate_parameters() == 590 * 65 + 45 + 1 + 100 + 10 + 1 + 6 + 5 * (4 * 45 * 45 + 45)
    with pytest.raises(ValueError):
        make(tmp_path, att_dim=7)                              # quirk Q11


def test_batchers_follow_the_reference_rules(tmp_path):
    m = make(tmp_path)
    data = {'X': [[i, i] for i in range(10)] + [[1, 2, 3]] * 3, 'Y': [float(i) for i in range(13)]}
    blk = m.get_ordered_block_from_data(data, 4, 2)
    assert blk['X'] == [[8, 8], [9, 9]] and blk['Y'] == [8.0, 9.0]      # stops where the row length changes
    assert m.get_ordered_block_from_data(data, 4, 4) == {'X': [], 'Y': []}
    np.random.seed(0)
    seen_dup = False
    for _ in range(50):
        b = m.get_random_block_from_data(data, 6)
        assert len(b['X']) <= 6 and all(len(r) == len(b['X'][0]) for r in b['X'])
        assert b['Y'][0] == [float(b['X'][0][0])]                       # labels wrapped as [y]
        starts = [r[0] for r in b['X']]
        if len(starts) != len(set(starts)):
            seen_dup = True                                             # backward fill re-adds the start row (Q9)
    assert seen_dup
    x, y = m.shuffle_in_unison_scary([1, 2, 3, 4, 5], [10, 20, 30, 40, 50])
    assert [v * 10 for v in x] == y
    assert m.shuffle_in_unison_scary([1, 2, 3, 4, 5], [1, 2, 3, 4, 5])[0] == x   # same permutation every call
    assert m.eva_termination([9, 1, 2, 3, 4, 5]) and not m.eva_termination([1, 2, 3, 4, 5])


def test_optimizer_strings(tmp_path):
    for name in ('AdamOptimizer', 'GradientDescentOptimizer', 'MomentumOptimizer'):   # CFFM.py:517-529
        assert make(tmp_path, optimizer_type=name).config.optimizer == name
    with pytest.raises(ValueError):
        make(tmp_path, optimizer_type='RMSPropOptimizer')


class _RecordingEngine(object):
    """The smallest thing CFFM.train() can drive: remembers every batch it is fed (host copies)."""
    import torch as _torch
    device = _torch.device('cpu')
    opt_step = 0

    def __init__(self, cfg, seed):
        self.cfg, self.fed = cfg, []

    def train_step(self, ids, y):
        self.fed.append((ids.numpy().tolist(), y.numpy().tolist()))

    def eval_sums(self, ids, y, lo, hi, block=8192):
        import torch
        yt = y.double()
        return torch.stack([(yt * yt).sum(), yt.sum(), (yt * yt).sum()])      # the constant-0 predictor


def _frappe_slice():
    with contextlib.redirect_stdout(io.StringIO()):
        return LoadData(PATH, 'frappe', 'square_loss')


def _const_rmse(data):
    y = np.asarray(data.Train_data['Y'])
    return float(np.sqrt(np.mean((y - y.mean()) ** 2)))


def test_batch_starts_come_from_the_injected_rng_and_batches_follow_the_reference_batcher(tmp_path):
    """the block starts (CFFM.py:561, `np.random.randint`) sit behind an injectable RNG.  With
    `batch_rng = RandomState(s)` train() draws exactly that generator's stream and leaves the process-global np.random alone, and
    every batch it feeds the engine is the block the reference's own list batcher (CFFM.py:183, :556-581: per-epoch sklearn
    shuffle with random_state 2021, then `get_random_block_from_data`) builds from a twin generator."""
    data = _frappe_slice()
    X, Y = list(data.Train_data['X']), list(data.Train_data['Y'])
    n, bs, epochs = len(Y), 16, 3
    M.CFFM.engine_factory = _RecordingEngine
    try:
        m = make(tmp_path, epoch=epochs, batch_size=bs, verbose=0)
        m.batch_rng = np.random.RandomState(5)
        np.random.seed(123)
        before = np.random.get_state()[1].copy()
        m.train(data)
        assert np.array_equal(np.random.get_state()[1], before)           # the global generator was not consumed
    finally:
        M.CFFM.engine_factory = None
    twin = np.random.RandomState(5)
    want = [twin.randint(0, n - bs, size=n // bs) for _ in range(epochs)]
    assert len(m.batch_starts) == epochs and all(np.array_equal(a, b) for a, b in zip(m.batch_starts, want))
    # the reference's loop, list semantics, on a generator in the same state (one randint per step is the same stream)
    ref = make(tmp_path, batch_size=bs)
    ref.batch_rng = np.random.RandomState(5)
    split, fed = {'X': X, 'Y': Y}, iter(m.engine.fed)
    for _ in range(epochs):
        split['X'], split['Y'] = ref.shuffle_in_unison_scary(split['X'], split['Y'])
        for _ in range(n // bs):
            blk = ref.get_random_block_from_data(split, bs)
            got_x, got_y = next(fed)
            assert got_x == blk['X'] and got_y == [v[0] for v in blk['Y']]
    assert next(fed, None) is None
    # the constructor takes the generator too (keyword after random_seed: the positional signature is the reference's)
    m2 = M.CFFM(590, 0, str(tmp_path / 'p2'), 32, 32, 'square_loss', 1, 16, 0.05, 0, [1.0, 1.0], 'AdagradOptimizer', 0, 0, 0, 10, 1,
                0, 1.0, 1, 1.0, 1, 1.0, 'selu', batch_rng=np.random.RandomState(9))
    assert m2.batch_rng.randint(0, 100) == np.random.RandomState(9).randint(0, 100) and make(tmp_path).batch_rng is np.random


TRAIN_SEED, TRAIN_EPOCHS = 3, 24       # a pinned generator for the block starts; the property below is checked on BOTH engines


def _check_seeded_run(m, data, epochs):
    """What a seeded train() on the 120-row frappe slice must show.  Adagrad from accumulators of 1e-8 moves every parameter by
    ~lr at its first touch, so the first epochs swing between the clip bounds (clipped RMSE 1.1 <-> 1.66 on the float64
    oracle as on the device) and two implementations drift apart after a few steps (tests/test_gpu_parity.py, trajectory
    test): the criterion is therefore a property of the pinned run, not a trajectory match - the run ends at `epochs` or by
    the reference's early-stop rule (CFFM.py:631-635), and by then the clipped train RMSE has been below the constant
    predictor's."""
    twin = np.random.RandomState(TRAIN_SEED)
    n, bs = len(data.Train_data['Y']), m.batch_size
    for got in m.batch_starts:
        assert np.array_equal(got, twin.randint(0, n - bs, size=n // bs))
    ran = len(m.valid_rmse)
    assert len(m.train_rmse) == len(m.test_rmse) == len(m.batch_starts) == ran >= 1
    assert ran == epochs or (m.eva_termination(m.valid_rmse) and not any(m.eva_termination(m.valid_rmse[:k]) for k in range(ran)))
    assert all(np.isfinite(v) for v in m.train_rmse + m.valid_rmse + m.test_rmse + m.valid_r2)
    assert min(m.train_rmse) < _const_rmse(data), (m.train_rmse, _const_rmse(data))


def test_seeded_train_on_the_oracle_engine(tmp_path):
    """The oracle-side twin of the GPU end-to-end test: the same seeded train() through the float64 oracle."""
    from tests.test_dist_cpu import OracleEngine
    data = _frappe_slice()
    M.CFFM.engine_factory = OracleEngine
    try:
        m = make(tmp_path, epoch=TRAIN_EPOCHS, batch_size=16, verbose=0)
        m.batch_rng = np.random.RandomState(TRAIN_SEED)
        m.train(data)
    finally:
        M.CFFM.engine_factory = None
    _check_seeded_run(m, data, TRAIN_EPOCHS)


@pytest.mark.gpu
def test_train_and_evaluate_end_to_end(tmp_path, caplog):
    """train() on the frappe slice through the HIP engine, block starts from a pinned generator (the kernels are
    bit-reproducible, so the run is deterministic): same property as the oracle twin above, plus the log lines."""
    import logging
    data = _frappe_slice()
    m = make(tmp_path, epoch=TRAIN_EPOCHS, batch_size=16)
    m.batch_rng = np.random.RandomState(TRAIN_SEED)
    with caplog.at_level(logging.INFO):
        m.train(data)
    print('train rmse: ' + ' '.join('%.3f' % v for v in m.train_rmse))
    print('valid rmse: ' + ' '.join('%.3f' % v for v in m.valid_rmse))
    _check_seeded_run(m, data, TRAIN_EPOCHS)
    text = caplog.text
    assert '#params: %d' % m.calculate_parameters() in text and 'Init_RMSE: train=' in text and 'Epoch 1 [' in text
    rmse, r2 = m.evaluate(data.Validation_data)
    assert rmse == pytest.approx(m.valid_rmse[-1], rel=1e-6)


@pytest.mark.gpu
def test_seeded_train_is_reproducible(tmp_path):
    """Two runs from the same generator state end with identical metric lists (fixed reduction orders on the device)."""
    runs = []
    for i in range(2):
        data = _frappe_slice()
        m = make(tmp_path, epoch=3, batch_size=16, verbose=0)
        m.batch_rng = np.random.RandomState(TRAIN_SEED)
        m.train(data)
        runs.append((m.train_rmse, m.valid_rmse, m.test_rmse, m.train_r2))
    assert runs[0] == runs[1]


@pytest.mark.gpu
def test_checkpoint_round_trip_of_the_pretrain_flags(tmp_path):
    """--pretrain -1 writes the model after every epoch, --pretrain 1 restores THIS model's tensors and optimizer slots (the
    reference's restore is broken, quirk Q7): a fresh model restored from the file predicts the same and resumes the same."""
    import torch
    data = _frappe_slice()
    m = make(tmp_path, epoch=2, batch_size=16, pretrain_flag=-1, verbose=0)
    m.batch_rng = np.random.RandomState(1)
    m.train(data)
    assert os.path.exists(m.save_file + '.pt')                # written every epoch by --pretrain -1
    rmse, r2 = m.evaluate(data.Validation_data)
    m2 = make(tmp_path, pretrain_flag=1)
    m2.build_graph()
    r2mse, r2r2 = m2.evaluate(data.Validation_data)
    assert (r2mse, r2r2) == (rmse, r2)                        # same tensors, same kernels: bit-identical
    for k, v in m.engine.export_accumulators().items():
        np.testing.assert_array_equal(m2.engine.export_accumulators()[k], v, err_msg=k)
    ids, y, _ = m._device_split(data.Train_data)
    for eng in (m.engine, m2.engine):                         # one more step from both: the optimizer state came along
        eng.train_step(ids[:16], y[:16])
    torch.cuda.synchronize()
    for k, v in m.engine.export_params().items():
        np.testing.assert_array_equal(m2.engine.export_params()[k], v, err_msg=k)


@pytest.mark.gpu
def test_evaluate_on_device_matches_oracle_and_sklearn(tmp_path):
    """evaluate() (CFFM.py:583-615) keeps predictions, clip and metric sums on the device.  Against (a) the oracle's
    forward + clipped RMSE/R2 on the same split and parameters, (b) sklearn's mean_squared_error / r2_score on the host
    copy of the device's own predictions (1e-12: only the float64 summation order differs), with a ragged last block."""
    import math
    from sklearn.metrics import mean_squared_error, r2_score
    from oracle import cffm_oracle as orc
    with contextlib.redirect_stdout(io.StringIO()):
        data = LoadData(PATH, 'frappe', 'square_loss')
    m = make(tmp_path, batch_size=7)
    m.build_graph()
    eng = m.engine
    eng.fbias.normal_(0.0, 0.4)                                  # non-trivial outputs, some beyond the clip range
    eng.outer.mul_(25.0)
    for split in (data.Validation_data, data.Test_data):
        rmse, r2 = m.evaluate(split)
        X, Y = LoadData.packed(split)
        y_true = Y.astype(np.float64)
        pred = m.predict_split(split)
        assert (pred < y_true.min()).any() or (pred > y_true.max()).any()      # the clip is exercised
        bounded = np.minimum(np.maximum(pred, y_true.min()), y_true.max())
        assert rmse == pytest.approx(math.sqrt(mean_squared_error(y_true, bounded)), rel=1e-12)
        assert r2 == pytest.approx(r2_score(y_true, bounded), rel=1e-12, abs=1e-12)
        p64 = {k: np.asarray(v, dtype=np.float64) for k, v in eng.export_params().items()}
        out_ref, _ = orc.forward(p64, X, m.config)
        o_rmse, o_r2 = orc.clipped_rmse_r2(out_ref, y_true)
        assert rmse == pytest.approx(o_rmse, rel=1e-5) and r2 == pytest.approx(o_r2, rel=1e-4, abs=1e-5)


@pytest.mark.gpu
def test_train_leaves_the_callers_lists_shuffled_like_the_reference(tmp_path):
    """CFFM.py:183 re-binds data.Train_data['X'] / ['Y'] to sklearn-shuffled lists (random_state 2021) every epoch."""
    from sklearn.utils import shuffle
    with contextlib.redirect_stdout(io.StringIO()):
        data = LoadData(PATH, 'frappe', 'square_loss')
    X0, Y0 = list(data.Train_data['X']), list(data.Train_data['Y'])
    m = make(tmp_path, epoch=3, batch_size=16, verbose=0)
    m.train(data)
    Xr, Yr = X0, Y0
    for _ in range(3):
        Xr, Yr = shuffle(Xr, Yr, random_state=2021)
    assert data.Train_data['X'] == Xr and data.Train_data['Y'] == Yr
    # and the device copy follows the same order: evaluate() on the re-bound lists does not re-pack
    ids, y, _ = m._device_split(data.Train_data)
    assert ids.cpu().numpy().tolist() == Xr


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment launches N ranks as CHILD processes through
    torch.distributed.run on 127.0.0.1 and hands their status back; nothing in that route may touch the GPU."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '5', '--warmup', '2'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    import torch
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: (_ for _ in ()).throw(AssertionError('GPU touched before the launch')))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-6:] == ['--gpus', '4', '--steps', '5', '--warmup', '2']
    assert seen['env'].get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'


def test_workspace_pool_frees_an_outgrown_buffer_with_its_last_graph():
    """Two captured batch shapes (two graph keys) and one regrowth: the outgrown buffer lives exactly as long as a graph of ITS
    generation is pinned, also when a graph of the other shape is never called again (cffm_amd.engine.WorkspacePool,
    DataParallelStep._drop_stale_graphs)."""
    import torch
    from cffm_amd.dist import DataParallelStep
    from cffm_amd.engine import WorkspacePool

    class Compute(object):                         # the slice of HipEngine the graph bookkeeping talks to
        def __init__(self):
            self.pool = WorkspacePool(lambda n: torch.empty(n, dtype=torch.uint8))
        ws_generation = property(lambda self: self.pool.generation)
        pin_workspace = lambda self: self.pool.pin()
        unpin_workspace = lambda self, gen: self.pool.unpin(gen)

    c = Compute()
    dp = DataParallelStep.__new__(DataParallelStep)
    dp.c, dp._graphs = c, {}
    first = c.pool.get(1000)
    for key in ('shape-a', 'shape-b'):             # two captures against generation 0
        dp._graphs[key] = {'graph': object(), 'ws_gen': c.ws_generation, 'pinned': True}
        c.pin_workspace()
    assert c.pool.pins == {0: 2} and c.pool.get(500) is first                     # smaller request: same buffer, same generation
    second = c.pool.get(4000)                                                     # regrowth under two pins: generation 0 retired
    assert c.ws_generation == 1 and c.pool.retired[0] is first and c.pool.retired_bytes() == 1000
    assert dp._drop_stale_graphs() == 2                                           # BOTH stale graphs go, whichever shape is called
    assert all(st['graph'] is None for st in dp._graphs.values())
    assert c.pool.pins == {} and c.pool.retired == {} and c.pool.buf is second    # the outgrown buffer is freed
    # one shape re-captures against generation 1, the buffer grows again, the other shape is never called: no leak either
    dp._graphs['shape-a'].update(graph=object(), ws_gen=c.ws_generation, pinned=True)
    c.pin_workspace()
    c.pool.get(9000)
    assert list(c.pool.retired) == [1] and dp._drop_stale_graphs() == 1 and c.pool.retired == {} and dp._drop_stale_graphs() == 0
    # an unpinned generation is not retired at all
    c.pool.get(20000)
    assert c.pool.retired == {} and c.ws_generation == 3


def test_bench_leaves_with_one_line_when_fewer_gpus_than_ranks(monkeypatch, capsys):
    """`bench.py --gpus N` on a node with fewer than N visible devices: every rank exits non-zero before any collective, rank 0
    prints one line (the count is read in the launcher's children, never in the parent)."""
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    import torch
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 1)
    for rank in (0, 1):
        monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2'])
        for k, v in (('WORLD_SIZE', '2'), ('RANK', str(rank)), ('LOCAL_RANK', str(rank))):
            monkeypatch.setenv(k, v)
        monkeypatch.delenv('CFFM_BENCH_REHEARSAL', raising=False)
        with pytest.raises(SystemExit) as e:
            bench.main()
        assert e.value.code == 3
        err = capsys.readouterr().err
        assert (err == 'bench.py: --gpus 2 needs 2 visible GPUs, this node shows 1\n') if rank == 0 else (err == '')
