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


@pytest.mark.gpu
def test_train_and_evaluate_end_to_end(tmp_path, caplog):
    """train() on the frappe slice must move clipped RMSE below the constant predictor's."""
    import logging
    with contextlib.redirect_stdout(io.StringIO()):
        data = LoadData(PATH, 'frappe', 'square_loss')
    m = make(tmp_path, epoch=40, batch_size=16, pretrain_flag=-1)   # the oracle needs ~10 epochs on this slice too
    with caplog.at_level(logging.INFO):
        m.train(data)
    assert len(m.train_rmse) == len(m.valid_rmse) == len(m.test_rmse) >= 1
    y = np.asarray(data.Train_data['Y'])
    const_rmse = float(np.sqrt(np.mean((y - y.mean()) ** 2)))
    assert min(m.train_rmse) < const_rmse
    assert all(np.isfinite(v) for v in m.train_rmse + m.valid_rmse + m.test_rmse + m.valid_r2)
    text = caplog.text
    assert '#params: %d' % m.calculate_parameters() in text and 'Init_RMSE: train=' in text and 'Epoch 1 [' in text
    rmse, r2 = m.evaluate(data.Validation_data)
    assert rmse == pytest.approx(m.valid_rmse[-1], rel=1e-6)
    # checkpoint round trip (--pretrain -1 / 1): a fresh model restored from the file predicts the same
    assert os.path.exists(m.save_file + '.pt')                # written every epoch by --pretrain -1
    m.save(m.save_file)
    m2 = make(tmp_path, pretrain_flag=1)
    m2.build_graph()
    r2mse, _ = m2.evaluate(data.Validation_data)
    assert r2mse == pytest.approx(rmse, rel=1e-6)


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
