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
