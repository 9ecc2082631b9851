"""cffm_amd.LoadData against golden output of the reference loader (LoadData.py:25-112), captured by
tests/golden/make_loader_golden.py on committed slices of the reference's frappe files."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from cffm_amd.LoadData import LoadData

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, 'golden', 'loader_golden.json')))
PATH = os.path.join(HERE, 'golden', 'frappe_slice') + '/'


@pytest.mark.parametrize('loss_type', ['square_loss', 'log_loss'])
def test_loader_matches_reference_golden(loss_type):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        d = LoadData(PATH, 'frappe', loss_type)
    g = GOLD[loss_type]
    assert d.features_M == g['features_M'] == 590
    assert d.features == g['features']                     # identical token -> id map
    assert buf.getvalue() == g['stdout']                   # same console prints
    for name, s in (('train', d.Train_data), ('validation', d.Validation_data), ('test', d.Test_data)):
        got = sorted([list(x) + [y] for x, y in zip(s['X'], s['Y'])])
        assert got == g['splits'][name]                    # multiset parity (argsort is unstable)
        assert isinstance(s['X'], list) and isinstance(s['Y'], list) and isinstance(s['Y'][0], float)


def test_ids_follow_train_test_validation_order():
    with contextlib.redirect_stdout(io.StringIO()):
        d = LoadData(PATH, 'frappe', 'square_loss')
    first_train_tok = open(PATH + 'frappe/frappe.train.libfm').readline().split()[1]
    assert d.features[first_train_tok] == 0
    # a token that first appears in the test file gets a smaller id than one first seen in validation
    train_toks = set(t for l in open(PATH + 'frappe/frappe.train.libfm') for t in l.split()[1:])
    test_new = [t for l in open(PATH + 'frappe/frappe.test.libfm') for t in l.split()[1:] if t not in train_toks]
    test_set = set(test_new)
    val_new = [t for l in open(PATH + 'frappe/frappe.validation.libfm') for t in l.split()[1:]
               if t not in train_toks and t not in test_set]
    assert test_new and val_new
    assert max(d.features[t] for t in test_new) < min(d.features[t] for t in val_new)


def test_packed_arrays():
    with contextlib.redirect_stdout(io.StringIO()):
        d = LoadData(PATH, 'frappe', 'square_loss')
    X, Y = LoadData.packed(d.Train_data)
    assert X.dtype == np.int32 and X.shape == (120, 10) and Y.dtype == np.float32 and Y.shape == (120,)
    assert set(np.unique(Y)) <= {-1.0, 1.0}


def _both(path, dataset, loss_type='square_loss'):
    out = []
    for native in (False, True):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            d = LoadData(path, dataset, loss_type, native=native)
        out.append((d, buf.getvalue()))
    return out


def test_native_reader_equals_python_reader_on_golden_slices():
    (py, out_py), (nat, out_nat) = _both(PATH, 'frappe')
    assert out_py == out_nat == GOLD['square_loss']['stdout']
    assert nat.features_M == py.features_M and nat.features == py.features == GOLD['square_loss']['features']
    for a, b in ((py.Train_data, nat.Train_data), (py.Validation_data, nat.Validation_data), (py.Test_data, nat.Test_data)):
        assert a['X'] == b['X'] and a['Y'] == b['Y']                 # same rows in the same (argsort) order
        assert isinstance(b['X'][0], list) and isinstance(b['Y'][0], float)


def test_native_reader_on_awkward_text(tmp_path):
    """Ragged rows, repeated tokens, a token that differs only in its value part, CRLF and trailing spaces,
    scientific-notation labels, a last line without newline."""
    d = tmp_path / 'odd'
    d.mkdir()
    (d / 'odd.train.libfm').write_text('1 3:1 7:1 3:2\n-1 7:1\r\n0.5 9:1 3:1 11:0.25 \n2e-1 3:1')
    (d / 'odd.test.libfm').write_text('-1 100:1 3:1\n')
    (d / 'odd.validation.libfm').write_text('1 7:1 200:1 100:1\n')
    (py, _), (nat, _) = _both(str(tmp_path) + '/', 'odd')
    assert py.features == nat.features == {'3:1': 0, '7:1': 1, '3:2': 2, '9:1': 3, '11:0.25': 4, '100:1': 5, '200:1': 6}
    assert py.features_M == nat.features_M == 7
    for a, b in ((py.Train_data, nat.Train_data), (py.Validation_data, nat.Validation_data), (py.Test_data, nat.Test_data)):
        assert sorted(zip(map(tuple, a['X']), a['Y'])) == sorted(zip(map(tuple, b['X']), b['Y']))
    assert sorted(nat.Train_data['Y']) == [-1.0, 0.2, 0.5, 1.0]
    assert [len(r) for r in nat.Train_data['X']] == sorted(len(r) for r in nat.Train_data['X'])   # sorted by length


def test_native_reader_through_synthetic_generator(tmp_path):
    from cffm_amd import synth
    synth.write_libfm(str(tmp_path), 'syn', M=400, F=6, n_train=300, n_valid=80, n_test=50, seed=7)
    (py, _), (nat, _) = _both(str(tmp_path) + '/', 'syn')
    assert py.features == nat.features and py.features_M == nat.features_M <= 400
    assert py.Train_data['X'] == nat.Train_data['X'] and py.Train_data['Y'] == nat.Train_data['Y']
    X, Y = LoadData.packed(nat.Train_data)
    assert X.shape == (300, 6) and X.max() < nat.features_M


def test_binary_cache_round_trip_and_invalidation(tmp_path):
    """N2: the parsed splits + token arena are cached next to the data, keyed by size and mtime of the three files."""
    from cffm_amd import synth
    synth.write_libfm(str(tmp_path), 'syn', M=300, F=5, n_train=200, n_valid=60, n_test=40, seed=3)
    path = str(tmp_path) + '/'

    def load(**kw):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            d = LoadData(path, 'syn', 'square_loss', native=True, **kw)
        return d, buf.getvalue()
    a, out_a = load()
    assert not a.cache_hit and os.path.exists(path + 'syn/.syn.cffm_cache.npz')
    b, out_b = load()
    assert b.cache_hit and out_a == out_b                                   # same console prints from the cache
    assert b.features_M == a.features_M and b.features == a.features
    for x, z in ((a.Train_data, b.Train_data), (a.Validation_data, b.Validation_data), (a.Test_data, b.Test_data)):
        assert x['X'] == z['X'] and x['Y'] == z['Y']
    c, _ = load(cache=False)
    assert not c.cache_hit and c.Train_data['X'] == a.Train_data['X']
    # a changed file invalidates the cache (size/mtime key), and the new content is what comes back
    with open(path + 'syn/syn.test.libfm', 'a') as fh:
        fh.write('1 999999:1 1:1 2:1 3:1 4:1\n')
    d, _ = load()
    assert not d.cache_hit and len(d.Test_data['Y']) == len(a.Test_data['Y']) + 1 and '999999:1' in d.features
    # log_loss labels come out of the same cache
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        e = LoadData(path, 'syn', 'log_loss', native=True)
    assert e.cache_hit and set(e.Train_data['Y']) <= {0.0, 1.0}


def test_lazy_split_behaves_like_the_reference_dict():
    with contextlib.redirect_stdout(io.StringIO()):
        d = LoadData(PATH, 'frappe', 'square_loss', native=True)
    s = d.Train_data
    assert isinstance(s, dict) and 'X' in s and 'Y' in s and len(s) == 2 and sorted(s.keys()) == ['X', 'Y']
    X, Y = LoadData.packed(s)                                 # no lists built yet
    assert not dict.__contains__(s, 'X')
    assert s['X'] == X.tolist() and s['Y'] == Y.astype(np.float64).tolist()
    s['X'] = [r[::-1] for r in s['X']]                        # re-binding (CFFM.py:183 does it every epoch)
    X2, _ = LoadData.packed(s)
    assert X2.tolist() == [r[::-1] for r in X.tolist()]
    assert d.truncate_features() == 10
