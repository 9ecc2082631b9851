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
