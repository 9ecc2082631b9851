"""Generates tests/golden/loader_golden.json by running the REFERENCE loader
(/root/reference/LoadData.py) in this container on small slices of the two frappe files that are
present in the reference tree.  Run once from the repo root:

    python tests/golden/make_loader_golden.py

The slices (data, not source) are committed under tests/golden/frappe_slice/ so that the test can run
on the GPU box, where /root/reference does not exist.  The train split of frappe is missing from the
reference tree (.MISSING_LARGE_BLOBS:7); rows 0-119 of the validation file stand in for it.
"""
import contextlib
import importlib.util
import io
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
SRC = os.path.join(REF, 'data', 'frappe')
DST = os.path.join(HERE, 'frappe_slice', 'frappe')


def take(src, lo, hi):
    with open(src) as fh:
        return fh.readlines()[lo:hi]


def main():
    os.makedirs(DST, exist_ok=True)
    val = os.path.join(SRC, 'frappe.validation.libfm')
    tst = os.path.join(SRC, 'frappe.test.libfm')
    slices = {'frappe.train.libfm': take(val, 0, 120),
              'frappe.validation.libfm': take(val, 120, 200),
              'frappe.test.libfm': take(tst, 0, 80)}
    for name, lines in slices.items():
        with open(os.path.join(DST, name), 'w') as fh:
            fh.writelines(lines)

    spec = importlib.util.spec_from_file_location('ref_LoadData', os.path.join(REF, 'LoadData.py'))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    golden = {}
    for loss_type in ('square_loss', 'log_loss'):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            d = ref.LoadData(os.path.join(HERE, 'frappe_slice') + '/', 'frappe', loss_type)
        golden[loss_type] = {
            'features_M': d.features_M,
            'features': d.features,
            'stdout': buf.getvalue(),
            'splits': {name: sorted([list(x) + [y] for x, y in zip(s['X'], s['Y'])])
                       for name, s in (('train', d.Train_data), ('validation', d.Validation_data),
                                       ('test', d.Test_data))},
        }
    with open(os.path.join(HERE, 'loader_golden.json'), 'w') as fh:
        json.dump(golden, fh, sort_keys=True)
    print('features_M', golden['square_loss']['features_M'])


if __name__ == '__main__':
    main()
