"""Synthetic libfm data of a given field/dimension shape.

Field f owns the disjoint raw-id range [f*ceil(M/F), min((f+1)*ceil(M/F), M)) - frappe's fields are
disjoint contiguous ranges too - one id per field per row, value 1, label +1 with probability 1/3 else -1
(README.md:47).  ``batches`` produces the post-loader representation directly (int32 [B,F], fp32 [B]);
``write_libfm`` writes text files at plumbing scale that go through ``LoadData``.
"""
import os

import numpy as np


def field_ranges(M, F):
    w = -(-M // F)
    return [(f * w, min((f + 1) * w, M)) for f in range(F)]


def sample_ids(rng, M, F, n, dist='uniform', zipf_s=1.05):
    X = np.empty((n, F), dtype=np.int32)
    for f, (lo, hi) in enumerate(field_ranges(M, F)):
        size = max(hi - lo, 1)
        if dist == 'uniform':
            X[:, f] = lo + rng.integers(0, size, size=n)
        elif dist == 'zipf':
            w = 1.0 / np.arange(1, size + 1) ** zipf_s
            X[:, f] = lo + rng.choice(size, size=n, p=w / w.sum())
        else:
            raise ValueError(dist)
    return X


def sample_labels(rng, n):
    return np.where(rng.random(n) < 1.0 / 3.0, 1.0, -1.0).astype(np.float32)


def batches(M, F, B, n_batches, seed=2021, dist='uniform'):
    rng = np.random.default_rng(seed)
    X = sample_ids(rng, M, F, B * n_batches, dist).reshape(n_batches, B, F)
    y = sample_labels(rng, B * n_batches).reshape(n_batches, B)
    return X, y


def write_libfm(path, dataset, M, F, n_train, n_valid, n_test, seed=2021, dist='uniform'):
    rng = np.random.default_rng(seed)
    d = os.path.join(path, dataset)
    os.makedirs(d, exist_ok=True)
    for split, n in (('train', n_train), ('validation', n_valid), ('test', n_test)):
        X = sample_ids(rng, M, F, n, dist)
        y = sample_labels(rng, n)
        with open(os.path.join(d, '%s.%s.libfm' % (dataset, split)), 'w') as fh:
            for row, lab in zip(X, y):
                fh.write('%d %s\n' % (int(lab), ' '.join('%d:1' % v for v in row)))
    return d
