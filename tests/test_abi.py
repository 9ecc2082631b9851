"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/cffm_hip.h
declares, and its layout queries agree with the Python-side shapes.  No compute call is made."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from cffm_amd import hip
from cffm_amd.spec import CFFMConfig, param_shapes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return hip.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    header = open(os.path.join(ROOT, 'include', 'cffm_hip.h')).read()
    declared = set(re.findall(r'^\s*(?:int|int32_t|int64_t|const char \*)\s*\*?\s*(cffm_\w+)\s*\(', header, flags=re.M))
    assert len(declared) >= 20
    assert declared == set(hip.PROTOTYPES), declared ^ set(hip.PROTOTYPES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.cffm_abi_version() == hip.ABI_VERSION == 9
    assert b'bad shape' in lib.cffm_error_string(10001)
    # the pybind11 layer (north_star's binding) exposes the same entry points and is what the engine calls through
    fast = hip.fast()
    assert hip.binding_name() == 'pybind11', 'cffm_amd/lib/_cffm_pybind*.so is not built (make)'
    for name in declared:
        assert callable(getattr(fast, name)), name
    assert fast.cffm_abi_version() == 9 and 'bad shape' in fast.cffm_error_string(10001)
    sh = hip.make_shape(CFFMConfig(M=10, F=3, K=8, D=8))
    assert fast.cffm_packed_row_floats(C.addressof(sh)) == lib.cffm_packed_row_floats(C.byref(sh)) == 20
    tl_a, tl_b = hip.ThetaLayout(), hip.ThetaLayout()
    assert fast.cffm_theta_layout(C.addressof(sh), C.addressof(tl_a)) == 0 == lib.cffm_theta_layout(C.byref(sh), C.byref(tl_b))
    assert bytes(tl_a) == bytes(tl_b)


def test_theta_layout_matches_reference_variable_sizes(lib):
    cfg = CFFMConfig(M=5382, F=10, K=32, D=32, activation='selu')
    tl = hip.theta_layout(hip.make_shape(cfg))
    shapes = param_shapes(cfg)
    assert (tl.P, tl.Pp, tl.Lc, tl.live) == (45, 48, 5, 4)
    members = [('att_W', 'bias_W'), ('att_b', 'bias_b'), ('inner_cw', 'inner_layer_conv_weight_0'),
               ('inner_dw', 'dense_kernel'), ('d1_w', 'dense_1_kernel'), ('d2_w', 'dense_2_kernel'),
               ('lin_w', 'dense_3_kernel')]
    spans = [(getattr(tl, m), int(np.prod(shapes[n]))) for m, n in members]
    # conv weights / biases are stored channel-padded: [4][Pp][Pp] and [Pp]
    spans += [(tl.conv_w[l], 4 * 48 * 48) for l in range(tl.live)] + [(tl.conv_b[l], 48) for l in range(tl.live)]
    spans += [(tl.bias, 1), (tl.inner_cb, 2), (tl.inner_db, 1), (tl.d1_b, 32), (tl.d2_b, 1), (tl.lin_b, 1)]
    spans.sort()
    for (o0, n0), (o1, _) in zip(spans, spans[1:]):
        assert o0 + n0 <= o1 and o0 % 4 == 0             # disjoint, 16-byte aligned members
    assert spans[-1][0] + spans[-1][1] <= tl.n
    # trained dense parameters = everything but the tables, outer_W/outer_b and the dead last conv layer
    trained = sum(int(np.prod(s)) if s else 1 for k, s in shapes.items()
                  if k not in ('inner_embeddings', 'outer_embeddings', 'feature_bias', 'outer_W', 'outer_b')
                  and not k.endswith('_%d' % (cfg.Lc - 1)))
    pad = tl.live * (4 * (48 * 48 - 45 * 45) + 3)
    assert trained + pad <= tl.n < trained + pad + 4 * len(spans)


def test_workspace_layout(lib):
    cfg = CFFMConfig(M=1000, F=6, K=32, D=32)
    sh = hip.make_shape(cfg)
    w1, w2 = hip.ws_layout(sh, 1), hip.ws_layout(sh, 512)
    assert w1.gpart == w2.gpart == 0                     # cffm_reduce_slabs relies on a B-independent offset
    assert w2.bytes > w1.bytes
    offs = [w2.Ei, w2.Eo, w2.fb, w2.inner_out, w2.t1, w2.h1, w2.att, w2.out, w2.dout, w2.dt1, w2.dEi, w2.dEo, w2.dfb]
    offs += [w2.C[l] for l in range(4)] + [w2.dC[l] for l in range(4)]
    assert len(set(offs)) == len(offs) and all(o % 256 == 0 and 0 < o < w2.bytes for o in offs)
    assert w2.C[1] - w2.C[0] >= 512 * 16 * 16 * 16 * 4   # [B,16,16,Pp=16] fp32


def test_bad_shapes_are_rejected(lib):
    tl = hip.ThetaLayout()
    for kw in (dict(F=1), dict(D=24), dict(D=2), dict(K=6), dict(F=65), dict(loss=6), dict(optimizer=4), dict(act=9)):
        base = dict(M=10, F=3, K=8, D=8, act=0, linear_att=1, inner_conv=1, outer_conv=1, loss=0,
                    lamda_att=1.0, beta_outer=1.0, lr=0.05)
        base.update(kw)
        assert lib.cffm_theta_layout(C.byref(hip.Shape(**base)), C.byref(tl)) == 10001
    with pytest.raises(ValueError):
        hip.make_shape(CFFMConfig(M=10, F=3, loss_type='hinge'))
    assert hip.make_shape(CFFMConfig(M=10, F=3, loss_type='hybrid')).loss == 5


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from cffm_amd.engine import HipEngine
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        HipEngine(CFFMConfig(M=10, F=3, K=8, D=8))


def test_div_chunks_guard_boundary():
    """The chunk -> (slot, piece) division of the packed gather / staging kernels (gather.hip: div_chunks), replayed in
    numpy float64 at the host guard's boundary (total < 2^31): exact with the DOUBLE reciprocal the kernels now use;
    the float reciprocal widened to double that they used before is off by more than the +-1 correction can repair."""
    rng = np.random.default_rng(7)
    for CH in (5, 9, 13, 17, 21, 25, 33, 129):
        top = (1 << 31) - 1
        g = np.unique(np.concatenate([np.arange(0, 4 * CH), top - np.arange(0, 4 * CH),
                                      (top // CH - np.arange(0, 64)) * CH, (top // CH - np.arange(0, 64)) * CH - 1,
                                      rng.integers(0, top, size=200000)]))
        g = g[(g >= 0) & (g <= top)].astype(np.int64)
        inv = np.float64(1.0) / np.float64(CH)
        q = ((g.astype(np.float64) + 0.5) * inv).astype(np.int64)
        r = g - q * CH
        q = np.where(r < 0, q - 1, np.where(r >= CH, q + 1, q))
        r = g - q * CH
        assert np.array_equal(q, g // CH) and np.array_equal(r, g % CH), CH
    # what round 2 shipped: 1.f / CH widened to double - wrong beyond the single correction near 2^30 for CH = 13
    CH = 13
    g = np.arange((1 << 30) - 5000, (1 << 30), dtype=np.int64)
    inv32 = np.float64(np.float32(1.0) / np.float32(CH))
    q = ((g.astype(np.float64) + 0.5) * inv32).astype(np.int64)
    r = g - q * CH
    assert np.any((r < -CH) | (r >= 2 * CH))
