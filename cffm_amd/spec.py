"""Shapes, hyper-parameters and parameter initialisation of the CFFM hot path.

Mirrors the constructor arguments of the reference class (CFFM.py:98-147) and the variable set it
creates (CFFM.py:239-293, :323, :375-377, :339, :409-410, :441).  Pure Python/numpy: importable
without a GPU and without the HIP library.
"""
import math
from dataclasses import dataclass

import numpy as np

ACTIVATIONS = ('relu', 'prelu', 'elu', 'selu', 'gelu')   # CFFM.py:132-141, enum order of cffm_hip.h
ADAGRAD_INIT_ACC = 1e-8                                  # CFFM.py:524


@dataclass
class CFFMConfig:
    M: int                      # features_M
    F: int                      # num_field
    K: int = 32                 # inner_dims
    D: int = 32                 # outer_dims
    activation: str = 'relu'
    lamda_att: float = 1.0
    beta_outer: float = 1.0
    linear_att: int = 1
    inner_conv: int = 1
    outer_conv: int = 1
    loss_type: str = 'square_loss'
    lamda_bilinear: float = 0.0
    optimizer: str = 'AdagradOptimizer'
    lr: float = 0.05

    def __post_init__(self):
        if self.activation not in ACTIVATIONS:
            raise ValueError('activation must be one of %s' % (ACTIVATIONS,))
        if self.F < 2:
            raise ValueError('num_field must be >= 2')
        if self.D < 4 or (self.D & (self.D - 1)):
            raise ValueError('outer_dims must be a power of two >= 4 (stride-2 VALID conv chain)')
        if self.K < 2 or self.K % 2:
            raise ValueError('inner_dims must be even (1x2 stride-2 window)')

    @property
    def P(self):                # CFFM.py:130
        return int(self.F * (self.F - 1) / 2)

    @property
    def Lc(self):               # CFFM.py:373
        return int(math.log(self.D, 2))

    @property
    def live_layers(self):      # layer Lc-1 never reaches t1 (CFFM.py:394-396)
        return self.Lc - 1

    @property
    def t1_width(self):
        return 2 * self.D - 2

    @property
    def act_id(self):
        return ACTIVATIONS.index(self.activation)


def pair_lists(F):
    """(i, j) of pair p in the reference's row-major i<j order (CFFM.py:304-305)."""
    ii = [i for i in range(F) for _ in range(i + 1, F)]
    jj = [j for i in range(F) for j in range(i + 1, F)]
    return ii, jj


def param_shapes(cfg):
    P, Lc, F, K, D, M = cfg.P, cfg.Lc, cfg.F, cfg.K, cfg.D, cfg.M
    s = {
        'inner_embeddings': (M, K), 'outer_embeddings': (M, D), 'feature_bias': (M, 1),
        'outer_W': (P, 1), 'outer_b': (1,),
        'bias_W': (F, F), 'bias_b': (F,), 'bias': (),
        'inner_layer_conv_weight_0': (1, 2, 1, 2), 'inner_layer_conv_bias_0': (2,),
        'dense_kernel': (P * K, 1), 'dense_bias': (1,),
        'dense_1_kernel': (2 * D - 2, 32), 'dense_1_bias': (32,),
        'dense_2_kernel': (32, 1), 'dense_2_bias': (1,),
        'dense_3_kernel': (F, 1), 'dense_3_bias': (1,),
    }
    for l in range(Lc):
        s['outer_layer_conv_weight_%d' % l] = (2, 2, P, P)
        s['outer_layer_conv_bias_%d' % l] = (P,)
    return s


TABLES = ('inner_embeddings', 'outer_embeddings', 'feature_bias')
UNTRAINED = ('outer_W', 'outer_b')   # created, counted, never used (CFFM.py:271-272, :412)


def logged_param_count(cfg):
    """The number ``calculate_parameters`` logs (CFFM.py:543-553): members of self.weights only."""
    n = 0
    for name, shp in param_shapes(cfg).items():
        if name.startswith('dense'):
            continue
        n += int(np.prod(shp)) if shp else 1
    return n


def _trunc_normal(rng, shape, std=1.0):
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2.0
    while bad.any():                     # tf.truncated_normal: re-draw beyond two sigma (CFFM.py:460)
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2.0
    return x * std


def _glorot_uniform(rng, shape):
    lim = math.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-lim, lim, size=shape)


def init_params(cfg, seed=2021, dtype=np.float32, tables=True):
    """Parameter dict with the reference's initial distributions.  The reference seeds
    nothing; a seed is taken here so that runs and tests are repeatable.  tables=False leaves the three
    embedding tables out (HipEngine(params='device') draws them on the GPU)."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shp in param_shapes(cfg).items():
        if not tables and name in TABLES:
            continue
        if name == 'inner_embeddings':
            v = rng.standard_normal(shp) * 0.1
        elif name == 'outer_embeddings':
            v = rng.standard_normal(shp) * 0.01
        elif name == 'feature_bias':
            v = np.zeros(shp)                          # random_normal(stddev=0) -> exactly 0
        elif name == 'bias':
            v = np.zeros(shp)
        elif name.endswith('conv_bias_0') or 'conv_bias_' in name:
            v = np.full(shp, 0.01)                     # bias_variable (CFFM.py:466)
        elif name.startswith('dense') and name.endswith('kernel'):
            v = _glorot_uniform(rng, shp)              # tf.layers.dense default
        elif name.startswith('dense'):
            v = np.zeros(shp)
        else:                                          # weight_variable: trunc-N(0,1)
            v = _trunc_normal(rng, shp)
        p[name] = np.ascontiguousarray(v, dtype=dtype)
    return p
