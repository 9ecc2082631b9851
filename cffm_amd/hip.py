"""Python binding of libcffm_hip.so (the C ABI declared in include/cffm_hip.h).

Two equivalent bindings of the same entry points:

* ``fast()``  - the thin pybind11 module ``cffm_amd/lib/_cffm_pybind*.so`` (csrc_host/pybind_module.cpp, built by
  ``make``): what the engine's per-step calls go through.  Every pointer is passed as a plain integer
  (``tensor.data_ptr()``, ``ctypes.addressof(struct)``).
* ``load()``  - ctypes (``PROTOTYPES`` below): used for the layout queries, by the tools and as the documented
  alternative binding; it accepts the same integer arguments, so ``fast()`` falls back to it when the
  pybind11 module has not been built.

There is deliberately no CPU fallback: if the library is missing or a call fails this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CFFM_HIP_LIB: another build of the library for the ctypes handle.  NOT enough for A/B timing of kernel variants: fast() calls go through
# _cffm_pybind, which binds to the libcffm_hip.so next to it (RUNPATH $ORIGIN) - point CFFM_HOST_LIB_DIR at a directory that holds the
# other library AND a pybind11 module linked against it (tools/experiments/ab_old_new.sh checks /proc/self/maps for exactly this).
LIB_PATH = os.environ.get('CFFM_HIP_LIB') or os.path.join(_HERE, 'lib', 'libcffm_hip.so')

ABI_VERSION = 9
MAX_LAYERS = 8
NSLAB = 64
HEAD_UNITS = 32
LOSS_IDS = {'square_loss': 0, 'mse': 1, 'mae': 2, 'log_loss': 3, 'hybrid': 5}
OPT_IDS = {'AdagradOptimizer': 0, 'GradientDescentOptimizer': 1, 'MomentumOptimizer': 2, 'AdamOptimizer': 3}


class Shape(C.Structure):
    _fields_ = [('M', C.c_int32), ('F', C.c_int32), ('K', C.c_int32), ('D', C.c_int32), ('act', C.c_int32),
                ('linear_att', C.c_int32), ('inner_conv', C.c_int32), ('outer_conv', C.c_int32),
                ('loss', C.c_int32), ('lamda_att', C.c_float), ('beta_outer', C.c_float), ('lr', C.c_float),
                ('lamda', C.c_float), ('optimizer', C.c_int32)]


class ThetaLayout(C.Structure):
    _fields_ = [('n', C.c_int64), ('att_W', C.c_int64), ('att_b', C.c_int64), ('bias', C.c_int64),
                ('inner_cw', C.c_int64), ('inner_cb', C.c_int64), ('inner_dw', C.c_int64), ('inner_db', C.c_int64),
                ('conv_w', C.c_int64 * MAX_LAYERS), ('conv_b', C.c_int64 * MAX_LAYERS),
                ('d1_w', C.c_int64), ('d1_b', C.c_int64), ('d2_w', C.c_int64), ('d2_b', C.c_int64),
                ('lin_w', C.c_int64), ('lin_b', C.c_int64),
                ('P', C.c_int32), ('Pp', C.c_int32), ('Lc', C.c_int32), ('live', C.c_int32)]


class WsLayout(C.Structure):
    _fields_ = [('bytes', C.c_int64), ('Ei', C.c_int64), ('Eo', C.c_int64), ('fb', C.c_int64),
                ('inner_out', C.c_int64), ('C', C.c_int64 * MAX_LAYERS),
                ('t1', C.c_int64), ('h1', C.c_int64), ('att', C.c_int64), ('out', C.c_int64),
                ('sqerr', C.c_int64), ('scalars', C.c_int64), ('dout', C.c_int64), ('dt1', C.c_int64),
                ('dC', C.c_int64 * MAX_LAYERS), ('dEi', C.c_int64), ('dEo', C.c_int64), ('dfb', C.c_int64),
                ('gpart', C.c_int64), ('gpart_floats', C.c_int64), ('sort_keys', C.c_int64), ('sort_vals', C.c_int64),
                ('sort_tmp', C.c_int64), ('sort_tmp_bytes', C.c_int64), ('Gi', C.c_int64), ('Go', C.c_int64), ('Gfb', C.c_int64),
                ('pool', C.c_int64 * MAX_LAYERS), ('pool_np', C.c_int32 * MAX_LAYERS),
                ('w0pack', C.c_int64), ('w0pack_floats', C.c_int64), ('relu0', C.c_int64),
                ('wb3', C.c_int64), ('wb3_bytes', C.c_int64)]


class Tables(C.Structure):
    _fields_ = [('inner_emb', C.c_void_p), ('outer_emb', C.c_void_p), ('feat_bias', C.c_void_p)]


_P = C.c_void_p
_SH = C.c_void_p          # const cffm_shape_t*: byref(Shape) or its address as an int
_TB = C.c_void_p          # const cffm_tables_t*

# name -> (restype, argtypes); every symbol include/cffm_hip.h declares
PROTOTYPES = {
    'cffm_abi_version': (C.c_int, []),
    'cffm_error_string': (C.c_char_p, [C.c_int]),
    'cffm_theta_layout': (C.c_int, [_SH, _P]),
    'cffm_ws_layout': (C.c_int, [_SH, C.c_int32, _P]),
    'cffm_gather': (C.c_int, [_SH, _TB, _P, C.c_int32, _P, _P, _P, _P]),
    'cffm_gather_inner_fwd_ok': (C.c_int, [_SH]),
    'cffm_gather_inner_fwd': (C.c_int, [_SH, _TB, _P, _P, C.c_int32, _P, _P]),
    'cffm_inner_fwd': (C.c_int, [_SH, _P, _P, C.c_int32, _P]),
    'cffm_inner_bwd': (C.c_int, [_SH, _P, _P, C.c_int32, _P]),
    'cffm_outer_conv0_fwd': (C.c_int, [_SH, _P, _P, C.c_int32, _P]),
    'cffm_outer_conv0_bwd': (C.c_int, [_SH, _P, _P, C.c_int32, _P]),
    'cffm_conv_fwd': (C.c_int, [_SH, _P, _P, C.c_int32, C.c_int32, _P]),
    'cffm_conv_bwd': (C.c_int, [_SH, _P, _P, C.c_int32, C.c_int32, _P]),
    'cffm_head_fwd': (C.c_int, [_SH, _P, _P, _P, C.c_int32, _P]),
    'cffm_head_bwd': (C.c_int, [_SH, _P, _P, _P, C.c_int32, C.c_int64, _P]),
    'cffm_reduce_slabs': (C.c_int, [_SH, _P, C.c_int32, _P, _P]),
    'cffm_dense_adagrad': (C.c_int, [_P, _P, _P, C.c_int64, C.c_float, _P]),
    'cffm_sparse_adagrad': (C.c_int, [_SH, _TB, _TB, _P, C.c_int64, _P, _P, _P, _P, C.c_int32, _P]),
    'cffm_predict': (C.c_int, [_SH, _TB, _P, _P, C.c_int32, _P, _P, _P]),
    'cffm_forward': (C.c_int, [_SH, _TB, _P, _P, _P, C.c_int32, _P, _P]),
    'cffm_backward': (C.c_int, [_SH, _P, _P, C.c_int32, C.c_int64, _P, _P, _P]),
    'cffm_backward_unscaled': (C.c_int, [_SH, _P, _P, _P, C.c_int32, C.c_int64, _P, _P, _P, _P]),
    'cffm_dp_apply': (C.c_int, [_SH, _TB, _TB, _P, _P, _P, C.c_int64, _P, C.c_int64, _P, C.c_int32, _P, C.c_int32, _P]),
    'cffm_dp_runs_ok': (C.c_int, [_SH, C.c_int32]),
    'cffm_dp_dense_floats': (C.c_int64, [_SH]),
    'cffm_dp_local_dense': (C.c_int, [_SH, _TB, _P, _P, _P, C.c_int32, C.c_int64, _P, _P, _P]),
    'cffm_dp_apply_dense': (C.c_int, [_SH, _TB, _TB, _P, _P, _P, C.c_int64, _P, _P]),
    'cffm_dp_local': (C.c_int, [_SH, _TB, _P, _P, _P, C.c_int32, C.c_int64, _P, _P, _P, _P]),
    'cffm_packed_row_floats': (C.c_int32, [_SH]),
    'cffm_gather_packed': (C.c_int, [_SH, _TB, _P, C.c_int64, _P, _P]),
    'cffm_stage_packed': (C.c_int, [_SH, _P, _P, C.c_int64, C.c_int32, _P, _P]),
    'cffm_forward_packed': (C.c_int, [_SH, _P, _P, _P, C.c_int64, _P, C.c_int32, _P, _P]),
    'cffm_backward_unscaled_packed': (C.c_int, [_SH, _P, _P, _P, C.c_int64, _P, C.c_int32, C.c_int64, _P, _P, _P]),
    'cffm_pack_rows_dedup': (C.c_int, [_SH, _P, _P, _P, C.c_int32, _P, _P, _P]),
    'cffm_shard_plan_scratch_bytes': (C.c_int64, [C.c_int64]),
    'cffm_shard_plan': (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P]),
    'cffm_eval_scratch_bytes': (C.c_int64, []),
    'cffm_eval_sums': (C.c_int, [_P, _P, C.c_int64, C.c_float, C.c_float, _P, _P, _P]),
    'cffm_probe_copy': (C.c_int, [_P, _P, C.c_int64, _P]),
    'cffm_probe_read': (C.c_int, [_P, _P, C.c_int64, _P]),
    'cffm_probe_mfma': (C.c_int, [_P, C.c_int32, _P, _P]),
    'cffm_probe_mfma_bf16': (C.c_int, [_P, C.c_int32, _P, _P]),
    'cffm_train_step_opt': (C.c_int, [_SH, _TB, _TB, _TB, _P, _P, _P, _P, _P, _P, C.c_int32, _P, _P, C.c_int64, _P]),
    'cffm_train_step': (C.c_int, [_SH, _TB, _TB, _P, _P, _P, _P, _P, C.c_int32, _P, _P, _P]),
}

_lib = None


def load():
    """Load libcffm_hip.so once.  torch must be imported first so that the HIP runtime the library
    binds to (SONAME libamdhip64.so.7) is the one torch already loaded."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('cffm_amd: %s is missing - build it with `make` or __graft_entry__.build(); '
                           'there is no CPU fallback' % LIB_PATH)
    import torch  # noqa: F401  (loads libamdhip64 / libhsa-runtime64 into the process)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.cffm_abi_version() != ABI_VERSION:
        raise RuntimeError('cffm_amd: ABI version mismatch')
    _lib = lib
    return lib


_fast = None


def fast():
    """The per-step binding: the pybind11 module when it is built, else the ctypes library (same names, same integer
    arguments).  ``binding_name()`` says which one is in use."""
    global _fast
    if _fast is None:
        lib = load()                                   # libcffm_hip.so (and through torch, libamdhip64) is in the process
        try:
            alt = os.environ.get('CFFM_HOST_LIB_DIR')          # another build of the host shims (`make asan`)
            if alt:
                import glob
                import importlib.util
                spec = importlib.util.spec_from_file_location('_cffm_pybind', glob.glob(os.path.join(alt, '_cffm_pybind*.so'))[0])
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
            else:
                from .lib import _cffm_pybind as mod
            if mod.cffm_abi_version() != ABI_VERSION:
                raise ImportError('stale _cffm_pybind')
            missing = [n for n in PROTOTYPES if not hasattr(mod, n)]
            if missing:
                raise ImportError('_cffm_pybind lacks %s' % missing)
            _fast = mod
        except ImportError:
            _fast = lib
    return _fast


def binding_name():
    return 'pybind11' if fast() is not _lib else 'ctypes'


def check(rc):
    if rc != 0:
        msg = load().cffm_error_string(rc)
        raise RuntimeError('libcffm_hip: error %d: %s' % (rc, msg.decode() if msg else '?'))


def make_shape(cfg):
    if cfg.loss_type not in LOSS_IDS:
        raise ValueError('loss_type %r is not supported by the HIP path' % (cfg.loss_type,))
    loss = LOSS_IDS[cfg.loss_type]
    if cfg.loss_type == 'square_loss' and cfg.lamda_bilinear > 0:
        loss = 4                                    # CFFM_LOSS_SQUARE_L2
    return Shape(M=cfg.M, F=cfg.F, K=cfg.K, D=cfg.D, act=cfg.act_id, linear_att=cfg.linear_att,
                 inner_conv=cfg.inner_conv, outer_conv=cfg.outer_conv, loss=loss,
                 lamda_att=cfg.lamda_att, beta_outer=float(cfg.beta_outer), lr=cfg.lr,
                 lamda=float(cfg.lamda_bilinear), optimizer=OPT_IDS[cfg.optimizer])


def theta_layout(shape):
    tl = ThetaLayout()
    check(load().cffm_theta_layout(C.byref(shape), C.byref(tl)))
    return tl


def ws_layout(shape, B):
    wl = WsLayout()
    check(load().cffm_ws_layout(C.byref(shape), int(B), C.byref(wl)))
    return wl

