"""CPU restatement of the CFFM hot path (TEST INFRASTRUCTURE - not product code).

This file is the *oracle*: a plain numpy, op-by-op, unfused, NHWC restatement of the
reference graph.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it; the product path (``cffm_amd``) never does and fails loudly when the HIP
library is missing.

PARITY UNPINNED: the reference arithmetic lives in tensorflow==1.14.0 (reference README.md:9), which
is neither in /root/reference nor installable here, and the reference ships no tests, golden vectors
or seeds.  This restatement follows the call sites of /root/reference/CFFM.py line by line (cited on
each function) plus the TF-1.14 op definitions written down in SURVEY.md Appendix A; it is
cross-checked against an independent torch-autograd formulation in tests/test_oracle.py, but it is
not pinned against TensorFlow output.

Conventions
-----------
``p``   : dict of numpy arrays (see ``param_shapes``)
``cfg`` : any object with attributes M F K D activation lamda_att beta_outer linear_att inner_conv
          outer_conv (``cffm_amd.spec.CFFMConfig`` satisfies this)
All maths runs in the dtype of the parameter arrays (float64 for validation, float32 for timing).
"""
import math

import numpy as np

try:  # vectorised erf for gelu (scipy is present in the image; fall back to math.erf)
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)

SELU_SCALE = 1.0507009873554804934193349852946
SELU_SCALE_ALPHA = 1.7580993408473768599402175208123
PRELU_ALPHA = 0.25  # CFFM.py:153-155


# --------------------------------------------------------------------------------------------
# shapes (CFFM.py:239-293, :323, :375-377, :339, :409-410, :441;  SURVEY A.2)
# --------------------------------------------------------------------------------------------
def num_pairs(F):
    return int(F * (F - 1) / 2)  # CFFM.py:130


def conv_depth(D):
    return int(math.log(D, 2))  # CFFM.py:373


def pair_index(F):
    """Row-major (i<j) pair list, CFFM.py:304-305 / :355-356."""
    ii, jj = [], []
    for i in range(F):
        for j in range(i + 1, F):
            ii.append(i)
            jj.append(j)
    return np.asarray(ii, dtype=np.int64), np.asarray(jj, dtype=np.int64)


def param_shapes(M, F, K, D):
    P, Lc = num_pairs(F), conv_depth(D)
    s = {
        'inner_embeddings': (M, K),            # :257
        'outer_embeddings': (M, D),            # :264
        'outer_W': (P, 1), 'outer_b': (1,),    # :271-272 (never used, :412 commented)
        'feature_bias': (M, 1),                # :276
        'bias_W': (F, F), 'bias_b': (F,),      # :281-282 (att_dim == F)
        'bias': (),                            # :284
        'inner_layer_conv_weight_0': (1, 2, 1, 2), 'inner_layer_conv_bias_0': (2,),  # :323
        'dense_kernel': (P * K, 1), 'dense_bias': (1,),             # :339 inner head
        'dense_1_kernel': (2 * D - 2, 32), 'dense_1_bias': (32,),   # :409
        'dense_2_kernel': (32, 1), 'dense_2_bias': (1,),            # :410
        'dense_3_kernel': (F, 1), 'dense_3_bias': (1,),             # :441
    }
    for l in range(Lc):
        s['outer_layer_conv_weight_%d' % l] = (2, 2, P, P)          # :375-377
        s['outer_layer_conv_bias_%d' % l] = (P,)
    return s


def count_logged_params(M, F, K, D):
    """What ``calculate_parameters`` logs (CFFM.py:543-553): self.weights only - the four
    tf.layers.dense layers are not in that dict, the unused outer_W/outer_b and the dead last
    conv layer are."""
    P, Lc = num_pairs(F), conv_depth(D)
    return M * K + M * D + P + 1 + M + F * F + F + 1 + 6 + Lc * (4 * P * P + P)


# --------------------------------------------------------------------------------------------
# activations (CFFM.py:132-155; TF-1.14 functor definitions)
# --------------------------------------------------------------------------------------------
def relu(x):
    return np.maximum(x, 0)


def act(x, kind):
    if kind == 'relu':
        return np.maximum(x, 0)
    if kind == 'elu':
        return np.where(x < 0, np.exp(np.minimum(x, 0)) - 1, x)
    if kind == 'selu':
        return np.where(x < 0, SELU_SCALE_ALPHA * (np.exp(np.minimum(x, 0)) - 1), SELU_SCALE * x)
    if kind == 'prelu':  # relu(x) + alpha * (-relu(-x))
        return np.maximum(x, 0) + PRELU_ALPHA * (-np.maximum(-x, 0))
    if kind == 'gelu':   # x * 0.5 * (1 + erf(x / sqrt(2)))
        return x * (0.5 * (1.0 + _erf(x / math.sqrt(2.0)))).astype(x.dtype)
    raise ValueError('unknown activation %r' % (kind,))


def act_grad(x, kind):
    """d act / d x as TF-1.14 autodiff produces it (ReluGrad/EluGrad/SeluGrad use the sign of the
    output; at 0 relu and prelu give 0, elu gives 1, selu gives scale)."""
    if kind == 'relu':
        return (x > 0).astype(x.dtype)
    if kind == 'elu':
        y = act(x, 'elu')
        return np.where(y < 0, y + 1, np.ones_like(x))
    if kind == 'selu':
        y = act(x, 'selu')
        return np.where(y < 0, y + SELU_SCALE_ALPHA, np.full_like(x, SELU_SCALE))
    if kind == 'prelu':
        return (x > 0).astype(x.dtype) + PRELU_ALPHA * (x < 0).astype(x.dtype)
    if kind == 'gelu':
        cdf = 0.5 * (1.0 + _erf(x / math.sqrt(2.0)))
        pdf = np.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)
        return (cdf + x * pdf).astype(x.dtype)
    raise ValueError('unknown activation %r' % (kind,))


# --------------------------------------------------------------------------------------------
# forward (CFFM.py:296-453)
# --------------------------------------------------------------------------------------------
def _im2col_2x2(A):
    """[B,S,S,C] -> [B,S/2,S/2,4C] with the last axis ordered (dh,dw,c): the VALID 2x2/stride-2
    patches of tf.nn.conv2d (CFFM.py:385-386) against a filter reshaped [4C, Cout]."""
    B, S, _, C = A.shape
    h = S // 2
    return A.reshape(B, h, 2, h, 2, C).transpose(0, 1, 3, 2, 4, 5).reshape(B, h, h, 4 * C)


def _col2im_2x2(G, C):
    """Inverse of _im2col_2x2 (patches never overlap)."""
    B, h, _, _ = G.shape
    return G.reshape(B, h, h, 2, 2, C).transpose(0, 1, 3, 2, 4, 5).reshape(B, 2 * h, 2 * h, C)


def forward(p, X, cfg, keep_cache=True):
    """Returns (out[B], cache).  X: int array [B,F]."""
    X = np.asarray(X)
    B, F = X.shape
    K, D, kind = cfg.K, cfg.D, cfg.activation
    P = num_pairs(F)
    ii, jj = pair_index(F)
    dt = p['feature_bias'].dtype
    c = {'X': X}
    out = np.zeros((B,), dtype=dt)

    # ---- inner convolution component, CFFM.py:301-343 -----------------------------------
    if cfg.inner_conv == 1:
        Ei = p['inner_embeddings'][X]                      # :303  [B,F,K]
        I = Ei[:, ii, :] * Ei[:, jj, :]                    # :304-310 -> [B,P,K] (:313-317 layout)
        x = act(I, kind)                                   # :319
        w = p['inner_layer_conv_weight_0'].reshape(2, 2)   # HWIO [1,2,1,2] -> [tap, ch]
        bc = p['inner_layer_conv_bias_0']
        x0, x1 = x[:, :, 0::2], x[:, :, 1::2]              # the two taps of the 1x2/stride-2 window
        z = x0[..., None] * w[0] + x1[..., None] * w[1] + bc   # :327 conv2d + bias  [B,P,K/2,2]
        r = relu(z)                                        # :478
        cc = act(r, kind)                                  # :330
        mp = np.maximum(x0, x1)                            # :331 max-pool on inner_input
        s = cc + mp[..., None]                             # :332
        flat = s.reshape(B, P * K)                         # :333 (16 -> K/2)
        inner_out = flat @ p['dense_kernel'][:, 0] + p['dense_bias'][0]   # :339
        out = out + inner_out
        c.update(Ei=Ei, I=I, x=x, z=z, r=r, flat=flat, inner_out=inner_out)

    # ---- outer convolution component, CFFM.py:348-418 -----------------------------------
    if cfg.outer_conv == 1:
        Lc = conv_depth(D)
        Eo = p['outer_embeddings'][X]                                  # :354  [B,F,D]
        A = Eo[:, ii, :, None] * Eo[:, jj, None, :]                    # :355-362 [B,P,D,D]
        A = np.ascontiguousarray(A.transpose(0, 2, 3, 1))              # :365-367 [B,D,D,P]
        pools = [A.sum(axis=(2, 3))]                                   # :381
        acts_in, zs, rs = [], [], []
        for l in range(Lc - 1):          # layer Lc-1 is dead code (:394-396 never reads its pool)
            W = p['outer_layer_conv_weight_%d' % l].reshape(4 * P, P)
            patches = _im2col_2x2(A)
            z = patches @ W + p['outer_layer_conv_bias_%d' % l]        # :385-386 conv2d+bias
            r = relu(z)                                                # :478
            acts_in.append(patches if keep_cache else None)
            zs.append(z if keep_cache else None)
            rs.append(r)
            A = act(r, kind)                                           # :387
            pools.append(A.sum(axis=(2, 3)))                           # :390-391
        t1 = np.concatenate(pools[:Lc], axis=1)                        # :394-396 [B,2D-2]
        h1 = t1 @ p['dense_1_kernel'] + p['dense_1_bias']              # :409
        o = h1 @ p['dense_2_kernel'][:, 0] + p['dense_2_bias'][0]      # :410
        outer_out = (cfg.beta_outer * o).astype(dt)                    # :414
        out = out + outer_out
        c.update(Eo=Eo, patches=acts_in, zs=zs, rs=rs, t1=t1, h1=h1, outer_out=outer_out, pools=pools)

    # ---- first-order term, CFFM.py:422-446 ----------------------------------------------
    fb = p['feature_bias'][X][:, :, 0]                                 # :422,:425 [B,F]
    if cfg.linear_att == 1:
        zl = (fb @ p['bias_W'] + p['bias_b']) / cfg.lamda_att          # :432-434
        zl = zl - zl.max(axis=1, keepdims=True)
        e = np.exp(zl)
        a = e / e.sum(axis=1, keepdims=True)                           # :436
        g = fb * a                                                     # :438
        lin = g @ p['dense_3_kernel'][:, 0] + p['dense_3_bias'][0]     # :441
        c.update(a=a)
    else:
        lin = fb.sum(axis=1)                                           # :444
    out = out + lin
    out = out + p['bias']                                              # :449-453
    c.update(fb=fb, lin=lin, out=out)
    return out.astype(dt), c


# --------------------------------------------------------------------------------------------
# loss (CFFM.py:486-514).  Returns (loss, dL/dout[B], out_for_eval)
# --------------------------------------------------------------------------------------------
def loss_and_grad(out, y, cfg, p=None):
    B = out.shape[0]
    lt = cfg.loss_type
    lam = getattr(cfg, 'lamda_bilinear', 0.0)
    if lt == 'square_loss' and not lam > 0:
        mse = np.mean((y - out) ** 2)
        L = np.sqrt(mse + 1e-10)                                       # :493
        return L, (out - y) / (B * L)
    if lt == 'square_loss':
        # :489-491: l2_loss = sum(t^2)/2 ; l2_regularizer(s)(w) = s * sum(w^2)/2 ;
        # the outer table is scaled by lamda_att (quirk Q13).  Table gradients: regularisation_grads()
        L = 0.5 * np.sum((y - out) ** 2)
        if p is not None:
            L = L + lam * 0.5 * np.sum(p['inner_embeddings'] ** 2) \
                  + cfg.lamda_att * 0.5 * np.sum(p['outer_embeddings'] ** 2)
        return L, (out - y)
    if lt == 'mse':
        return np.mean((y - out) ** 2), 2.0 * (out - y) / B            # :506
    if lt == 'mae':
        return np.mean(np.abs(y - out)), np.sign(out - y) / B          # :508
    if lt == 'log_loss':                                               # :495-504 (lamda == 0 branch)
        s = 1.0 / (1.0 + np.exp(-out))
        eps = 1e-7
        L = -np.mean(y * np.log(s + eps) + (1 - y) * np.log(1 - s + eps))
        dLds = -(y / (s + eps) - (1 - y) / (1 - s + eps)) / B
        return L, dLds * s * (1 - s)
    if lt == 'hybrid':                                                 # :510-513, log_loss on the RAW out
        eps = 1e-7
        with np.errstate(invalid='ignore', divide='ignore'):
            ll = -np.mean(y * np.log(out + eps) + (1 - y) * np.log(1 - out + eps))
        L = 0.5 * 0.5 * np.sum((y - out) ** 2) + 0.5 * ll
        return L, 0.5 * (out - y) - 0.5 * (y / (out + eps) - (1 - y) / (1 - out + eps)) / B
    raise ValueError('unsupported loss_type %r' % (lt,))


# --------------------------------------------------------------------------------------------
# backward (what Optimizer.minimize differentiates, CFFM.py:517-529)
# --------------------------------------------------------------------------------------------
def backward(p, cache, dout, cfg):
    """dout: dL/dout [B].  Returns dict of gradients.  Table gradients come back as the
    IndexedSlices TF would build: ids = X.reshape(B*F), values 'd_inner_rows' [B,F,K],
    'd_outer_rows' [B,F,D], 'd_bias_rows' [B,F]."""
    X = cache['X']
    B, F = X.shape
    K, D, kind = cfg.K, cfg.D, cfg.activation
    P = num_pairs(F)
    ii, jj = pair_index(F)
    dt = dout.dtype
    g = {'bias': dout.sum()}

    if cfg.inner_conv == 1:
        Ei, I, x, z, r = cache['Ei'], cache['I'], cache['x'], cache['z'], cache['r']
        w = p['inner_layer_conv_weight_0'].reshape(2, 2)
        g['dense_bias'] = np.array([dout.sum()], dtype=dt)
        g['dense_kernel'] = (cache['flat'].T @ dout)[:, None]
        ds = (dout[:, None] * p['dense_kernel'][:, 0][None, :]).reshape(B, P, K // 2, 2)
        # s = act(relu(z)) + maxpool(x)
        dz = ds * act_grad(r, kind) * (r > 0)
        x0, x1 = x[:, :, 0::2], x[:, :, 1::2]
        g['inner_layer_conv_bias_0'] = dz.sum(axis=(0, 1, 2))
        gw = np.stack([(dz * x0[..., None]).sum(axis=(0, 1, 2)),
                       (dz * x1[..., None]).sum(axis=(0, 1, 2))])      # [tap, ch]
        g['inner_layer_conv_weight_0'] = gw.reshape(1, 2, 1, 2)
        dmp = ds.sum(axis=3)                                            # broadcast add over ch
        first = x0 >= x1                                                # max-pool: first element on ties
        first = cache.get('first_override', first)                      # tests: either decision on a near-tie
        dx0 = dz @ w[0] + np.where(first, dmp, 0)
        dx1 = dz @ w[1] + np.where(first, 0, dmp)
        dx = np.empty_like(x)
        dx[:, :, 0::2], dx[:, :, 1::2] = dx0, dx1
        dI = dx * act_grad(I, kind)
        dEi = np.zeros_like(Ei)
        np.add.at(dEi, (slice(None), ii), dI * Ei[:, jj, :])
        np.add.at(dEi, (slice(None), jj), dI * Ei[:, ii, :])
        g['d_inner_rows'] = dEi

    if cfg.outer_conv == 1:
        Lc = conv_depth(D)
        Eo = cache['Eo']
        do = dout * cfg.beta_outer
        g['dense_2_bias'] = np.array([do.sum()], dtype=dt)
        g['dense_2_kernel'] = (cache['h1'].T @ do)[:, None]
        dh1 = do[:, None] * p['dense_2_kernel'][:, 0][None, :]
        g['dense_1_bias'] = dh1.sum(axis=0)
        g['dense_1_kernel'] = cache['t1'].T @ dh1
        dt1 = dh1 @ p['dense_1_kernel'].T                               # [B,2D-2]
        dpools, off = [], 0
        for l in range(Lc):
            wdt = D >> l
            dpools.append(dt1[:, off:off + wdt])
            off += wdt
        dA = None                                                       # grad wrt A_l (post-activation)
        g['_dt1'] = dt1
        g['_dC'] = {}
        for l in range(Lc - 2, -1, -1):
            S = D >> (l + 1)
            dAl = np.broadcast_to(dpools[l + 1][:, :, None, None], (B, S, S, P)).astype(dt)
            if dA is not None:
                dAl = dAl + dA
            r = cache['rs'][l]
            dz = dAl * act_grad(r, kind) * (r > 0)
            g['_dC'][l] = dz
            W = p['outer_layer_conv_weight_%d' % l].reshape(4 * P, P)
            patches = cache['patches'][l]
            g['outer_layer_conv_bias_%d' % l] = dz.sum(axis=(0, 1, 2))
            g['outer_layer_conv_weight_%d' % l] = (
                patches.reshape(-1, 4 * P).T @ dz.reshape(-1, P)).reshape(2, 2, P, P)
            dA = _col2im_2x2(dz @ W.T, P)                               # grad wrt the layer input
        dAm1 = np.broadcast_to(dpools[0][:, :, None, None], (B, D, D, P)).astype(dt)
        if dA is not None:
            dAm1 = dAm1 + dA
        # A_-1[b,h,w,p] = Eo[b,i_p,h] * Eo[b,j_p,w]
        dEo = np.zeros_like(Eo)
        gi = np.einsum('bhwp,bpw->bph', dAm1, Eo[:, jj, :])
        gj = np.einsum('bhwp,bph->bpw', dAm1, Eo[:, ii, :])
        np.add.at(dEo, (slice(None), ii), gi)
        np.add.at(dEo, (slice(None), jj), gj)
        g['d_outer_rows'] = dEo

    fb = cache['fb']
    if cfg.linear_att == 1:
        a = cache['a']
        g['dense_3_bias'] = np.array([dout.sum()], dtype=dt)
        g['dense_3_kernel'] = ((fb * a).T @ dout)[:, None]
        dg = dout[:, None] * p['dense_3_kernel'][:, 0][None, :]
        da = dg * fb
        dz = a * (da - (da * a).sum(axis=1, keepdims=True)) / cfg.lamda_att
        g['bias_b'] = dz.sum(axis=0)
        g['bias_W'] = fb.T @ dz
        g['d_bias_rows'] = dg * a + dz @ p['bias_W'].T
    else:
        g['d_bias_rows'] = np.broadcast_to(dout[:, None], fb.shape).astype(dt)
    return g


# --------------------------------------------------------------------------------------------
# optimiser: TF-1.14 AdagradOptimizer(lr, initial_accumulator_value=1e-8)  (CFFM.py:523-524, A.5)
# --------------------------------------------------------------------------------------------
ADAGRAD_INIT_ACC = 1e-8


def adagrad_dense(v, acc, grad, lr):
    acc += grad * grad
    v -= lr * grad / np.sqrt(acc)


def adagrad_sparse(table, acc, ids, rows, lr):
    """Duplicate ids are summed FIRST (_apply_sparse_duplicate_indices), then one update per
    unique row; untouched rows and their accumulators stay as they were."""
    ids = np.asarray(ids).reshape(-1)
    rows = rows.reshape(ids.shape[0], -1)
    uniq, inv = np.unique(ids, return_inverse=True)
    summed = np.zeros((uniq.shape[0], rows.shape[1]), dtype=rows.dtype)
    np.add.at(summed, inv, rows)
    a = acc[uniq].reshape(uniq.shape[0], -1) + summed * summed
    acc[uniq] = a.reshape(acc[uniq].shape)
    table[uniq] = table[uniq] - (lr * summed / np.sqrt(a)).reshape(table[uniq].shape)


DENSE_TRAINED = ('bias', 'bias_W', 'bias_b', 'inner_layer_conv_weight_0', 'inner_layer_conv_bias_0',
                 'dense_kernel', 'dense_bias', 'dense_1_kernel', 'dense_1_bias', 'dense_2_kernel',
                 'dense_2_bias', 'dense_3_kernel', 'dense_3_bias')


def init_accumulators(p):
    return {k: np.full_like(v, ADAGRAD_INIT_ACC) for k, v in p.items()}


# --------------------------------------------------------------------------------------------
# the other optimizers of create_optimizer (CFFM.py:519-529), TF-1.14 semantics
#   GradientDescentOptimizer(lr)             v -= lr*g; sparse: duplicates accumulate (== summed)
#   MomentumOptimizer(lr, 0.95)              a = 0.95*a + g; v -= lr*a; sparse: only the touched rows (duplicates summed first)
#   AdamOptimizer(lr, 0.9, 0.999, 1e-8)      m,v moments; lr_t = lr*sqrt(1-b2^t)/(1-b1^t); v -= lr_t*m/(sqrt(v)+eps);
#                                            sparse (_apply_sparse_shared): m and v of EVERY row decay and every row moves
# --------------------------------------------------------------------------------------------
def init_opt_state(p, optimizer):
    if optimizer == 'AdagradOptimizer':
        return {'acc': init_accumulators(p)}
    z = lambda: {k: np.zeros_like(v) for k, v in p.items()}
    if optimizer == 'MomentumOptimizer':
        return {'acc': z()}
    if optimizer == 'AdamOptimizer':
        return {'m': z(), 'v': z(), 't': 0}
    return {}


def _dense_table_grad(p, name, ids, rows):
    g = np.zeros_like(p[name])
    np.add.at(g, ids, rows.reshape(ids.shape[0], -1).reshape((ids.shape[0],) + p[name].shape[1:]))
    return g


def apply_optimizer(p, st, g, X, cfg):
    opt, lr = cfg.optimizer, cfg.lr
    ids = np.asarray(X).reshape(-1)
    dense = {k: np.asarray(v).reshape(np.shape(p[k])) for k, v in g.items() if not k.startswith('d_') and not k.startswith('_')}
    tabs = {'inner_embeddings': g.get('d_inner_rows'), 'outer_embeddings': g.get('d_outer_rows'), 'feature_bias': g['d_bias_rows']}
    lam = getattr(cfg, 'lamda_bilinear', 0.0)
    if cfg.loss_type == 'square_loss' and lam > 0:
        # CFFM.py:489-491: the l2_regularizer terms add lamda * w (outer table: lamda_att * w, quirk Q13) to the table
        # gradients; IndexedSlices + dense aggregates to a DENSE gradient, so every optimizer of :517-529 treats the two
        # embedding tables as dense variables (feature_bias keeps its IndexedSlices gradient)
        for k, scale in (('inner_embeddings', lam), ('outer_embeddings', cfg.lamda_att)):
            dense[k] = _dense_table_grad(p, k, ids, tabs[k]) + scale * p[k]
            tabs[k] = None
    if opt == 'AdamOptimizer':
        st['t'] += 1
        b1, b2, eps, t = 0.9, 0.999, 1e-8, st['t']
        lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        for k, rows in tabs.items():
            if rows is not None:
                dense[k] = _dense_table_grad(p, k, ids, rows)          # every row takes part (zero gradient elsewhere)
        for k, gk in dense.items():
            st['m'][k] = b1 * st['m'][k] + (1 - b1) * gk
            st['v'][k] = b2 * st['v'][k] + (1 - b2) * gk * gk
            p[k] = p[k] - lr_t * st['m'][k] / (np.sqrt(st['v'][k]) + eps)
        return
    for k, gk in dense.items():
        if opt == 'GradientDescentOptimizer':
            p[k] = p[k] - lr * gk
        else:                                                          # Momentum
            st['acc'][k] = 0.95 * st['acc'][k] + gk
            p[k] = p[k] - lr * st['acc'][k]
    for k, rows in tabs.items():
        if rows is None:
            continue
        uniq, inv = np.unique(ids, return_inverse=True)
        summed = np.zeros((uniq.shape[0],) + p[k].shape[1:], dtype=p[k].dtype)
        np.add.at(summed, inv, rows.reshape((ids.shape[0],) + p[k].shape[1:]))
        if opt == 'GradientDescentOptimizer':
            p[k][uniq] = p[k][uniq] - lr * summed
        else:
            a = 0.95 * st['acc'][k][uniq] + summed
            st['acc'][k][uniq] = a
            p[k][uniq] = p[k][uniq] - lr * a


def train_step_opt(p, st, X, y, cfg, cache_hook=None):
    """train_step for any optimizer of CFFM.py:517-529 (st from init_opt_state)."""
    if cfg.optimizer == 'AdagradOptimizer':
        return train_step(p, st['acc'], X, y, cfg, cache_hook)
    out, cache = forward(p, X, cfg)
    if cache_hook is not None:
        cache_hook(cache)
    L, dout = loss_and_grad(out, y.astype(out.dtype), cfg, p)
    g = backward(p, cache, dout.astype(out.dtype), cfg)
    apply_optimizer(p, st, g, X, cfg)
    return L, out


def train_step(p, acc, X, y, cfg, cache_hook=None):
    """One ``sess.run((loss, optimizer))`` (CFFM.py:200): forward, loss, backward, Adagrad.
    Mutates p and acc in place; returns (loss, out_before_update).  ``cache_hook(cache)`` lets a test
    adopt the device's relu decisions for pre-activations that sit on the kink (see tests)."""
    out, cache = forward(p, X, cfg)
    if cache_hook is not None:
        cache_hook(cache)
    L, dout = loss_and_grad(out, y.astype(out.dtype), cfg, p)
    g = backward(p, cache, dout.astype(out.dtype), cfg)
    lr = cfg.lr
    for name, grad in g.items():
        if name.startswith('d_') or name.startswith('_'):
            continue
        if name == 'bias':
            a = acc['bias'] + grad * grad
            acc['bias'] = a
            p['bias'] = p['bias'] - lr * grad / np.sqrt(a)
        else:
            adagrad_dense(p[name], acc[name], grad.reshape(p[name].shape), lr)
    ids = np.asarray(X).reshape(-1)
    lam = getattr(cfg, 'lamda_bilinear', 0.0)
    if cfg.loss_type == 'square_loss' and lam > 0:
        # CFFM.py:489-491: the l2_regularizer terms make the table gradients DENSE (IndexedSlices + dense is
        # aggregated to a dense tensor by TF), so Adagrad updates every row and every accumulator; the outer
        # table is scaled by lamda_att (quirk Q13)
        for name, key, scale in (('inner_embeddings', 'd_inner_rows', lam), ('outer_embeddings', 'd_outer_rows', cfg.lamda_att)):
            if key in g:
                dense = scale * p[name]
                np.add.at(dense, ids, g[key].reshape(ids.shape[0], -1))
                adagrad_dense(p[name], acc[name], dense, lr)
        adagrad_sparse(p['feature_bias'], acc['feature_bias'], ids, g['d_bias_rows'], lr)
        return L, out
    if 'd_inner_rows' in g:
        adagrad_sparse(p['inner_embeddings'], acc['inner_embeddings'], ids, g['d_inner_rows'], lr)
    if 'd_outer_rows' in g:
        adagrad_sparse(p['outer_embeddings'], acc['outer_embeddings'], ids, g['d_outer_rows'], lr)
    adagrad_sparse(p['feature_bias'], acc['feature_bias'], ids, g['d_bias_rows'], lr)
    return L, out


# --------------------------------------------------------------------------------------------
# evaluation metrics (CFFM.py:607-615) - numpy restatement of clip + sklearn RMSE / R2
# --------------------------------------------------------------------------------------------
def clipped_rmse_r2(y_pred, y_true):
    y_true = np.asarray(y_true, dtype=np.float64)
    yp = np.minimum(np.maximum(np.asarray(y_pred, dtype=np.float64), y_true.min()), y_true.max())
    rmse = math.sqrt(np.mean((y_true - yp) ** 2))
    ss_res = np.sum((y_true - yp) ** 2)
    ss_tot = np.sum((y_true - y_true.mean()) ** 2)
    return rmse, 1.0 - ss_res / ss_tot


def eva_termination(valid):  # CFFM.py:631-635
    if len(valid) > 5:
        if valid[-1] > valid[-2] > valid[-3] > valid[-4] > valid[-5]:
            return True
    return False
