"""Independent torch-autograd formulation of the CFFM graph (TEST INFRASTRUCTURE, like the rest of oracle/: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; PARITY UNPINNED against TF-1.14, see
cffm_oracle.py).

Written against the reference call sites (CFFM.py:296-453) with torch's own conv2d / max_pool2d /
softmax / autograd, i.e. NOT sharing code or layout tricks with oracle/cffm_oracle.py (which uses
im2col matmuls and a hand-derived backward).  Two independent readings of the same graph are the only
defence against a shared misreading, since TF-1.14 cannot run here.
"""
import math

import torch
import torch.nn.functional as Fn

SELU_SCALE = 1.0507009873554804934193349852946
SELU_ALPHA = 1.6732632423543772848170429916717


def act(x, kind):
    if kind == 'relu':
        return torch.relu(x)
    if kind == 'elu':
        return Fn.elu(x)
    if kind == 'selu':
        return SELU_SCALE * torch.where(x > 0, x, SELU_ALPHA * (torch.exp(torch.clamp(x, max=0)) - 1))
    if kind == 'prelu':
        return torch.relu(x) + 0.25 * (-torch.relu(-x))
    if kind == 'gelu':
        return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))
    raise ValueError(kind)


def forward(p, X, cfg):
    """p: dict name -> torch tensor (requires_grad as wanted); X: LongTensor [B,F]."""
    B, F = X.shape
    K, D, kind = cfg.K, cfg.D, cfg.activation
    parts = []
    if cfg.inner_conv == 1:
        emb = p['inner_embeddings'][X]                                   # [B,F,K]
        prods = [emb[:, i, :] * emb[:, j, :] for i in range(F) for j in range(i + 1, F)]
        inp = torch.stack(prods).permute(1, 0, 2).unsqueeze(-1)          # NHWC [B,P,K,1]
        inp = act(inp, kind)
        nchw = inp.permute(0, 3, 1, 2)                                   # [B,1,P,K]
        w = p['inner_layer_conv_weight_0'].permute(3, 2, 0, 1)           # HWIO -> OIHW [2,1,1,2]
        conv = Fn.conv2d(nchw, w, bias=p['inner_layer_conv_bias_0'], stride=(1, 2))
        conv = act(torch.relu(conv), kind)                               # [B,2,P,K/2]
        mp = Fn.max_pool2d(nchw, kernel_size=(1, 2), stride=(1, 2))      # [B,1,P,K/2]
        s = (conv + mp).permute(0, 2, 3, 1)                              # NHWC [B,P,K/2,2]
        flat = s.reshape(B, -1)
        parts.append(flat @ p['dense_kernel'] + p['dense_bias'])
    if cfg.outer_conv == 1:
        emb = p['outer_embeddings'][X]
        outs = [emb[:, i, :].unsqueeze(-1) * emb[:, j, :].unsqueeze(1)
                for i in range(F) for j in range(i + 1, F)]              # P x [B,D,D]
        cur = torch.stack(outs).permute(1, 0, 2, 3)                      # NCHW [B,P,D,D]
        Lc = int(math.log(D, 2))
        pools = [cur.sum(dim=(1, 3))]                                    # NHWC axes [2,3] = (W, C)
        for l in range(Lc):
            w = p['outer_layer_conv_weight_%d' % l].permute(3, 2, 0, 1)  # HWIO -> OIHW
            cur = Fn.conv2d(cur, w, bias=p['outer_layer_conv_bias_%d' % l], stride=2)
            cur = act(torch.relu(cur), kind)
            pools.append(cur.sum(dim=(1, 3)))
        t1 = torch.cat(pools[:Lc], dim=1)
        h = t1 @ p['dense_1_kernel'] + p['dense_1_bias']
        parts.append(cfg.beta_outer * (h @ p['dense_2_kernel'] + p['dense_2_bias']))
    fb = p['feature_bias'][X].squeeze(-1)                                # [B,F]
    if cfg.linear_att == 1:
        a = torch.softmax((fb @ p['bias_W'] + p['bias_b']) / cfg.lamda_att, dim=-1)
        parts.append((fb * a) @ p['dense_3_kernel'] + p['dense_3_bias'])
    else:
        parts.append(fb.sum(dim=1, keepdim=True))
    parts.append(p['bias'] * torch.ones(B, 1, dtype=fb.dtype))
    return sum(parts).reshape(B)


def loss(out, y, cfg, p=None):
    if cfg.loss_type == 'square_loss' and not cfg.lamda_bilinear > 0:
        return torch.sqrt(torch.mean((y - out) ** 2) + 1e-10)
    if cfg.loss_type == 'square_loss':
        # tf.nn.l2_loss(t) = sum(t^2)/2; l2_regularizer(s)(w) = s * l2_loss(w)   (CFFM.py:489-491)
        return 0.5 * torch.sum((y - out) ** 2) + cfg.lamda_bilinear * 0.5 * torch.sum(p['inner_embeddings'] ** 2) \
            + cfg.lamda_att * 0.5 * torch.sum(p['outer_embeddings'] ** 2)
    if cfg.loss_type == 'mse':
        return torch.mean((y - out) ** 2)
    if cfg.loss_type == 'mae':
        return torch.mean(torch.abs(y - out))
    if cfg.loss_type == 'log_loss':
        s = torch.sigmoid(out)
        return -torch.mean(y * torch.log(s + 1e-7) + (1 - y) * torch.log(1 - s + 1e-7))
    if cfg.loss_type == 'hybrid':
        ll = -torch.mean(y * torch.log(out + 1e-7) + (1 - y) * torch.log(1 - out + 1e-7))
        return 0.5 * (0.5 * torch.sum((y - out) ** 2)) + 0.5 * ll
    raise ValueError(cfg.loss_type)


TABLES = ('inner_embeddings', 'outer_embeddings', 'feature_bias')


def train_step(p, acc, X, y, cfg):
    """One Adagrad step (CFFM.py:200, :523-524) on torch CPU tensors: autograd for the gradients, then the TF update
    (acc += g^2; v -= lr*g/sqrt(acc)) on every dense variable and on the LOOKED-UP rows of the three tables (the
    autograd table gradient already sums duplicate ids, as TF's sparse apply does).  bench.py times this on the host
    cores as the multi-threaded op-by-op CPU stand-in for the TF1 graph (conv2d / max_pool2d / matmul, NCHW outer map
    materialised like the reference's)."""
    for v in p.values():
        v.grad = None
    out = forward(p, X, cfg)
    L = loss(out, y, cfg, p)
    L.backward()
    touched = torch.zeros(cfg.M, dtype=torch.bool)
    touched[X.reshape(-1)] = True
    with torch.no_grad():
        for k, v in p.items():
            if v.grad is None:
                continue
            g = v.grad
            if k in TABLES:
                rows = touched.nonzero().reshape(-1)
                gr = g[rows]
                a = acc[k][rows] + gr * gr
                acc[k][rows] = a
                v[rows] -= cfg.lr * gr / torch.sqrt(a)
            else:
                acc[k] += g * g
                v -= cfg.lr * g / torch.sqrt(acc[k])
    return float(L.detach())
