"""Where is the fp32 noise floor of the parity criterion?  (test infrastructure: imports oracle/ and tests/)

The GPU parity tests hold every tensor to close()'s bound 1e-5 * (|ref| + rms(ref)) against the float64 oracle and log the worst
err / bound per tensor (gpurun_out/parity_worst.json).  This tool runs THE SAME checks - check_backward_stages and step_check on the
same seeded cases, same slack terms - with a stand-in engine that is nothing but the numpy oracle evaluated in float32 (numpy /
BLAS summation orders, no HIP code): the ratios it logs are what ANY fp32 implementation of this graph scores against the fp64
oracle.  A device ratio at or below the float32-numpy ratio of the same tensor is the noise floor, not a kernel's summation order.

    python tools/fp32_floor.py [case ...]        # needs torch.cuda (the shared helpers move their inputs through .cuda())
writes gpurun_out/fp32_floor.json = {tensor: [worst ratio, case]}."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cffm_oracle as orc  # noqa: E402
from oracle import parity as P  # noqa: E402


class _TL(object):
    pass


class Fp32OracleEngine(object):
    """HipEngine's test-facing surface, computed by the numpy oracle in float32."""

    def __init__(self, cfg, p32):
        self.cfg = cfg
        self.p = {k: np.array(v, dtype=np.float32) for k, v in p32.items()}
        self.acc = {k: np.asarray(v, dtype=np.float32) for k, v in orc.init_accumulators(self.p).items()}
        self.tl = _TL()
        self.tl.Pp = (cfg.P + 15) // 16 * 16
        self.g = None

    def forward(self, ids, y=None):
        self.X = ids.cpu().numpy()
        self.y = None if y is None else y.cpu().numpy().astype(np.float32)
        self.out, self.c = orc.forward(self.p, self.X, self.cfg)

    def backward(self, y, B):
        self.L, self.dout = orc.loss_and_grad(self.out, self.y, self.cfg, self.p)
        self.g = orc.backward(self.p, self.c, self.dout, self.cfg)

    def ws_tensor(self, B, member, shape, dtype=torch.float32, index=None):
        c, g, Pp = self.c, self.g, self.tl.Pp
        if member == 'C':
            v = P.pad_channels(np.asarray(c['rs'][index]), Pp)
        elif member == 'dC':
            v = P.pad_channels(np.asarray(g['_dC'][index]), Pp)
        elif member == 'scalars':
            v = np.zeros(16, dtype=np.float32)
            v[1] = self.L
        else:
            v = {'inner_out': lambda: c['inner_out'], 'out': lambda: self.out, 'dout': lambda: self.dout, 't1': lambda: c['t1'],
                 'dt1': lambda: g['_dt1'], 'dEo': lambda: g['d_outer_rows'], 'dEi': lambda: g['d_inner_rows'],
                 'dfb': lambda: g['d_bias_rows']}[member]()
        return torch.from_numpy(np.ascontiguousarray(np.asarray(v, dtype=np.float32)).reshape(shape))

    def export_grad(self):
        return {k: np.asarray(v, dtype=np.float32) for k, v in self.g.items() if not k.startswith('d_') and not k.startswith('_')}

    def train_step(self, ids, y):
        L, _ = orc.train_step(self.p, self.acc, ids.cpu().numpy(), y.cpu().numpy().astype(np.float32), self.cfg)
        return torch.tensor([float(L)])

    def export_params(self):
        return {k: np.asarray(v).copy() for k, v in self.p.items() if k in self.acc}

    def export_accumulators(self):
        return {k: np.asarray(v).copy() for k, v in self.acc.items()}


def main(names):
    from tests import test_gpu_parity as T
    names = names or [n for n in T.CASES if n not in T.HEAVY]
    floor = {}
    for name in names:
        P.WORST.clear()
        cfg, p32, X, y = T.make_case(name)
        try:
            P.check_backward_stages(cfg, p32, X, y, Fp32OracleEngine(cfg, p32), name=name, heavy=True)
            if name in T.TRAIN_CASES:
                for trained_like in (True, False):
                    cfg, p32, X, y = T.make_case(name, trained_like=trained_like)
                    T.step_check(cfg, Fp32OracleEngine(cfg, p32), p32, None, X, y, name)
            verdict = 'within the bound'
        except AssertionError as e:                  # the float32 oracle itself breaks the criterion on this case: worth knowing
            verdict = 'float32 numpy FAILS the criterion: %s' % str(e)[:160]
        for k, v in P.WORST.items():
            if v > floor.get(k, (0.0, ''))[0]:
                floor[k] = (float(v), name)
        print('%-28s %s; worst: %s' % (name, verdict, ', '.join('%s %.2f' % kv for kv in sorted(P.WORST.items(), key=lambda kv: -kv[1])[:3])),
              flush=True)
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'fp32_floor.json'), 'w') as fh:
        json.dump(dict(sorted(floor.items(), key=lambda kv: -kv[1][0])), fh, indent=1)


if __name__ == '__main__':
    main(sys.argv[1:])
