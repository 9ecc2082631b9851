"""Times cffm_outer_conv0_fwd (the tiled layer-0 forward) alone at the stress shape: HIP events, 3 launches."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cffm_amd import hip  # noqa: E402
from cffm_amd.engine import HipEngine  # noqa: E402

cfg, B = bench.workload_cfg('syn1m')
B = int(os.environ.get('B', B))
eng = HipEngine(cfg, params='device', seed=2021)
lib = hip.load()
buf, wl = eng.workspace(B)
buf[int(wl.Eo):int(wl.dEi)].view(torch.float32).normal_(0, 0.01) if False else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
for which, fn in (('conv0_fwd', lambda: lib.cffm_outer_conv0_fwd(C.byref(eng.shape), P(eng.theta), P(buf), B, st)),):
    print(which, round(bench.event_time_ms(lambda: hip.check(fn()), 2, warm=1), 2), 'ms', flush=True)
