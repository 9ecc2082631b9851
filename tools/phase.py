"""Debug (make PHASE_TIMERS=1): phase boundaries inside fwd_all_kernel of workgroup 7, 100 MHz clock."""
import ctypes as C, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
from bench import workload_cfg
from cffm_amd import hip, synth
from cffm_amd.engine import HipEngine
cfg, B = workload_cfg('frappe')
eng = HipEngine(cfg, seed=2021)
Xh, yh = synth.batches(cfg.M, cfg.F, B, 8, seed=2021)
X, y = torch.from_numpy(Xh).cuda(), torch.from_numpy(yh).cuda()
for i in range(20):
    eng.train_step(X[i % 8], y[i % 8])
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
hip.load().cffm_debug_phase_times.argtypes = [C.c_void_p]
rows = []
for i in range(10):
    eng.train_step(X[i % 8], y[i % 8]); torch.cuda.synchronize()
    hip.load().cffm_debug_phase_times(buf)
    t = np.array(list(buf), dtype=np.int64)[:8]
    rows.append(np.diff(t) * 10)       # ns
    t2 = np.array(list(buf), dtype=np.int64)[8:15]
    rows2 = globals().setdefault('rows2', []); rows2.append(np.diff(t2) * 10)
print('phases (ns): inner+gather, conv0, conv1, conv2, conv3, head')
print(np.median(np.array(rows), axis=0), 'total', np.median(np.array(rows).sum(axis=1)))
print('head sub-phases (ns): independent loads + first-order term, pooling sweeps, barrier, row sums + s0, t1 + dense32, dense1, (last barrier)')
print(np.median(np.array(rows2), axis=0))
