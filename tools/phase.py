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

# ---- backward launches (workgroup 7 of each role) ---------------------------------------------------------------------
lib = hip.load()
bb, ub = (C.c_ulonglong * 48)(), (C.c_ulonglong * 48)()
lib.cffm_debug_bwd_times.argtypes = [C.c_void_p]; lib.cffm_debug_upd_times.argtypes = [C.c_void_p]
R = []
for i in range(10):
    eng.train_step(X[i % 8], y[i % 8]); torch.cuda.synchronize()
    lib.cffm_debug_bwd_times(bb); lib.cffm_debug_upd_times(ub)
    t = np.array(list(bb), dtype=np.int64); u = np.array(list(ub), dtype=np.int64)
    t[26:30] = u[26:30]
    R.append(t)
R = np.array(R)
d = lambda a, b: float(np.median(R[:, b] - R[:, a]) * 10)
print('bwd_top example role (ns): begin %d, loss %d, head example %d, head end %d, top wgrad %d, top dgrad %d, next wgrad %d, next dgrad %d | total %d'
      % (d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(5, 6), d(6, 7), d(7, 8), d(0, 8)))
print('bwd_top inner role (ns): loss %d, keys %d, inner bwd %d | total %d; starts %d after the example role' % (d(36, 37), d(37, 38), d(38, 39), d(36, 39), d(0, 36)))
fused01 = bool(R[:, 40].any())          # conv01_bwd_kernel ran (the frappe command): the pair launch and conv0_fact_bwd_kernel did not
if not fused01:
    print('pair (ns): dgrad role %d, wgrad role %d; starts %d after bwd_top start' % (d(21, 22), d(23, 24), d(0, 21)))
    print('pair offsets from the dgrad role start (ns): wgrad role starts %d, ends %d; deferred top role starts %d, ends %d; dgrad ends %d' % (d(21, 23), d(21, 24), d(21, 30), d(21, 31), d(21, 22)))
    print('pair wgrad role, workgroup 7 (ns): loads issued + dC staged %d, MFMAs %d, slab stores issued %d' % (d(32, 33), d(33, 34), d(34, 24)))
print('%s (ns): W+E load %d, A %d, B %d, C %d, D %d, E %d, rs+sync+reduce %d, F %d, tail %d | total %d'
      % ('layer-0 phases of conv01_bwd' if fused01 else 'conv0_fact_bwd', d(12, 13), d(13, 14), d(14, 15), d(15, 16), d(16, 17), d(17, 18), d(18, 19), d(19, 20), d(20, 25), d(12, 25)))
if fused01:
    print('conv01_bwd (ns): input gradient of layer 1 %d, weight gradients of layers 1-3 %d, layer 0 %d | total %d; starts %d after bwd_top start'
          % (d(40, 41), d(41, 42), d(42, 25), d(40, 25), d(0, 40)))
print('update_all (ns): reduce role %d, sparse role %d; starts %d after conv0_fact_bwd start; step span bwd_top start -> update end %d'
      % (d(26, 27), d(28, 29), d(12, 26), max(d(0, 27), d(0, 29))))
