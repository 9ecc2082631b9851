"""Debug (make PHASE_TIMERS=1): start / end of every workgroup of conv_bwd_pair_kernel at the frappe shape, 100 MHz clock."""
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
buf = (C.c_ulonglong * 2048)()
lib = hip.load()
lib.cffm_debug_wg_times.argtypes = [C.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 848
roles = [('top L3', 0, 16), ('top L2', 16, 80), ('dgrad', 80, 592), ('wgrad L1', 592, 848)] if n == 848 else [('dgrad', 0, 512), ('wgrad', 512, 768)]
for i in range(3):
    eng.train_step(X[i % 8], y[i % 8]); torch.cuda.synchronize()
    lib.cffm_debug_wg_times(buf)
    t = np.array(list(buf), dtype=np.int64)[:2 * n].reshape(n, 2) * 10
    t0 = t[:, 0].min()
    s, e = t[:, 0] - t0, t[:, 1] - t0
    print('all: start max %d, end max %d' % (s.max(), e.max()))
    for name, lo, hi in roles:
        ss, ee = s[lo:hi], e[lo:hi]
        print('  %-9s start p50 %5d max %5d | dur p50 %5d max %5d | end p50 %5d max %5d' % (name, np.median(ss), ss.max(), np.median(ee - ss), (ee - ss).max(), np.median(ee), ee.max()))
