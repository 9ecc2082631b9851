"""Debug (make PHASE_TIMERS=1): start / end of every workgroup of the last launch that records them - bwd_top_kernel at the frappe
shape (512 workgroups: inner-branch role, example role), conv_bwd_pair_kernel at shapes that still use it."""
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
# usage: tools/wg_pair.py [top|pair848|pair768]
which = sys.argv[1] if len(sys.argv) > 1 else 'top'
roles = {'pair848': [('wgrad L1', 0, 256), ('top L3', 256, 272), ('top L2', 272, 336), ('dgrad', 336, 848)],
         'pair768': [('wgrad', 0, 256), ('dgrad', 256, 768)],
         'top': [('inner role', 0, 256), ('key placement', 256, 512), ('example role', 512, 768)]}[which]
n = roles[-1][2]
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
    late = np.nonzero(s > 2000)[0]
    print('  workgroups starting later than 2 us: %d' % late.size, 'first/last index', (late.min(), late.max()) if late.size else None,
          'start p50 %d' % (np.median(s[late]) if late.size else 0))
    first_end = np.sort(e)[:8]
    print('  earliest ends:', first_end)
