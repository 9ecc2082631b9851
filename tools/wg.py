"""Debug (make PHASE_TIMERS=1): start / end of every workgroup of fwd_all_kernel, 100 MHz clock."""
import ctypes as C, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
from bench import workload_cfg
from cffm_amd import synth
from cffm_amd.engine import HipEngine
cfg, B = workload_cfg('frappe')
eng = HipEngine(cfg, seed=2021)
Xh, yh = synth.batches(cfg.M, cfg.F, B, 8, seed=2021)
X, y = torch.from_numpy(Xh).cuda(), torch.from_numpy(yh).cuda()
for i in range(20):
    eng.train_step(X[i % 8], y[i % 8])
torch.cuda.synchronize()
buf = (C.c_ulonglong * 2048)()
hip.load().cffm_debug_wg_times.argtypes = [C.c_void_p]
for i in range(3):
    eng.train_step(X[i % 8], y[i % 8]); torch.cuda.synchronize()
    hip.load().cffm_debug_wg_times(buf)
    t = np.array(list(buf), dtype=np.int64)[:512].reshape(256, 2) * 10
    s, e = t[:, 0] - t[:, 0].min(), t[:, 1] - t[:, 0].min()
    print('start spread %d ns (p50 %d), end min %d max %d, dur p50 %d min %d max %d' % (s.max(), np.median(s), e.min(), e.max(), np.median(e - s), (e - s).min(), (e - s).max()))
    order = np.argsort(e)
    print(' slowest WGs', order[-5:], (e - s)[order[-5:]], ' by xcd mean dur', [(int((e - s)[x::8].mean())) for x in range(8)])
    pbuf = (C.c_ulonglong * 16)()
    hip.load().cffm_debug_phase_times.argtypes = [C.c_void_p]
    hip.load().cffm_debug_phase_times(pbuf)
    pt = np.array(list(pbuf), dtype=np.int64)[:8] * 10
    print(' WG7: start->mark0 (rank_keys) %d ns; phases %s; last mark -> end %d ns' % (pt[0] - t[7, 0], np.diff(pt[:7]), t[7, 1] - pt[6]))
