"""cProfile of the host side of the training loop (CFFM.train's inner loop) at the frappe shape."""
import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import workload_cfg
from cffm_amd import synth
from cffm_amd.engine import HipEngine
cfg, B = workload_cfg('frappe')
eng = HipEngine(cfg, seed=2021)
n = 200000
ids = torch.from_numpy(synth.sample_ids(np.random.default_rng(1), cfg.M, cfg.F, n)).cuda().reshape(n, cfg.F)
y = torch.ones(n, device='cuda')
def loop(k):
    for _ in range(k):
        start = np.random.randint(0, n - B)
        eng.train_step(ids[start:start + B], y[start:start + B])
    torch.cuda.synchronize()
loop(200)
import time
t0 = time.time(); loop(2000); dt = time.time() - t0
print('2000 steps: %.1f us/step wall' % (dt / 2000 * 1e6))
pr = cProfile.Profile(); pr.enable(); loop(2000); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
