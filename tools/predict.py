"""predict() throughput at the frappe shape for several block sizes (evaluate() sweeps a split in blocks)."""
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
from bench import workload_cfg, event_time_ms
from cffm_amd import synth
from cffm_amd.engine import HipEngine
cfg, _ = workload_cfg('frappe')
eng = HipEngine(cfg, seed=2021)
for B in (256, 400, 1024, 4096, 16384, 65536):
    X = torch.from_numpy(synth.sample_ids(np.random.default_rng(1), cfg.M, cfg.F, B)).cuda().reshape(B, cfg.F)
    ms = event_time_ms(lambda: eng.predict(X), 20)
    print('B=%6d predict %.1f us -> %.2f M examples/s' % (B, ms * 1e3, B / ms / 1e3))
