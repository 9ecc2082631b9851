"""Sparse Adagrad (sorted duplicates-first table update) on the stress shape of BASELINE.json configs[3]: F32 K=D=64 M=1M
B=8192 uniform ids.  Algorithmic traffic per distinct row: (K+D+1)*4*5 bytes (read summed grad; read+write row; read+write
accumulator) - SURVEY 8d."""
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
from bench import event_time_ms
from cffm_amd import synth
from cffm_amd.engine import HipEngine
from cffm_amd.spec import CFFMConfig
M, F, K, D, B = 1000000, 32, 64, 64, 8192
cfg = CFFMConfig(M=M, F=F, K=K, D=D, activation='relu')
eng = HipEngine(cfg, seed=1)
ids = torch.from_numpy(synth.sample_ids(np.random.default_rng(2021), M, F, B)).cuda().reshape(B, F)
n = B * F
dEi = torch.randn(n, K, device='cuda') * 1e-3
dEo = torch.randn(n, D, device='cuda') * 1e-3
dfb = torch.randn(n, device='cuda') * 1e-3
uniq = int(torch.unique(ids).numel())
ms = event_time_ms(lambda: eng.apply_sparse(ids, dEi, dEo, dfb, B), 20)
alg = uniq * (K + D + 1) * 4 * 5
print('rows %d distinct %d: sort + update %.1f us, algorithmic %.1f MB -> %.0f GB/s (%.3f of 8 TB/s)' % (n, uniq, ms * 1e3, alg / 1e6, alg / ms / 1e6, alg / ms / 1e6 / 8000))
