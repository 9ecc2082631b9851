"""Time cffm_dp_apply alone for the gathered-row counts of N = 1, 2, 4, 8 ranks at the frappe shape (B = 256 per rank)."""
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
from bench import workload_cfg, event_time_ms
from cffm_amd.engine import HipEngine
cfg, B = workload_cfg('frappe')
eng = HipEngine(cfg, seed=2021)
W = 1 + cfg.K + cfg.D + 1
for N in (1, 2, 4, 8):
    n = N * B * cfg.F
    rows = torch.randn(n, W, device='cuda') * 1e-3
    ids = torch.randint(0, cfg.M, (n,), dtype=torch.int32, device='cuda')
    rows[:, 0] = ids.view(torch.float32)
    grad = torch.zeros(int(eng.tl.n) + 4, device='cuda'); grad[int(eng.tl.n)] = 100.0
    ms = event_time_ms(lambda: eng.dp_apply(grad, rows, N * B), 30)
    # the same with per-rank sorted runs (what DataParallelStep hands over): N dp_local blocks
    blocks = []
    for r in range(N):
        Xi = torch.randint(0, cfg.M, (B, cfg.F), dtype=torch.int32, device='cuda')
        yi = torch.ones(B, device='cuda')
        g, blk = eng.dp_local(Xi, yi, B, N * B)
        blocks.append(blk.clone())
    allb = torch.cat(blocks).contiguous()
    ms2 = event_time_ms(lambda: eng.dp_apply(grad, allb, N * B, N), 30)
    print('N=%d rows=%d dp_apply: unsorted rows %.1f us, sorted runs %.1f us' % (N, n, ms * 1e3, ms2 * 1e3))
