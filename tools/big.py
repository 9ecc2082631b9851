"""Functional + timing check of the big BASELINE configs on one GPU (not part of the test suite)."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from cffm_amd import synth
from cffm_amd.engine import HipEngine
from cffm_amd.spec import CFFMConfig, init_params
from oracle import cffm_oracle as orc

def run(name, M, F, K, D, B, act, steps):
    cfg = CFFMConfig(M=M, F=F, K=K, D=D, activation=act)
    t0 = time.time()
    rng = np.random.default_rng(0)
    p = init_params(cfg, seed=1)
    print(name, 'init params %.1fs' % (time.time() - t0), flush=True)
    eng = HipEngine(cfg, params=p)
    X, y = synth.batches(M, F, B, 2, seed=3)
    ids, yt = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    out = eng.predict(ids[0]); torch.cuda.synchronize()
    nb = 3
    p64 = {k: np.asarray(v, dtype=np.float64) for k, v in p.items()}
    ref, _ = orc.forward(p64, X[0][:nb], cfg)      # examples are independent in the forward pass
    got = out[:nb].cpu().numpy()
    print(name, 'predict first rows', got, 'oracle', ref, 'max rel err', np.abs(got - ref).max() / max(1e-30, np.abs(ref).max()), flush=True)
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
    losses = []
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(steps):
        losses.append(eng.train_step(ids[i % 2], yt[i % 2]).clone())
    torch.cuda.synchronize(); dt = (time.time() - t0) / steps
    print(name, 'B=%d  %.2f ms/step  %.0f examples/s  losses %s' % (B, dt * 1e3, B / dt, [round(float(l), 4) for l in losses]), flush=True)
    print(name, 'ws GB %.2f' % (eng.workspace(B)[1].bytes / 1e9), flush=True)

which = sys.argv[1]
if which == 'bookx':
    run('book-crossing', 226336, 6, 32, 32, 512, 'relu', 50)
elif which == 'mltag':
    run('ml-tag', 90445, 3, 32, 32, 1024, 'elu', 50)
elif which == 'syn1m':
    run('syn-1M', 1000000, 32, 64, 64, int(sys.argv[2]) if len(sys.argv) > 2 else 8192, 'relu', 3)
