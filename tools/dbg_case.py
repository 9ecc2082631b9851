import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_parity as T
from oracle import cffm_oracle as orc
name = sys.argv[1] if len(sys.argv) > 1 else 'f13-k64-d64-b9-gelu'
cfg, p32, X, y = T.make_case(name)
eng = T.engine_for(cfg, p32)
B = X.shape[0]; p64 = T.to64(p32)
out_ref, c = orc.forward(p64, X, cfg)
L, dout = orc.loss_and_grad(out_ref, y.astype(np.float64), cfg, p64)
ids = torch.from_numpy(X).cuda(); yt = torch.from_numpy(y).cuda()
eng.forward(ids, yt); torch.cuda.synchronize()
Pp = eng.tl.Pp
nk = T.adopt_device_kinks(cfg, eng, B, c)
g = orc.backward(p64, c, dout, cfg)
eng.backward(yt, B); torch.cuda.synchronize()
for l in range(cfg.live_layers - 1, -1, -1):
    S = cfg.D >> (l + 1)
    got = eng.ws_tensor(B, 'dC', (B, S, S, Pp), index=l).cpu().numpy()[..., :cfg.P]
    ref = g['_dC'][l]
    err = np.abs(got - ref); rms = np.sqrt((ref**2).mean())
    idx = np.unravel_index(np.argsort(err.ravel())[-4:], err.shape)
    print('layer', l, 'rms', rms, 'max err', err.max())
    for k in range(4):
        i = tuple(a[k] for a in idx)
        print('   ', i, 'got', got[i], 'ref', ref[i], 'z', c['zs'][l][i], 'zmax', np.abs(c['zs'][l]).max())
print('dout', dout, 'out', out_ref, 'y', y)
dt1 = eng.ws_tensor(B, 'dt1', (B, 2*cfg.D-2)).cpu().numpy()
print('dt1 err', np.abs(dt1 - g['_dt1']).max(), np.abs(g['_dt1']).max())
