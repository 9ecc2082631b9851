"""Debug build only (make TILE_DBG=1): time the tiled layer-0 forward at the stress shape with phases switched off
(CFFM_DBG bits: 1 no stores, 2 no W loads, 4 no step-1 MFMAs, 8 no step 2, 16 no E loads) to see where its time goes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cffm_amd import hip, synth
from cffm_amd.engine import HipEngine
from cffm_amd.spec import CFFMConfig
cfg = CFFMConfig(M=1000000, F=32, K=64, D=64, activation='relu')
B = 8192
eng = HipEngine(cfg, params='device', seed=1)
X, y = synth.batches(cfg.M, cfg.F, B, 1, seed=3)
ids, yt = torch.from_numpy(X[0]).cuda(), torch.from_numpy(y[0]).cuda()
eng.forward(ids, yt)
torch.cuda.synchronize()
lib = hip.load()
buf, _ = eng.workspace(B)
for bits in (0, 1, 2, 4, 8, 16, 1 | 8, 2 | 4, 1 | 2 | 4 | 8, 31):
    os.environ['CFFM_DBG'] = str(bits)
    for _ in range(2):
        lib.cffm_outer_conv0_fwd(C.byref(eng.shape), C.c_void_p(eng.theta.data_ptr()), C.c_void_p(buf.data_ptr()), B, None)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        lib.cffm_outer_conv0_fwd(C.byref(eng.shape), C.c_void_p(eng.theta.data_ptr()), C.c_void_p(buf.data_ptr()), B, None)
    b.record(); torch.cuda.synchronize()
    print('CFFM_DBG=%2d  %.2f ms' % (bits, a.elapsed_time(b) / 3), flush=True)
