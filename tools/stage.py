import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from cffm_amd import synth
from cffm_amd.engine import HipEngine
from cffm_amd.spec import CFFMConfig
D = int(sys.argv[1]); B = int(sys.argv[2]); F = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cfg = CFFMConfig(M=5382, F=F, K=32, D=D, activation='selu')
eng = HipEngine(cfg)
X, y = synth.batches(cfg.M, cfg.F, B, 1)
ids, yt = torch.from_numpy(X[0]).cuda(), torch.from_numpy(y[0]).cuda()
eng.train_step(ids, yt)
print(bench.stage_times(eng, ids, yt))
