import sys, torch
sys.path.insert(0, '.')
import bench
print(bench.gather_roofline(torch.device('cuda', 0)))
