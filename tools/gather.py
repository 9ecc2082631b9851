import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
print(bench.gather_roofline(torch.device("cuda", 0))[1])
