"""bench.py's measured peaks alone (float4 copy, read-only stream, dependency-free fp32 and bf16 MFMA loops)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == '__main__':
    torch.cuda.set_device(0)
    print(json.dumps(bench.measured_peaks(torch.device('cuda', 0))))
