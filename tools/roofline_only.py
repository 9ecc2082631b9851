"""bench.py's gather roofline alone (the kernel the HBM-gather figure is quoted on), for quick iterations and rocprofv3 passes."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == '__main__':
    torch.cuda.set_device(0)
    r = bench.gather_roofline(torch.device("cuda", 0))[1]
    print(json.dumps(r))
