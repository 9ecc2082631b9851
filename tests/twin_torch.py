"""The torch-autograd twin lives with the oracle (oracle/cffm_twin_torch.py) since bench.py's cpu_baseline leg also
times it; tests keep importing it under this name."""
from oracle.cffm_twin_torch import *  # noqa: F401,F403
from oracle.cffm_twin_torch import act, forward, loss, train_step  # noqa: F401
