import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """Parity log: the worst |err| / bound ratio seen per compared tensor (tests/test_gpu_parity.py::close), kept under
    gpurun_out/ so that a green run still shows how close each tensor came to its bound."""
    try:
        import json
        mod = sys.modules.get('tests.test_gpu_parity')
        worst = getattr(mod, 'WORST', None)
        if worst:
            out = os.path.join(ROOT, 'gpurun_out')
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, 'parity_worst.json'), 'w') as fh:
                json.dump(dict(sorted(worst.items(), key=lambda kv: -kv[1])), fh, indent=1)
            where = getattr(sys.modules.get('oracle.parity'), 'WORST_AT', {})
            with open(os.path.join(out, 'parity_worst_at.json'), 'w') as fh:
                json.dump({k: where.get(k, '') for k, _ in sorted(worst.items(), key=lambda kv: -kv[1])}, fh, indent=1)
    except Exception:
        pass
