"""cffm_amd: MI355X-native implementation of the CFFM convolutional feature-interaction hot path.

Host side in Python (this package) over a C-ABI HIP library (csrc/ -> libcffm_hip.so, declared in
include/cffm_hip.h).  The package deliberately has no CPU fallback: every compute entry point goes
through the HIP library and raises if it is missing.
"""
__version__ = '0.1.0'
