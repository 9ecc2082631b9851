"""`import LoadData as DATA` as in the reference (CFFM.py:9); the loader lives in cffm_amd/LoadData.py."""
from cffm_amd.LoadData import LoadData  # noqa: F401
