#!/usr/bin/env python3
"""`python CFFM.py --dataset frappe ...` - the reference's README commands (README.md:20,24,28) run unchanged;
everything lives in cffm_amd/CFFM.py."""
from cffm_amd.CFFM import CFFM, configure_logging, main, parse_args  # noqa: F401

if __name__ == '__main__':
    main()
