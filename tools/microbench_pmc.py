"""The three calibration kernels of tools/pmc_summary.py (2^28 random touches per dispatch over a footprint like the
coupled arrays'): 8-byte loads, 64-bit atomic ORs, 8-byte stores."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmcex_amd import api

for mode in (0, 1, 4, 8, 5):
    s = api.microbench(mode, 760 << 20, 1 << 28, 2)
    print(mode, (1 << 28) / s / 1e9, "G touches/s", flush=True)
