"""4-byte random gathers, variants of the load instruction (kmx_microbench modes 8-11), two footprints."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmcex_amd import api
for mb in (380, 3800):
    for mode, name in ((8, "plain"), (9, "nontemporal"), (10, "sc1 (agent-scope relaxed)"), (11, "plain, 16 in flight per lane"), (5, "32-bit atomic OR"), (0, "8-byte plain")):
        s = api.microbench(mode, mb << 20, 1 << 28, 3)
        print(f"{mb:5d} MiB  {name:32s} {(1 << 28) / s / 1e9:7.2f} G touches/s", flush=True)
    s = api.microbench(12, mb << 20, 1 << 28, 3)
    print(f"{mb:5d} MiB  {'32-bit atomic OR, 3/8 of the lanes':32s} {(1 << 28) * 0.375 / s / 1e9:7.2f} G touches/s", flush=True)
