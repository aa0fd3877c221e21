"""Atomic flavours at random addresses: agent scope (what the insert uses), workgroup scope, returning."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmcex_amd import api
for mode, name in ((1, "atomic_or8 agent-scope"), (6, "atomic_or8 workgroup-scope"), (7, "atomic_or8 returning")):
    for mb in (32, 512, 4096):
        s = api.microbench(mode, mb << 20, 1 << 28, 3)
        print(f"{name:28s} {mb:5d} MiB {(1 << 28) / s / 1e9:6.1f} G/s", flush=True)
