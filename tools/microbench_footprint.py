"""Random 4-byte gathers and 32-bit atomic ORs against the footprint (kmx_microbench modes 8 and 5): where does the
random-access ceiling come from -- caches, translation, or the memory itself?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmcex_amd import api
T = 1 << 28
for mb in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 4096, 16384):
    g = api.microbench(8, mb << 20, T, 3)
    a = api.microbench(5, mb << 20, T, 3)
    print(f"{mb:6d} MiB   gathers {T/g/1e9:7.1f} G/s   atomic ORs {T/a/1e9:7.1f} G/s", flush=True)
