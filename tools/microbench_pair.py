"""Do the gather-bound and the atomic-bound kernels hide behind each other?  kmx_microbench mode 20 runs the 4-byte random
gathers and the 32-bit atomic ORs side by side on two streams (own buffers); compare with each alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmcex_amd import api
T = 1 << 28
for mb in (190, 380):
    g = api.microbench(8, mb << 20, T, 3)
    a = api.microbench(5, mb << 20, T, 3)
    p = api.microbench(20, mb << 20, T, 3)
    print(f"{mb:4d} MiB per buffer: gathers alone {g*1e3:7.2f} ms ({T/g/1e9:5.1f} G/s)  atomics alone {a*1e3:7.2f} ms ({T/a/1e9:5.1f} G/s)  "
          f"side by side {p*1e3:7.2f} ms  = {p/(g+a):.2f} of the sum, {p/max(g,a):.2f} of the longer", flush=True)
