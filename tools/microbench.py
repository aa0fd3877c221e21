"""Random-access ceilings of one MI355X for this path's access pattern (8-byte touches at hashed addresses)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from kmcex_amd import api

res = []
for mode, name in ((0, "gather8"), (1, "atomic_or8"), (2, "byte_store"), (3, "byte_gather"), (4, "store8"), (5, "atomic_or4")):
    for mb in (32, 512, 2048, 8192):
        touches = 1 << 28
        s = api.microbench(mode, mb << 20, touches, 3)
        r = {"op": name, "footprint_MiB": mb, "touches": touches, "seconds": s, "Gtouch_per_s": touches / s / 1e9,
             "GBps_at_32B": touches * 32 / s / 1e9, "GBps_at_64B": touches * 64 / s / 1e9}
        res.append(r)
        print(json.dumps(r), flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
