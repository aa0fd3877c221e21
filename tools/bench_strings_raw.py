"""kmer_to_occ over ASCII strings of which one in a thousand holds an N: the whole batch takes the byte-string kernel."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kmcex_amd import KModel, synth, api
n = 20_000_000
km, cnt = synth.make_stream(n, 31, 1, 1023)
m = KModel(1, 1023, 7, 5); m.build_packed(31, km, cnt)
asc = np.ascontiguousarray(synth.to_ascii(km, 31))
asc[::1000, 5] = ord('N')
out = np.zeros(len(cnt), dtype=np.int32)
L = api.load_library()
for rep in range(3):
    t = time.time(); rc = L.kmx_query_ascii(m.h, asc.ctypes.data_as(C.c_char_p), 31, 31, len(cnt), out.ctypes.data); dt = time.time() - t
    assert rc == 0
    print(f"byte-string kernel path: {len(cnt)/dt/1e6:.1f} M strings/s ({dt*1e3:.0f} ms)", flush=True)
