"""kmer_to_occ over ASCII k-mer strings through the C ABI (kmx_query_ascii): host strings -> answers on the host."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kmcex_amd import KModel, synth, api
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
km, cnt = synth.make_stream(n, 31, 1, 1023)
m = KModel(1, 1023, 7, 5); m.build_packed(31, km, cnt)
asc = np.ascontiguousarray(synth.to_ascii(km, 31))                # uint8[n, 31]
out = np.zeros(len(cnt), dtype=np.int32)
L = api.load_library()
for rep in range(3):
    t = time.time()
    rc = L.kmx_query_ascii(m.h, asc.ctypes.data_as(C.c_char_p), 31, 31, len(cnt), out.ctypes.data)
    dt = time.time() - t
    assert rc == 0
    print(f"kmx_query_ascii: {len(cnt)/dt/1e6:.1f} M k-mers/s ({dt*1e3:.0f} ms for {len(cnt)} strings), nonzero {np.count_nonzero(out)/len(out):.4f}", flush=True)
t = time.time(); o2 = m.kmer_to_occ_packed(km); dt = time.time() - t
print(f"kmx_query_packed (host packed in, host out): {len(cnt)/dt/1e6:.1f} M k-mers/s")
assert np.array_equal(out, o2)
