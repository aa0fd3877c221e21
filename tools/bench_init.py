"""End-to-end KModel::init(db) from a KMC database on disk (host listing -> pinned H2D -> GPU insert)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from kmcex_amd import KModel, api, kmcdb, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
km, cnt = synth.make_stream(n, 31, 1, 1023)
with tempfile.TemporaryDirectory() as d:
    db = os.path.join(d, "db")
    kmcdb.write_kmc1(db, km, cnt, 31, 1, 1023)
    t = time.time(); k, total, okm, ocnt = api.kmc_list(db); t_list = time.time() - t
    print(f"listing only: {len(ocnt)/t_list/1e6:.1f} M k-mers/s")
    m = KModel(1, 1023, 7, 5)
    for rep in range(3):
        t = time.time(); m.init(db); dt = time.time() - t
        print(f"init(db) rep {rep}: {len(cnt)/dt/1e6:.1f} M k-mers/s end to end ({dt*1e3:.1f} ms, two passes over the file)")
    m2 = KModel(1, 1023, 7, 5); m2.build_packed(31, km, cnt)
    assert all(np.array_equal(m.download("tag", a), m2.download("tag", a)) for a in range(5))
    print("same arrays as the resident build")
