"""One-off large-scale parity run: positions beyond 2^32 bits per array (N ~ 1.5*10^9 31-mers), GPU vs CPU oracle.
--oracle compares with the CPU oracle; with --golden <file> the oracle's statistics and array digests are written as the
pinned reference of tests/test_gpu_fullsize.py::test_hc14_scale_properties (only after the comparison has passed)."""
import hashlib, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib as O
from kmcex_amd import KModel, synth_torch

def beat():
    t0 = time.time()
    while True:
        time.sleep(30); print(f"[heartbeat] {time.time()-t0:.0f}s", flush=True)
threading.Thread(target=beat, daemon=True).start()

n_draw = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_500_000_000
with_oracle = "--oracle" in sys.argv
light = "--light" in sys.argv            # build + query + invariants on the device only (no array downloads)
K, CI, CS, NH, NB = 31, 1, 1023, 7, 5
dev = torch.device("cuda", 0)
t = time.time()
km, cnt = synth_torch.make_stream(n_draw, K, CI, CS, dev)
torch.cuda.synchronize(); n = km.numel()
torch.cuda.empty_cache()                      # the generator's temporaries would otherwise stay in torch's pool
print(f"device memory in use before build: {(torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 2**30:.1f} GiB of {torch.cuda.mem_get_info()[1] / 2**30:.0f}", flush=True)
print(f"stream: {n} k-mers in {time.time()-t:.1f}s", flush=True)
m = KModel(CI, CS, NH, NB)
t = time.time(); m.build_dev(K, km.data_ptr(), cnt.data_ptr(), n); dt = time.time() - t
if "--rebuild" in sys.argv:               # a second build on the warm model (allocations kept), optionally with per-class kernel times
    if "--profile" in sys.argv: m.set_profile(True); m.kernel_times(True)
    t = time.time(); m.build_dev(K, km.data_ptr(), cnt.data_ptr(), n); dt2 = time.time() - t
    print(f"warm rebuild {dt2:.2f}s = {n/dt2/1e6:.1f} M k-mers/s", flush=True)
    if "--profile" in sys.argv:
        for kname, v in m.kernel_times(True).items(): print(f"  {kname:14s} {v['seconds']*1e3:10.1f} ms {v['launches']} launches", flush=True)
        m.set_profile(False)
st = m.stats()
print(f"GPU build {dt:.2f}s = {n/dt/1e6:.1f} M k-mers/s; L = {st.km_byte_size*8} bits per array (2^32 = {2**32}); attempts {st.attempts} successes {st.successes} rest {st.rest_entries} contended {st.contended} fin_iters {st.finisher_iters}", flush=True)
assert st.successes + st.rest_entries >= st.n_km and st.successes + st.rest_entries - st.n_km < NB
out = torch.empty(200_000_000, dtype=torch.int32, device=dev)
t = time.time(); m.kmer_to_occ_dev(km.data_ptr(), 200_000_000, out.data_ptr()); torch.cuda.synchronize(); dq = time.time() - t
print(f"query 2e8 present k-mers: {2e8/dq/1e6:.1f} M/s, nonzero {(out != 0).float().mean().item():.5f}", flush=True)
print(f"device memory in use after build: {(torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 2**30:.1f} GiB", flush=True)
if light:
    sys.exit(0)
def sha(a): return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
gh = {}
for a in range(NB):
    tag, val = m.download("tag", a), m.download("value", a)
    assert not (val & ~tag).any()
    assert not m.download("claims", a).any()
    # how far up the array do set bits reach?  (positions above 2^32 must be in use)
    nz = np.flatnonzero(tag)
    gh[a] = (sha(tag), sha(val), int(nz[-1]) * 8)
    print(f"array {a}: tag sha {gh[a][0][:16]} value sha {gh[a][1][:16]} highest set bit near position {gh[a][2]}", flush=True)
    assert gh[a][2] > 2**32 or st.km_byte_size * 8 <= 2**32
gkb = sha(m.download("km_back"))
if with_oracle:
    hk, hc = km.cpu().numpy().view(np.uint64), cnt.cpu().numpy().view(np.uint32)
    o = O.OracleModel(CI, CS, NH, NB)
    t = time.time(); o.build(K, hk, hc); print(f"oracle build {time.time()-t:.0f}s", flush=True)
    so = o.stats()
    assert (st.attempts, st.successes, st.rest_entries) == (so.attempts, so.successes, so.rest_entries), (so.attempts, so.successes, so.rest_entries)
    for a in range(NB):
        assert gh[a][0] == sha(o.array_bytes("tag", a)) and gh[a][1] == sha(o.array_bytes("value", a)), a
    assert gkb == sha(o.array_bytes("km_back"))
    q = hk[::50]
    got = m.kmer_to_occ_packed(q)
    assert np.array_equal(got, o.query_packed(K, q, threads=64))
    print("BIT-EXACT vs oracle at this scale (arrays, km_back, stats, sampled queries)", flush=True)
    if "--golden" in sys.argv:
        d = {f"tag{a}": sha(o.array_bytes("tag", a)) for a in range(NB)}
        d.update({f"value{a}": sha(o.array_bytes("value", a)) for a in range(NB)})
        d.update(km_back=sha(o.array_bytes("km_back")), bf0=sha(o.array_bytes("bf", 0)), bf_back0=sha(o.array_bytes("bf_back", 0)))
        assert d["bf0"] == sha(m.download("bf", 0)) and d["bf_back0"] == sha(m.download("bf_back", 0))
        g = {"what": "CPU oracle (oracle/kmx_oracle.c, pinned to the compiled reference) on synth_torch.make_stream(n_draws, 31, 1, 1023), nh=7 nb=5; the GPU build of the same run was identical",
             "n_draws": n_draw, "n_kmers": n, "k": K, "ci": CI, "cs": CS, "nh": NH, "nb": NB,
             "stats": {"n_km": int(so.n_km), "attempts": int(so.attempts), "successes": int(so.successes), "rest_entries": int(so.rest_entries)}, "sha256": d}
        json.dump(g, open(sys.argv[sys.argv.index("--golden") + 1], "w"), indent=1)
        print("golden written", flush=True)
