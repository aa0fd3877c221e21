"""Print the figures of a bench.py JSON line that matter when comparing two versions."""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(f, "unreadable:", e)
        continue
    print(f"{f}: insert {d['value'] / 1e9:.3f} G k-mers/s ({d['ms_per_step']:.2f} ms)  query {d['query_value'] / 1e9:.3f} G/s")
    for k in ("init_db_value", "init_db_ms"):
        if d.get(k):
            print(f"   {k} {d[k]:.4g}")
    if d.get("single_model"):
        sm = d["single_model"]
        print("   single_model", {k: sm.get(k) for k in ("value", "ms_per_build", "transport", "error")})
    r = d.get("roofline")
    if r:
        print(f"   dominant {r['kernel']}: {r['achieved']:.0f} GB/s alg = {r['frac']:.3f} of 8 TB/s, avg launch {r['avg_launch_us']:.1f} us, traffic {r.get('traffic')}")
    for k, v in (d.get("kernel_classes") or {}).items():
        print(f"   {k:12s} {v['seconds_per_step'] * 1e3:7.2f} ms/step  {v['avg_launch_us']:8.1f} us/launch  {v.get('alg_GBps', 0):7.0f} GB/s alg")
    st = d.get("stats")
    if st:
        print("   ", {k: st[k] for k in ("attempts", "successes", "contended", "finisher_iters")})
