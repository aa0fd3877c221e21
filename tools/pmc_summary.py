"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into per-kernel HBM traffic.

usage: pmc_summary.py <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/ pmcmicro_FETCH_SIZE/ pmcmicro_WRITE_SIZE/> <out.json> [label] [queries per k_query launch]
Units/corrections (MI355X_MICROARCH.md §HBM): counters are in KiB.  The wide-streaming x2 read correction does not
apply to this pattern, so the absolute scale is calibrated on kmx_microbench kernels with a known touch count
(k_micro_gather: 2^28 random 8-byte loads per dispatch; k_micro_atomic_or: 2^28 random 64-bit atomics).
"""
import collections, csv, glob, json, sys

root, out = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else ""
TOUCHES = 1 << 28


def agg(tag, c):
    fs = glob.glob(f"{root}/{tag}{c}/*/*counter_collection.csv") + glob.glob(f"{root}/{tag}{c}/*counter_collection.csv")
    a = collections.defaultdict(lambda: [0, 0.0])
    if fs:
        for r in csv.DictReader(open(fs[0])):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            a[k][0] += 1
            a[k][1] += float(r["Counter_Value"])
    return a


import os
head = os.environ.get("KMX_GIT_HEAD") or None      # the GPU box has no .git: gpurun -- "KMX_GIT_HEAD=$(git rev-parse --short HEAD) tools/pmc_run.sh <tag>"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                       # source_digest(): bench.py reports `traffic` only for the sources the counters were taken on
res = {"label": label, "head": head, "src_sha": bench.source_digest(), "unit_note": "FETCH_SIZE / WRITE_SIZE in KiB as reported by rocprofv3; bytes = KiB * 1024", "kernels": {}}
f, w = agg("pmc_", "FETCH_SIZE"), agg("pmc_", "WRITE_SIZE")
mf, mw = agg("pmcmicro_", "FETCH_SIZE"), agg("pmcmicro_", "WRITE_SIZE")
cal = {}
if "k_micro_gather" in mf:
    n, v = mf["k_micro_gather"]
    cal["bytes_fetched_per_random_8B_load"] = v * 1024 / n / TOUCHES
if "k_micro_atomic_or" in mw:
    n, v = mw["k_micro_atomic_or"]
    cal["bytes_written_per_random_64bit_atomic"] = v * 1024 / n / TOUCHES
    n2, v2 = mf.get("k_micro_atomic_or", [1, 0.0])
    cal["bytes_fetched_per_random_64bit_atomic"] = v2 * 1024 / n2 / TOUCHES
if "k_micro_gather32" in mf:
    n, v = mf["k_micro_gather32"]
    cal["bytes_fetched_per_random_4B_load"] = v * 1024 / n / TOUCHES
if "k_micro_atomic_or32" in mw:
    n, v = mw["k_micro_atomic_or32"]
    cal["bytes_written_per_random_32bit_atomic"] = v * 1024 / n / TOUCHES
    n2, v2 = mf.get("k_micro_atomic_or32", [1, 0.0])
    cal["bytes_fetched_per_random_32bit_atomic"] = v2 * 1024 / n2 / TOUCHES
if "k_micro_store8" in mw:
    n, v = mw["k_micro_store8"]
    cal["bytes_written_per_random_8B_store"] = v * 1024 / n / TOUCHES
res["calibration"] = cal
if len(sys.argv) > 4:
    res["n_queries"] = int(sys.argv[4])      # queries per k_query launch of the profiled bench.py run (for the counted touches per query)
for k in sorted(set(f) | set(w)):
    if not k.startswith("k_"):
        continue
    nf, vf = f.get(k, [0, 0.0])
    nw, vw = w.get(k, [0, 0.0])
    n = max(nf, nw)
    res["kernels"][k] = {"launches": n, "fetch_bytes_total": vf * 1024, "write_bytes_total": vw * 1024,
                         "hbm_bytes_per_launch": (vf + vw) * 1024 / max(n, 1)}
json.dump(res, open(out, "w"), indent=1)
for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["fetch_bytes_total"] - kv[1]["write_bytes_total"])[:8]:
    print("%-36s launches %5d fetch %8.2f GB write %8.2f GB per-launch %10.2f MB" % (k[:36], v["launches"], v["fetch_bytes_total"] / 1e9, v["write_bytes_total"] / 1e9, v["hbm_bytes_per_launch"] / 1e6))
print(cal)
