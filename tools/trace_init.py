"""Kernel time and idle time of the last KModel::init(db) of a rocprofv3 --kernel-trace csv (tools/profile_init.sh): from its first
k_kmc_decode to its last kernel -- what the device does while the host feeds it."""
import collections, csv, glob, sys
f = (glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("void k_", "k_"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
# a call of init = a run of kernels that starts with k_kmc_decode after a long pause
starts = [i for i, r in enumerate(rows) if name(r) == "k_kmc_decode" and (i == 0 or int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 5_000_000)]
st = starts[-1]
b = rows[st:]
span = (int(b[-1]["End_Timestamp"]) - int(b[0]["Start_Timestamp"])) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in b) / 1e6
print(f"last init(db): {len(b)} kernels from the first decode to the last kernel: span {span:.2f} ms, inside kernels {busy:.2f} ms, idle {span - busy:.2f} ms")
gaps = collections.defaultdict(lambda: [0, 0.0])
prev_end = None
for i, r in enumerate(b):
    s = int(r["Start_Timestamp"])
    if prev_end is not None and s > prev_end:
        g = (s - prev_end) / 1e3
        if g > 20:
            gaps[name(b[i - 1]) + " -> " + name(r)][0] += 1
            gaps[name(b[i - 1]) + " -> " + name(r)][1] += g
    prev_end = max(prev_end or 0, int(r["End_Timestamp"]))
print("idle gaps above 20 us, by the kernels around them:")
for k, (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {k:58s} {n:4d} x {t / n:8.1f} us = {t / 1e3:6.3f} ms")
per = collections.defaultdict(lambda: [0, 0.0])
for r in b:
    per[name(r)][0] += 1; per[name(r)][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("kernel time of that call:")
for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {k:32s} {n:5d} x {t / n:8.2f} us = {t / 1e3:7.3f} ms")

# timeline around one batch boundary (the 4th idle gap above 200 us): kernels and memory copies
import os
mc = (glob.glob(sys.argv[1] + "/*memory_copy_trace.csv") + glob.glob(sys.argv[1] + "/*/*memory_copy_trace.csv"))
big = [i for i in range(1, len(b)) if int(b[i]["Start_Timestamp"]) - max(int(r["End_Timestamp"]) for r in b[max(0, i - 3):i]) > 200_000]
if big and mc:
    i = big[min(3, len(big) - 1)]
    g0, g1 = int(b[i - 1]["End_Timestamp"]), int(b[i]["Start_Timestamp"])
    t0 = g0 - 7_000_000
    print(f"\ntimeline around one batch boundary (times in ms relative to the start of the idle gap; gap = {(g1 - g0) / 1e3:.0f} us):")
    ev = []
    for r in b:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e >= t0 and s <= g1 + 1_500_000 and name(r) in ("k_kmc_decode", "k_classify_count", "k_classify_scatter", "k_kmback_emit", "k_block_init", "k_rest_append"):
            ev.append((s, e, "kernel " + name(r)))
    rows_mc = list(csv.DictReader(open(mc[0])))
    print("  (memory copy columns:", ", ".join(rows_mc[0].keys()) if rows_mc else "none", ")")
    for r in rows_mc:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e >= t0 and s <= g1 + 1_500_000:
            ev.append((s, e, "copy   " + " ".join(str(r.get(k, "")) for k in ("Direction", "Bytes", "Stream_Id") if k in r)))
    for s, e, what in sorted(ev):
        if e - s > 20_000 or what.startswith("copy"):
            print(f"  {(s - g0) / 1e6:8.3f} .. {(e - g0) / 1e6:8.3f} ms  {what}")
