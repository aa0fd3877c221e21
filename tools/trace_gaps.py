"""Where the time of ONE build goes between the kernels: from a rocprofv3 --kernel-trace csv, the last build of the run
(k_histogram ... the last kernel before the next k_histogram / k_query): span, time inside kernels, and the idle gaps between
consecutive kernels by size class and by the kernel that precedes them.  usage: trace_gaps.py <rocprof output dir>"""
import collections, csv, glob, sys
f = (glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("void k_", "k_")) and "k_micro" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
starts = [i for i, r in enumerate(rows) if name(r) == "k_histogram"]
ends = [next((i for i in range(st + 1, len(rows)) if name(rows[i]) in ("k_query", "k_histogram")), len(rows)) for st in starts]
# the last build made of the PRODUCT's kernels (bench.py's last build runs the fused launches' accounting variant, <..., true>)
pick = [j for j in range(len(starts)) if not any(", true>" in r["Kernel_Name"] for r in rows[starts[j]:ends[j]])]
st, en = (starts[pick[-1]], ends[pick[-1]]) if pick else (starts[-1], ends[-1])
b = rows[st:en]
span = (int(b[-1]["End_Timestamp"]) - int(b[0]["Start_Timestamp"])) / 1e3
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in b) / 1e3
classes = [(0, 1.5), (1.5, 2.5), (2.5, 4), (4, 8), (8, 20), (20, 100), (100, 1e9)]
hist = collections.OrderedDict((c, [0, 0.0]) for c in classes)
by_prev = collections.defaultdict(lambda: [0, 0.0])
prev_end = None
for i, r in enumerate(b):
    s = int(r["Start_Timestamp"])
    if prev_end is not None:
        g = (s - prev_end) / 1e3
        for c in classes:
            if c[0] <= g < c[1]:
                hist[c][0] += 1; hist[c][1] += g
        by_prev[name(b[i - 1]) + " -> " + name(r)][0] += 1
        by_prev[name(b[i - 1]) + " -> " + name(r)][1] += g
    prev_end = max(prev_end or 0, int(r["End_Timestamp"]))
print(f"last build: {len(b)} kernels, span {span / 1e3:.2f} ms, inside kernels {busy / 1e3:.2f} ms, idle between kernels {(span - busy) / 1e3:.2f} ms")
for c, (n, t) in hist.items():
    print(f"  gaps of {c[0]:>5}..{c[1] if c[1] < 1e8 else 'inf':>5} us: {n:5d}  total {t / 1e3:7.3f} ms")
print("largest contributors (kernel -> next kernel):")
for k, (n, t) in sorted(by_prev.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {k:58s} {n:5d} x {t / n:6.2f} us = {t / 1e3:6.3f} ms")
per = collections.defaultdict(lambda: [0, 0.0])
for r in b:
    per[name(r)][0] += 1; per[name(r)][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("kernel time of that build:")
for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:32s} {n:5d} x {t / n:8.2f} us = {t / 1e3:7.3f} ms")
