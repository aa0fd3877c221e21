"""Concurrency summary of the kernels of the LAST build in a rocprofv3 kernel trace (csv): busy union, average number
of kernels in flight, per-queue busy time, and a slice of the timeline with queue ids."""
import csv, glob, sys
from collections import defaultdict
f = (glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
ours = [r for r in rows if r["Kernel_Name"].startswith(("void k_", "k_"))]
ours.sort(key=lambda r: int(r["Start_Timestamp"]))
qcol = "Stream_Id" if "Stream_Id" in ours[0] else "Queue_Id"
# last build = from the last k_classify_count that follows a gap > 5 ms
starts = [i for i in range(1, len(ours)) if int(ours[i]["Start_Timestamp"]) - int(ours[i - 1]["End_Timestamp"]) > 5_000_000]
lo = starts[-1] if starts else 0
seg = ours[lo:]
t0 = int(seg[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seg)
ev = []
for r in seg:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
busy = 0; depth = 0; last = t0; wsum = 0
for t, d in ev:
    if depth > 0: busy += t - last
    wsum += depth * (t - last); last = t; depth += d
print("kernels %d span %.2f ms busy-union %.2f ms sum-dur %.2f ms avg-in-flight(while busy) %.2f" % (len(seg), (t1 - t0) / 1e6, busy / 1e6, wsum / 1e6, wsum / max(busy, 1)))
perq = defaultdict(lambda: [0, 0])
for r in seg:
    perq[r[qcol]][0] += 1; perq[r[qcol]][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for q, (n, d) in sorted(perq.items(), key=lambda x: -x[1][1]):
    print("  %s=%s kernels %d busy %.2f ms" % (qcol, q, n, d / 1e6))
byname = defaultdict(lambda: [0, 0])
for r in seg:
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "")[:30]
    byname[nm][0] += 1; byname[nm][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for nm, (n, d) in sorted(byname.items(), key=lambda x: -x[1][1]):
    print("  %-30s n %5d total %.2f ms avg %.1f us" % (nm, n, d / 1e6, d / n / 1e3))
mid = len(seg) // 2
for r in seg[mid:mid + int(sys.argv[2]) if len(sys.argv) > 2 else mid + 60]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%10.1f dur %7.1f q %s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r[qcol], r["Kernel_Name"].split("(")[0].replace("void ", "")[:30]))
