"""Per-round (t = 0..nb-1) average duration of each insert kernel from a rocprofv3 kernel trace of a one-stream build."""
import csv, glob, sys
from collections import defaultdict
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 5
f = (glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("void k_", "k_"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = defaultdict(lambda: [0, 0.0, 0.0])
t = -1
for r in rows:
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    if nm == "k_block_init": t = -1
    if nm in ("k_round_check_emit", "k_round_commit_check"): t += 1          # the launch a round starts with
    if nm.startswith(("k_round", "k_slow", "k_reorder")) and t >= 0:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = acc[(t % nb, nm)]; a[0] += 1; a[1] += d; a[2] = max(a[2], d)
for (t, nm), (n, d, mx) in sorted(acc.items()):
    print("round %d %-24s n %5d avg %7.1f us max %7.1f total %7.2f ms" % (t, nm, n, d / n, mx, d / 1e3))
if len(sys.argv) > 3:   # sequence of one kernel's durations in round 0, last build
    want = sys.argv[3]
    seq = []
    t = -1
    for r in rows:
        nm = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if nm == "k_block_init": t = -1
        if nm in ("k_round_check_emit", "k_round_commit_check"): t += 1          # the launch a round starts with
        if nm == want and t == 0: seq.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    per = len(seq) // 3 if len(seq) >= 3 else len(seq)
    print(want, "round 0, last build:", " ".join("%.0f" % v for v in seq[-per:]))
