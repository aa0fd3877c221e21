"""Print the kernel timeline of one mid-build block from a rocprofv3 kernel trace (csv)."""
import csv, glob, sys
f = (glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
ours = [r for r in rows if r["Kernel_Name"].startswith(("void k_", "k_")) and "k_micro" not in r["Kernel_Name"]]
ours.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(ours[0]["Start_Timestamp"])
idx = [i for i, r in enumerate(ours) if "k_block_init" in r["Kernel_Name"]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
st, en = idx[which], idx[which + 1]
prev_end = None
for r in ours[st:en + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print("%10.1f dur %8.1f gap %6.1f %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"].split("(")[0].replace("void ", "")[:36]))
    prev_end = e
print("block span us", (int(ours[en]["Start_Timestamp"]) - int(ours[st]["Start_Timestamp"])) / 1e3)
