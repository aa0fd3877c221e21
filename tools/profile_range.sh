#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_range.sh <tag> [ring|range]
# rocprofv3 --kernel-trace --stats of a multi-GPU partition's engine at one rank (bench.py --partition <p>, nothing else timed):
# kernel stats, the un-profiled ms per build beside the time inside kernels of one build (host-bound orchestration shows as the difference)
set -e
tag=$1; part=${2:-range}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
args="--partition $part --steps 1 --warmup 0 --no-init-db --cpu-sample 0 --no-query-strings --no-roofline --single-model-steps 2"
python "$root/bench.py" $args > "$root/gpurun_out/${tag}_plain_run.json" 2> "$root/gpurun_out/${tag}_plain.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python "$root/bench.py" $args > "$root/gpurun_out/${tag}_run.json" 2> "$root/gpurun_out/${tag}.err"
cd "$root"
f=$(ls "$out"/*/*kernel_stats.csv "$out"/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "gpurun_out/${tag}_kernel_stats.csv"
python - "$tag" "$out" > "gpurun_out/${tag}_summary.txt" <<'PY'
import csv, glob, json, sys
tag, out = sys.argv[1], sys.argv[2]
line = lambda p: json.loads(open(p).read().strip().splitlines()[-1])["single_model"]
plain, prof = line(f"gpurun_out/{tag}_plain_run.json"), line(f"gpurun_out/{tag}_run.json")
f = (glob.glob(out + "/*kernel_trace.csv") + glob.glob(out + "/*/*kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
# the single-model builds: from a k_histogram that is followed by ring/range kernels to the next k_query
idx = [i for i, r in enumerate(rows) if nm(r) == "k_histogram"]
builds = []
for s in idx:
    e = next((i for i in range(s + 1, len(rows)) if nm(rows[i]) in ("k_query", "k_histogram")), len(rows))
    if any(nm(r).startswith(("k_ring_", "k_range_")) for r in rows[s:e]): builds.append((s, e))
s, e = builds[-1]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[s:e]) / 1e6
span = (int(rows[e - 1]["End_Timestamp"]) - int(rows[s]["Start_Timestamp"])) / 1e6
print(f"partition {plain['partition']}, one rank, 1e8 k-mers: {plain['ms_per_build']:.1f} ms per build un-profiled; under rocprofv3 {prof['ms_per_build']:.1f} ms, "
      f"last build {e - s} kernels, span {span:.1f} ms, inside kernels {busy:.1f} ms")
PY
cat "gpurun_out/${tag}_summary.txt"
rm -rf "$out"
