#!/bin/bash
# usage (GPU box, repo root): tools/pmc_per_dispatch.sh <tag> <kernel substring> "<counters>"  -> gpurun_out/<tag>_dispatches.txt
# one --pmc pass over a one-step bench.py run; one line per dispatch of the kernels whose name contains the substring
tag=$1; kern=$2; C=$3
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmcd_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $C --output-format csv -d "$out/pass" -- python "$root/bench.py" --steps 1 --warmup 0 --no-roofline --no-init-db --cpu-sample 0 --no-single-model > "$out/bench.json" 2> "$out/bench.err" || { echo "pass failed"; tail -5 "$out/bench.err"; }
cd "$root"
python - "$out" "$kern" > "gpurun_out/${tag}_dispatches.txt" <<'PY'
import collections, csv, glob, sys
root, kern = sys.argv[1], sys.argv[2]
rows = collections.OrderedDict()
for f in glob.glob(f"{root}/pass/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern not in r["Kernel_Name"]:
            continue
        d = rows.setdefault(int(r["Dispatch_Id"]), {"grid": r.get("Grid_Size"), "wg": r.get("Workgroup_Size")})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
for n, (i, d) in enumerate(sorted(rows.items())):
    print(n, i, " ".join(f"{k}={v:.4g}" if isinstance(v, float) else f"{k}={v}" for k, v in d.items()))
PY
rm -rf "$out"/pass
head -40 "gpurun_out/${tag}_dispatches.txt"
