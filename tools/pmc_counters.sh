#!/bin/bash
# usage (GPU box, repo root): tools/pmc_counters.sh <tag> "<counters of pass 1>" "<counters of pass 2>" ...
# one rocprofv3 --pmc pass per argument over a short bench.py run; per-kernel averages -> gpurun_out/<tag>_counters.txt
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmcc_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d "$out/pass_$i" -- python "$root/bench.py" --steps 1 --warmup 0 --no-roofline --no-init-db --cpu-sample 0 --no-single-model > "$out/bench_$i.json" 2> "$out/bench_$i.err" || { echo "pass $i ($C) failed"; tail -5 "$out/bench_$i.err"; }
done
cd "$root"
python - "$out" > "gpurun_out/${tag}_counters.txt" <<'PY'
import collections, csv, glob, sys
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(f"{root}/pass_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(acc, key=lambda k: -max(v[0] for v in acc[k].values())):
    print(k, " ".join(f"{c}={v[1] / v[0]:.4g} (n={v[0]})" for c, v in sorted(acc[k].items())))
PY
rm -rf "$out"/pass_*
cat "gpurun_out/${tag}_counters.txt"
