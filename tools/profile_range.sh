#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_range.sh <tag>
# rocprofv3 --kernel-trace --stats of the range partition's engine at one rank (bench.py --partition range, nothing else timed)
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python "$root/bench.py" --partition range --steps 1 --warmup 0 --no-init-db --cpu-sample 0 --no-query-strings --no-roofline --single-model-steps 2 "$@" > "$root/gpurun_out/${tag}_run.json" 2> "$root/gpurun_out/${tag}.err"
cd "$root"
f=$(ls "$out"/*/*kernel_stats.csv "$out"/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "gpurun_out/${tag}_kernel_stats.csv"
rm -rf "$out"
