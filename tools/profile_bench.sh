#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_bench.sh <tag> [bench.py args...]
# rocprofv3 --kernel-trace --stats of one bench.py run -> gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_rounds.txt
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python "$root/bench.py" --steps 1 --warmup 1 --no-init-db --cpu-sample 0 --no-single-model "$@" > "$root/gpurun_out/${tag}_rocprof_run.json" 2> "$root/gpurun_out/${tag}_rocprof.err"
cd "$root"
python tools/trace_rounds.py "$out" 5 k_round_detect > "gpurun_out/${tag}_rounds.txt" 2>&1 || true
python tools/trace_block.py "$out" 134 > "gpurun_out/${tag}_block_first.txt" 2>&1 || true      # first block of the last of 3 builds (10^8 k-mers: 67 blocks each)
python tools/trace_block.py "$out" 167 > "gpurun_out/${tag}_block_mid.txt" 2>&1 || true
f=$(ls "$out"/*/*kernel_stats.csv "$out"/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "gpurun_out/${tag}_kernel_stats.csv"
rm -rf "$out"
