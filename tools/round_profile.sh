#!/bin/bash
# usage (on the GPU box, from the repo root): tools/round_profile.sh <tag>
# The evidence of a round in one call: bench line, rocprofv3 kernel stats + trace analyses of the same command, init(db) phase
# trace.  (PMC passes: tools/pmc_run.sh, a call of its own.)
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
python bench.py --steps 5 --warmup 1 > gpurun_out/${tag}_bench100m.json 2> gpurun_out/${tag}_bench100m.err
python tools/show_bench.py gpurun_out/${tag}_bench100m.json > gpurun_out/${tag}_bench100m.txt 2>&1 || true
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python "$root/bench.py" --steps 1 --warmup 1 --no-init-db --no-query-strings --cpu-sample 0 --no-single-model --genome-bases 0 > "$root/gpurun_out/${tag}_rocprof_run.json" 2> "$root/gpurun_out/${tag}_rocprof.err"
cd "$root"
python tools/trace_rounds.py "$out" 5 k_round_detect > "gpurun_out/${tag}_round_timings.txt" 2>&1 || true
python tools/trace_block.py "$out" 167 > "gpurun_out/${tag}_block_timeline.txt" 2>&1 || true
python tools/trace_gaps.py "$out" > "gpurun_out/${tag}_build_gaps.txt" 2>&1 || true
f=$(ls "$out"/*/*kernel_stats.csv "$out"/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "gpurun_out/${tag}_bench100m_kernel_stats.csv"
rm -rf "$out"
KMX_INIT_TRACE=1 python tools/bench_init_trace.py > gpurun_out/${tag}_init_trace.txt 2>&1 || true
cat gpurun_out/${tag}_bench100m.txt gpurun_out/${tag}_build_gaps.txt
tail -30 gpurun_out/${tag}_init_trace.txt
