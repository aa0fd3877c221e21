#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_init.sh <tag>
# rocprofv3 kernel trace of KModel::init(db) (tools/bench_init_trace.py, 3 calls): kernel time and idle time between the kernels
# of the last call -> gpurun_out/<tag>_init_gaps.txt
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_init_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out" -- python "$root/tools/bench_init_trace.py" 100000000 3 > "$root/gpurun_out/${tag}_init_run.txt" 2>&1
cd "$root"
python tools/trace_init.py "$out" > "gpurun_out/${tag}_init_gaps.txt" 2>&1 || true
rm -rf "$out"
cat "gpurun_out/${tag}_init_gaps.txt"
