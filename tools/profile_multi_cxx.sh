#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_multi_cxx.sh <tag> <handles> [ring|range]
# rocprofv3 --kernel-trace --stats of KModel::init(db) by <handles> handles from C++ (kmx_build_from_kmc_multi_ex) on the
# bench's 1e8-k-mer stream: the kernel stats of the run + the un-profiled timings with the library's phase trace.
set -e
tag=$1; h=${2:-1}; part=${3:-range}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
KMX_INIT_TRACE=1 python "$root/tools/bench_multi_cxx.py" 100000000 $h 2 parts=$part min_handles=$h nopython > "$root/gpurun_out/${tag}_plain.log" 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python "$root/tools/bench_multi_cxx.py" 100000000 $h 1 parts=$part min_handles=$h nopython > "$root/gpurun_out/${tag}_prof.log" 2>&1
cd "$root"
f=$(ls "$out"/*/*kernel_stats.csv "$out"/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "gpurun_out/${tag}_kernel_stats.csv"
tail -12 "gpurun_out/${tag}_plain.log"
rm -rf "$out"
