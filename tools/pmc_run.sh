#!/bin/bash
# usage (on the GPU box, from the repo root): KMX_GIT_HEAD=<commit> tools/pmc_run.sh <tag>
# HBM traffic per kernel from rocprofv3 PMC counters, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE passes, nothing else traced; the absolute scale is calibrated on kmx_microbench kernels with a
# known touch count under the same counters (tools/pmc_summary.py).  -> gpurun_out/<tag>_pmc_traffic.json
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d "$out/pmc_$C" -- python "$root/bench.py" --steps 1 --warmup 0 --no-roofline --no-init-db --no-query-strings --cpu-sample 0 --no-single-model --genome-bases 0 > "$out/bench_$C.json" 2> "$out/bench_$C.err"
  echo "bench under $C done"
  rocprofv3 --pmc $C --output-format csv -d "$out/pmcmicro_$C" -- python "$root/tools/microbench_pmc.py" > "$out/micro_$C.log" 2>&1
  echo "microbench under $C done"
done
cd "$root"
nq=$(python -c "import json,re,sys; d=json.loads(open('$out/bench_FETCH_SIZE.json').read().strip().splitlines()[-1]); print(re.search(r'\((\d+) queries', d['config']['workload']).group(1))")
cp "$out/bench_FETCH_SIZE.json" "gpurun_out/${tag}_pmc_bench_line.json"
python tools/pmc_summary.py "$out" "gpurun_out/${tag}_pmc_traffic.json" "$tag" "$nq" > "gpurun_out/${tag}_pmc_summary.txt"
rm -rf "$out"/pmc_* "$out"/pmcmicro_*
cat "gpurun_out/${tag}_pmc_summary.txt"
