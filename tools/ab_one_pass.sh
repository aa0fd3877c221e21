#!/bin/bash
# usage (GPU box, repo root): tools/ab_one_pass.sh <tag> -- same-box A/B of KModel::init(db) in two passes (the product: a host scan of the
# counter bytes, then the listing streamed under the rounds) against ONE pass (KMX_ONE_PASS=1: the listing streamed to the device and
# counted there, then the build on the resident listing), 1e8 31-mers, KMC1 and KMC2 layouts, alternating.
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${tag}.txt
: > "$out"
for layout in kmc1 kmc2; do
for rep in 1 2; do
for op in 0 1; do
	KMX_TEST_HOOKS=1 KMX_ONE_PASS=$op KMX_INIT_TRACE=0 python "$root/tools/bench_init_trace.py" 100000000 4 layout=$layout 2>/dev/null | grep "^init(db)" | tail -3 | sed "s/^/pass $rep one_pass=$op: /" >> "$out"
done
done
done
cat "$out"
