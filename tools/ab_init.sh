#!/bin/bash
# usage (GPU box): tools/ab_init.sh <tag> "VAR=.." "" ...   -- KModel::init(db) of 10^8 k-mers, 4 calls per environment setting, alternating
# (same-box A/B of the host feed; KMX_TEST_HOOKS=1 is exported so that KMX_LIBRARY variants load)
tag=$1; shift
export KMX_TEST_HOOKS=1
out=gpurun_out/${tag}_ab_init.txt; : > $out
i=0
for setting in "$@"; do
	i=$((i+1))
	echo "=== [$i] ${setting:-default}" >> $out
	env $setting python tools/bench_init_trace.py 100000000 4 2>/dev/null | grep "init(db) rep" | tail -3 >> $out
done
cat $out
