#!/bin/bash
# usage (GPU box): tools/ab_bench.sh <tag> "VAR=.. VAR=.." "VAR=.." ...   -- one short bench.py run per environment setting
# (test hooks; KMX_TEST_HOOKS=1 is exported), per-class kernel times of each into gpurun_out/<tag>_ab.txt
tag=$1; shift
export KMX_TEST_HOOKS=1
out=gpurun_out/${tag}_ab.txt; : > $out
i=0
for setting in "$@"; do
	i=$((i+1))
	echo "=== [$i] ${setting:-default}" >> $out
	env $setting python bench.py --steps 3 --warmup 1 --no-init-db --no-single-model --no-query-strings --cpu-sample 0 --genome-bases 0 > gpurun_out/${tag}_ab_$i.json 2> gpurun_out/${tag}_ab_$i.err
	python tools/show_bench.py gpurun_out/${tag}_ab_$i.json >> $out 2>&1
done
cat $out
