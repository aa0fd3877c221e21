#!/bin/bash
# usage (GPU box, repo root): tools/ab_pregather_probe.sh <tag> -- same-box A/B of the pre-gather PROBE (kernels.hip k_probe_pregather):
# the build with and without an imitation of the next round's gathers running beside every round's ordered chain, alternating.
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${tag}.txt
: > "$out"
args="--steps 6 --warmup 1 --no-init-db --no-query-strings --no-single-model --cpu-sample 0 --genome-bases 0"
for rep in 1 2 3; do
for pr in 0 1; do
	KMX_TEST_HOOKS=1 KMX_PREGATHER_PROBE=$pr python "$root/bench.py" $args > "$root/gpurun_out/${tag}_p${pr}.json" 2> /dev/null
	python - "$pr" "$rep" "$root/gpurun_out/${tag}_p${pr}.json" >> "$out" <<'PY'
import json, sys
pr, rep, path = sys.argv[1:4]
d = json.loads(open(path).read().strip().splitlines()[-1])
kc = d.get("kernel_classes", {})
cls = " ".join(f"{k} {v['seconds_per_step'] * 1e3:.2f}" for k, v in kc.items() if k in ("commit_check", "detect", "file", "slow_path", "reorder"))
print(f"pass {rep} probe={pr}: {d['ms_per_step']:.2f} ms per build; per class (ms, timed separately): {cls}")
PY
done
done
cat "$out"
