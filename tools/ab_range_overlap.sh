#!/bin/bash
# usage (GPU box, repo root): tools/ab_range_overlap.sh <tag>  -- same-box A/B of KMX_RANGE_OVERLAP = 0 / 1 / 2 for the C++ range build,
# 1 and 2 handles on cuda:0, 1e8 k-mers: the rounds alone (phase trace of handle 0) and the whole init(db)
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${tag}.txt
: > "$out"
for rep in 1 2; do
for ov in 0 1 2; do
	KMX_TEST_HOOKS=1 KMX_RANGE_OVERLAP=$ov KMX_INIT_TRACE=1 python "$root/tools/bench_multi_cxx.py" 100000000 2 2 parts=range nopython > "$root/gpurun_out/${tag}_ov${ov}.log" 2>&1
	python - "$ov" "$rep" "$root/gpurun_out/${tag}_ov${ov}.log" >> "$out" <<'PY'
import re, sys
ov, rep, path = sys.argv[1:4]
rounds = {}
cur = None
for line in open(path):
    m = re.match(r"\[kmx multi range P=(\d+)\] (.*?)\s+([\d.]+) ms", line)
    if m:
        p, what, t = int(m.group(1)), m.group(2).strip(), float(m.group(3))
        if what == "decode, classify, routing": cur = t
        if what == "rounds done on every handle" and cur is not None: rounds.setdefault(p, []).append(t - cur)
    m = re.match(r"range (\d+) handles: \[(.*)\]", line)
    if m:
        print(f"pass {rep} KMX_RANGE_OVERLAP={ov} handles={m.group(1)}: rounds alone {['%.1f' % x for x in rounds.get(int(m.group(1)), [])[1:]]} ms, init(db) {m.group(2)} ms")
PY
done
done
cat "$out"
