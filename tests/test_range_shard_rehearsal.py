"""CPU, gloo: the north star's multi-GPU partition -- every coupled array cut by position range across the ranks, three
all-to-alls per round (triples out, verdicts back, commits out) -- rehearsed with numpy as the per-rank engine
(tools/range_shard_rehearsal.py) and compared bit for bit with the CPU oracle.  SURVEY.md §8e(1); DESIGN.md §5 quotes the
collective and byte counts."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("case,world", [("tiny_k31", 2), ("tiny_k31", 8), ("k31_ci2_200k", 3)], ids=lambda v: str(v))
def test_range_sharded_arrays_reproduce_the_oracle(case, world):
    import range_shard_rehearsal as R
    res = R.run(case, world, timeout=600)
    for r in res:
        assert "error" not in r, r.get("error")
        assert r["ok"], f"rank {r['rank']}: its range of the arrays differs from the oracle's"
    r0 = res[0]
    assert r0["successes"] == r0["oracle"][1]                       # every success of the sequential algorithm, no more
    assert r0["survivors"] == r0["n_km"] - r0["oracle"][1]          # the others are the rest table's (Q1 duplicates aside)
    assert r0["collectives"] == 3 * r0["rounds"]                    # triples, verdicts, commits: three data all-to-alls per round
