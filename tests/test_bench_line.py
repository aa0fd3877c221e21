"""bench.py's line at N > 1: `value` is the ONE model all ranks build together, the independent models move to `replica_*`
(the reference's KModel::init is one model, kmodel.hpp:57-86); at N = 1 the headline stays the plain build."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

LINE = {"metric": "m", "value": 8.0e9, "unit": "k-mers/s", "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 100.0,
        "query_value": 2.0e10, "query_ms_per_step": 44.0, "config": {"workload": "w", "parallelism": "8 independent models"}}
SINGLE = {"value": 3.0e9, "ms_per_build": 266.0, "query_value": 1.9e10, "query_ms_per_step": 46.0, "steps": 20, "transport": "nccl"}


def test_single_model_becomes_the_headline_at_n_gt_1():
    out = bench.promote_single_model(copy.deepcopy(LINE), dict(SINGLE), 8)
    assert (out["value"], out["ms_per_step"], out["query_value"], out["query_ms_per_step"]) == (3.0e9, 266.0, 1.9e10, 46.0)
    assert (out["replica_value"], out["replica_ms_per_step"], out["replica_query_value"], out["replica_query_ms_per_step"]) == (8.0e9, 100.0, 2.0e10, 44.0)
    assert out["steps"] == 20 and out["single_model"]["transport"] == "nccl"
    assert "one model over 8 ranks" in out["config"]["parallelism"] and "ONE model" in out["value_is"]


def test_one_rank_keeps_the_plain_build_as_headline():
    out = bench.promote_single_model(copy.deepcopy(LINE), dict(SINGLE), 1)
    assert out["value"] == 8.0e9 and "replica_value" not in out and out["single_model"]["value"] == 3.0e9


def test_a_failed_single_model_leg_leaves_no_headline():
    """N > 1 and the exchange failed: `value` (ONE model) was not measured -- the replicas' rate must not stand in for it
    (bench.py then exits 3, like its watchdog)."""
    out = bench.promote_single_model(copy.deepcopy(LINE), {"error": "exchange timed out"}, 8)
    assert out["value"] is None and out["replica_value"] == 8.0e9 and out["single_model"]["error"] and out["failed"]
    assert "NOT MEASURED" in out["value_is"]


def test_a_switched_off_single_model_leg_keeps_the_replica_headline():
    out = bench.promote_single_model(copy.deepcopy(LINE), None, 8)
    assert out["value"] == 8.0e9 and "replica_value" not in out
