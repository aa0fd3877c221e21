"""bench.py's N > 1 control flow end to end on real kernels: two ranks sharing cuda:0 over gloo (KMX_BENCH_REHEARSAL=1, the
transport a one-GPU box allows; on a node the same code runs one rank per GPU over RCCL), launched the way the driver
launches it.  For both partitions of the single model: ONE JSON line from rank 0, `value` = the one model all ranks built
(replicas beside it), the layout fields of the partition, exit status 0."""
import json
import os
import subprocess
import sys

import pytest

from dist_workers import free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("partition", ["ring", "range"])
def test_bench_two_ranks_one_line(partition):
    env = dict(os.environ, KMX_BENCH_REHEARSAL="1", KMX_BENCH_CXX_MULTI="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--kmers", "3000000", "--partition", partition, "--cpu-sample", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(p.stdout.strip().splitlines()) == 1, p.stdout[-2000:]      # ONE line on stdout, whatever the libraries print
    d = json.loads(lines[0])
    sm = d["single_model"]
    assert d["n_gpus"] == 2 and d["value"] == sm["value"] and d["replica_value"] > 0 and "ONE model" in d["value_is"]
    assert sm["partition"] == partition and sm["kmers"] == 2 * d["config"]["kmers_per_gpu"] and f"--partition {partition}" in d["config"]["parallelism"]
    assert d["scaling"].startswith("weak") and sm["bytes_exchanged_per_build"] > 0
    if partition == "range":
        assert sm["all_to_alls_per_build"] >= 2 * sm["blocks"] * d["config"]["nb"] and sm["idle_ranks_in_the_ordered_rounds"] == 0
    else:
        assert sm["array_owners"] == 2 and sm["ring_hops_per_build"] > 0
    other = sm["other_partition"]                                    # the partition that was not selected is timed beside the headline
    assert other["partition"] != partition and other["value"] > 0 and other["kmers"] == sm["kmers"]
    # the C++ entry driven from rank 0's process (on a node: one handle per GPU; here both on cuda:0, so RCCL -- one rank per device -- refuses)
    cx = sm["cxx_multi"]
    assert cx["handles"] == 2 and cx["range"]["value"] > 0 and cx["ring"]["value"] > 0 and cx["range"]["same_model"] and cx["ring"]["same_model"]
    assert "one handle per device" in cx["range-rccl"]["error"]
    # one model = the sequential build of the concatenated streams: what all ranks inserted and what went to the rest table add up
    st = sm["stats"]
    assert st["successes"] + st["rest_entries"] >= st["n_km"] and st["attempts"] >= st["n_km"]
