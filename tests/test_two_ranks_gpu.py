"""The N > 1 code path of bench.py on a 1-GPU box: two ranks, both on GPU 0, collectives over gloo (BMQ_BENCH_ONE_GPU=1) -- what the
driver launches for the scaling curve (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`), end to end: tenant
sharding, the step loop with BOTH exchange forms (fan-out all-gather in the timed region, all-gatherv of the CSR in the extra steps), the
node-wide batch (device partition through bmq_partition_batch_dev, hot-tenant filter split, fan-out all-reduce) and the JSON line.
The numbers of such a run mean nothing; the test keeps the path from rotting."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, BMQ_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29613",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--topics", "100000", "--no-cpu-baseline", "--csr-exchange-steps", "2",
           "--node-batch-steps", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["parallelism"] == "tenant-shard x2" and d["config"]["exchange_impl"]
    ex = d["exchange"]
    assert ex["fanout_ms"] > 0 and ex["csr_ms"] > 0  # both exchange forms ran
    nb = d["node_batch"]
    assert nb and "error" not in nb and "skipped" not in nb
    assert len(nb["publishes_per_rank"]) == 2 and nb["imbalance_max_over_mean"] >= 1.0 and nb["fanout_total"] > 0
