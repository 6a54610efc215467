"""The N > 1 path of bench.py on hardware: `python bench.py --gpus 2` as a user types it -- the script starts its own
ranks.  On a one-GPU box the two ranks share the device (`--share-gpu`: a functional rehearsal, collective on gloo because
RCCL refuses two ranks on one device); on a box with two or more GPUs the same test takes the real path (one rank per GPU,
RCCL all-gather)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_starts_its_own_two_ranks():
    import torch
    share = [] if torch.cuda.device_count() >= 2 else ["--share-gpu"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--secondary", "config4", "--secondary-steps", "1", "--batch", "256"] + share,
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["config"]["batch_total"] == 512
    assert d["converged_fraction"] == 1.0 and len(d["kernel_ms_per_rank"]) == 2 and d["allgather_ms"] > 0
    assert d["collective_backend"] == ("gloo" if share else "rccl")
    c4 = d["secondary"]["config4"]                # 65536 Monte-Carlo problems in two shards of 32768
    assert c4["scaling"] == "strong" and c4["batch_total"] == 65536 and c4["batch_per_gpu"] == 32768 and c4["converged_fraction"] == 1.0
