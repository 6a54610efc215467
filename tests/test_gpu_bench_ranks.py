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


def test_bench_on_rccl_under_the_drivers_launcher_one_rank():
    """The collective path on RCCL itself, launched the way the driver launches N > 1 (`python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`), with the one rank a one-GPU box allows:
    `nccl` process group on the rank's device, barrier, the all-gather of the compacted records after every launch, max over ranks.
    What N > 1 adds to this is peers, not code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--force-collective", "--no-cpu-baseline", "--secondary", "none"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["ranks"] == 1 and d["collective_backend"] == "rccl" and d["allgather_ms"] > 0
    assert d["converged_fraction"] == 1.0 and d["sync_giveups"] == 0 and d["config"]["batch_total"] == 256
