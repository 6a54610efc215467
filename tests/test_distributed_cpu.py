"""The N > 1 path on CPU: two gloo ranks shard a batch, solve their shards (here with the float64
oracle standing in for the GPU kernel -- tests may use it), all-gather the compact outputs, and
every rank must hold exactly what a single process computes for the whole batch."""
import os
import subprocess
import sys

import numpy as np
import pytest

import cmpc_amd as cm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import cmpc_amd as cm
from oracle import oracle_lib as ol, problem_nlp
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
T = int(sys.argv[3])
cfg, P, X0 = cm.synthetic.config3_external_push(T)
lo, hi = cm.distributed.shard_bounds(T, world, rank)
counts = [b - a for a, b in (cm.distributed.shard_bounds(T, world, r) for r in range(world))]
# a rank that builds only its shard gets the same rows as a slice of the whole batch
_, Ps, X0s = cm.synthetic.config4_monte_carlo(T, seed=1, shard=(lo, hi))
assert (Ps == P[lo:hi]).all() and (X0s == X0[lo:hi]).all()
X, info = ol.ref_solve_batch(problem_nlp.oracle_cfg(cfg), P[lo:hi], X0[lo:hi], ol.ipm_opts(tol=1e-8, mu_min=1e-9))
info8 = np.zeros((hi - lo, 8)); info8[:, 0] = info[:, 0]; info8[:, 5] = info[:, 5]
cols = torch.from_numpy(cm.distributed.compact_columns(cfg.N))
local = cm.distributed.compact_output(torch.from_numpy(X), torch.from_numpy(info8), cols)
full = cm.distributed.all_gather_solutions(local, world, counts=counts)
np.save(os.path.join(sys.argv[2], f"gathered_{rank}.npy"), full.numpy())
dist.destroy_process_group()
'''


def test_shard_bounds_partition():
    for total, world in ((65536, 8), (10, 3), (7, 8)):
        b = [cm.distributed.shard_bounds(total, world, r) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == total and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
    assert cm.distributed.shard_bounds(65536, 8, 3) == (3 * 8192, 4 * 8192)


@pytest.mark.parametrize("total,port", [(8, 29611), (7, 29613)])   # equal and ragged shards
def test_two_rank_gloo_gather_matches_single_process(tmp_path, total, port):
    from oracle import oracle_lib as ol, problem_nlp
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, str(tmp_path), str(total)],
                          env=env, timeout=300)
    cfg, P, X0 = cm.synthetic.config3_external_push(total)
    X, info = ol.ref_solve_batch(problem_nlp.oracle_cfg(cfg), P, X0, ol.ipm_opts(tol=1e-8, mu_min=1e-9))
    info8 = np.zeros((total, 8)); info8[:, 0] = info[:, 0]; info8[:, 5] = info[:, 5]
    ref = cm.distributed.compact_output(X, info8, cm.distributed.compact_columns(cfg.N))
    assert ref.shape == (total, 3 * (cfg.N + 1) + 24 + 12 + 2)
    for r in range(2):
        got = np.load(tmp_path / f"gathered_{r}.npy")
        np.testing.assert_array_equal(got, ref)


def test_bench_multi_gpu_launch_fails_with_a_message_not_an_assert():
    """`python bench.py --gpus 2` starts its own ranks; on a machine without (enough) GPUs it must say so."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    assert "bench.py:" in r.stderr and "GPU" in r.stderr and "AssertionError" not in r.stderr
