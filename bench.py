#!/usr/bin/env python
"""Benchmark of the hot path: batched centroidal-MPC solves (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
(N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`,
one rank per GPU.)  A step = one batched solve of the workload on every GPU (weak scaling: each rank
solves its own batch), inputs resident in HBM, followed for N > 1 by the RCCL all-gather of the
compact solutions.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# canonical algorithmic work: F_iter(N) = 4.25e5 * N flop per interior-point iteration
F_ITER_PER_STAGE = 4.25e5
PEAK_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32 vector == f32-input MFMA peak


def cpu_baseline(cfg, P32, X032, tol, mu_min, sample):
    """oracle/ipm_ref.c (float64 port of the same algorithm) on the host cores, OpenMP over a
    bounded sample of the same workload."""
    from oracle import oracle_lib as ol, problem_nlp
    # the box's CPU share for one GPU is 16 cores; never more threads than the affinity mask allows
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    oc = problem_nlp.oracle_cfg(cfg)
    reps = int(np.ceil(sample / P32.shape[0]))
    P = np.tile(P32.astype(np.float64), (reps, 1))[:sample]
    X0 = np.tile(X032.astype(np.float64), (reps, 1))[:sample]
    ol.ref_solve_batch(oc, P[:cores], X0[:cores], ol.ipm_opts(tol=tol, mu_min=mu_min, max_iter=60), nthreads=cores)
    t = time.perf_counter()
    _, info = ol.ref_solve_batch(oc, P, X0, ol.ipm_opts(tol=tol, mu_min=mu_min, max_iter=60), nthreads=cores)
    dt = time.perf_counter() - t
    return {"value": round(sample / dt, 1), "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": f"{sample} problems of the same workload, float64, OpenMP x{cores}, {dt:.1f} s wall, "
                      f"{int((info[:, 5] == 0).sum())}/{sample} converged, mean {info[:, 0].mean():.1f} iterations"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="problems per GPU (config 2: 256)")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3"])
    ap.add_argument("--cpu-sample", type=int, default=6144)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-collective", action="store_true", help="run the all-gather path with one rank too (rehearsal on a single GPU)")
    args = ap.parse_args()

    import torch
    import cmpc_amd as cm

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll = world > 1 or args.force_collective
    if coll:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    B = args.batch
    gen = cm.synthetic.config2_perturbed_com if args.workload == "config2" else cm.synthetic.config3_external_push
    cfg, P, X0 = gen(B, seed=(0 if args.workload == "config2" else 1) + 1000 * rank)
    P32, X032 = P.astype(np.float32), X0.astype(np.float32)
    solver = cm.BatchSolver(cfg, B, device=local_rank)
    dP, dX0 = torch.from_numpy(P32).to(dev), torch.from_numpy(X032).to(dev)
    dX = torch.empty_like(dX0)
    dInfo = torch.empty((B, 8), dtype=torch.float32, device=dev)
    # N > 1: one kernel packs the compact record of every problem, one RCCL all-gather shares it (preallocated buffers)
    W = 3 * (cfg.N + 1) + 38
    cbuf = torch.empty((B, W), dtype=torch.float32, device=dev) if coll else None
    gbuf = torch.empty((world * B, W), dtype=torch.float32, device=dev) if coll else None

    def gather():
        return cm.distributed.all_gather_solutions(solver.compact_output_device(dX, dInfo, cbuf), world, out=gbuf, force=True)

    def step():
        solver.solve_device(dP, dX0, dX, dInfo)
        return gather() if coll else None

    def barrier():
        if coll:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * args.steps)]
    ks = solver.launch_stream       # the HIP stream the solve kernel is launched on
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[2 * i].record(ks)
        solver.solve_device(dP, dX0, dX, dInfo)
        ev[2 * i + 1].record(ks)
        if coll:
            gather()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = np.array([ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(args.steps)])
    info = dInfo.cpu().numpy()
    ag_ms = None
    if coll:  # packing + collective alone, for the record
        barrier()
        t1 = time.perf_counter()
        for _ in range(10):
            gather()
        barrier()
        ag_ms = (time.perf_counter() - t1) / 10 * 1e3

    if rank == 0:
        total = world * B * args.steps
        iters = info[:, 0]
        traffic = None   # HBM bytes per launch from the committed PMC passes (profiles/), same command
        try:
            import glob
            pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
            if pm and args.workload == "config2" and B == 256:
                traffic = json.load(open(pm[-1]))["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        flop_per_launch = float(iters.sum()) * F_ITER_PER_STAGE * cfg.N
        achieved = flop_per_launch / (kern_ms.mean() * 1e-3) / 1e12
        out = {
            "metric": "centroidal-MPC solves/sec (batch, horizon=20)",
            "value": round(total / elapsed, 1), "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: batch={B}/GPU perturbed iCub/ergoCub problems, ergoCubGazeboV1 "
                                   f"parameters, horizon={cfg.N}, dt={cfg.sampling_time}, cold start, converged to 1e-6",
                       "batch_per_gpu": B, "horizon": cfg.N, "precision": "f32 storage+factorisation, f64 residuals"},
            "p50_solve_latency_ms": round(float(np.median(kern_ms)), 4),
            "iterations_mean": round(float(iters.mean()), 2), "iterations_max": int(iters.max()),
            "converged_fraction": round(float((info[:, 5] == 0).mean()), 4),
            "roofline": {"bound": "mfma", "achieved": round(achieved, 4), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F32_TFLOPS, 5), "traffic": traffic,
                         "kernel": "cmpc_solve_kernel", "kernel_ms_avg": round(float(kern_ms.mean()), 4),
                         "algorithmic_flop_per_launch": flop_per_launch,
                         "note": "f32 vector peak == f32-input MFMA peak (157.3 TFLOP/s); flop = executed IP iterations x "
                                 "4.25e5*N (SURVEY 8d canonical count); HBM traffic is ~11 KB/solve (see profiles/)"},
        }
        if ag_ms is not None:
            out["allgather_ms"] = round(ag_ms, 4)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, P32, X032, 1e-6, 5e-8, args.cpu_sample)
        print(json.dumps(out), flush=True)
    if coll:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
