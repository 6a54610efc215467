"""Multi-GPU use of the batched solver: the batch dimension shards trivially (problems are
independent, no exchange during the solve); the only collective is one all-gather of the compact
solutions at the end (RCCL over xGMI when torch.distributed runs on `nccl`)."""
from __future__ import annotations

import numpy as np

from .layout import Layout


def shard_bounds(total: int, world: int, rank: int):
    """Contiguous split: rank r owns problems [lo, hi)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def compact_columns(N: int) -> np.ndarray:
    """Column indices of x that make the compact per-problem output: CoM trajectory 3(N+1),
    first-knot corner forces 24, knot-0 and knot-1 foot positions 12."""
    L = Layout(N)
    cols = list(range(L.com, L.com + 3 * (N + 1)))
    for c in range(2):
        for j in range(4):
            cols += list(range(L.f[c][j], L.f[c][j] + 3))
    for c in range(2):
        cols += list(range(L.pos[c], L.pos[c] + 6))
    return np.asarray(cols, np.int64)


def compact_output(X, info, cols):
    """[B, len(cols) + 2]: selected columns of x, then iterations and status.  X/info/cols are all
    torch tensors (any device) or all numpy arrays."""
    if isinstance(X, np.ndarray):
        return np.concatenate([X[:, cols], info[:, [0, 5]]], axis=1)
    import torch
    return torch.cat([X.index_select(1, cols), info[:, [0, 5]]], dim=1)


def all_gather_solutions(local, world: int, out=None, force: bool = False, counts=None):
    """Gathers the per-rank compact outputs [B_r, W] into [sum B_r, W] on every rank (torch.distributed must be
    initialised when world > 1).  Equal shards: one all_gather_into_tensor straight into `out` (a preallocated
    result for steady-state loops).  `counts` (problems per rank, from shard_bounds) allows ragged shards: every
    rank pads to the largest shard and the padding is dropped after the collective.  `force` runs the collective
    for one rank too (rehearsal on a single GPU).  On the `gloo` backend (CPU tests, --share-gpu rehearsals)
    device tensors are staged through the host."""
    if world == 1 and not force:
        return local
    import torch
    import torch.distributed as dist
    total = int(sum(counts)) if counts is not None else world * local.shape[0]
    if out is None:
        out = torch.empty((total, local.shape[1]), dtype=local.dtype, device=local.device)
    staged = local.is_cuda and dist.get_backend() == "gloo"
    src = local.contiguous().cpu() if staged else local.contiguous()
    if counts is None or len(set(int(c) for c in counts)) == 1:
        if staged:
            tmp = torch.empty((total, local.shape[1]), dtype=local.dtype)
            dist.all_gather_into_tensor(tmp, src)
            out.copy_(tmp)
        else:
            dist.all_gather_into_tensor(out, src)
        return out
    mx = int(max(counts))
    pad = torch.zeros((mx, local.shape[1]), dtype=src.dtype, device=src.device)
    pad[:src.shape[0]] = src
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    o = 0
    for r, c in enumerate(counts):
        out[o:o + int(c)] = parts[r][:int(c)].to(out.device)
        o += int(c)
    return out
