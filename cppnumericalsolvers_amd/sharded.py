"""Multi-GPU sharding of a batch (one process per GPU, torch.distributed).

The hot path shards trivially: problems are independent (no reference code
couples them), so each rank owns the contiguous range
[rank*B/G, (rank+1)*B/G) and runs its own solve kernel.  The ONLY collective
is an all-reduce of a 3-word convergence record (RCCL over xGMI when the
backend is "nccl"; 24 bytes, latency-bound) that gives every rank the global
stop flag: `unconverged == 0` means every problem of every shard stopped on a
convergence criterion rather than on the iteration limit.
"""
from dataclasses import dataclass

import numpy as np


def shard_range(B, rank, world_size):
    """Contiguous [lo, hi) range of problems owned by `rank`."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    lo = (B * rank) // world_size
    hi = (B * (rank + 1)) // world_size
    return lo, hi


@dataclass
class GlobalFlag:
    total: int          # problems solved by all ranks
    unconverged: int    # of those, stopped by IterationLimit or still Continue/NotStarted
    iterations: int     # sum of outer iterations over all ranks
    all_converged: bool


def local_counts(status, num_iterations):
    """[total, unconverged, iterations] for one shard (numpy or torch inputs)."""
    if hasattr(status, "detach"):
        import torch
        bad = (status <= 1).sum()  # NotStarted(-1), Continue(0), IterationLimit(1)
        return torch.stack([torch.tensor(status.numel(), device=status.device, dtype=torch.int64),
                            bad.to(torch.int64), num_iterations.to(torch.int64).sum()])
    status = np.asarray(status)
    return np.array([status.size, int((status <= 1).sum()), int(np.asarray(num_iterations).sum())],
                    dtype=np.int64)


def allreduce_flag(counts, group=None):
    """Sum the per-shard counts over all ranks -> GlobalFlag (no-op without a process group)."""
    import torch
    import torch.distributed as dist
    t = counts if hasattr(counts, "detach") else torch.from_numpy(np.asarray(counts, dtype=np.int64))
    if dist.is_available() and dist.is_initialized():
        if t.is_cuda and dist.get_backend(group) == "gloo":
            t = t.cpu()   # a host-side group (tests, bench.py's shared-device dry run): 24 bytes through the host
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    v = [int(z) for z in t.cpu().tolist()]
    return GlobalFlag(total=v[0], unconverged=v[1], iterations=v[2], all_converged=(v[1] == 0))


def progress_fields_device(prog_bytes):
    """View the device progress buffer (uint8[B*40]) as (status, num_iterations, nfev, sum_k) int32
    tensors without leaving the GPU."""
    import torch
    w = prog_bytes.view(torch.int32).view(-1, 10)
    return w[:, 0], w[:, 1], w[:, 2], w[:, 3]


class ShardedLbfgs:
    """Solve this rank's shard of a global batch and agree on the global stop flag.

    `solver` is a BatchedLbfgs (GPU).  `make_x0(first, count)` builds the shard's
    start points on the solver's device (the synthetic generator is counter
    based, so no data moves between ranks).
    """

    def __init__(self, solver, rank=0, world_size=1, group=None):
        self.solver = solver
        self.rank = rank
        self.world_size = world_size
        self.group = group

    def minimize_global(self, objective, B_global, make_x0):
        lo, hi = shard_range(B_global, self.rank, self.world_size)
        x0 = make_x0(lo, hi - lo)
        x, f, g, prog = self.solver.minimize(objective, x0)
        status, iters, _, _ = progress_fields_device(prog)
        flag = allreduce_flag(local_counts(status, iters), self.group)
        return (lo, hi), (x, f, g, prog), flag


def al_progress_fields_device(prog_bytes):
    """View the device augmented-Lagrangian progress buffer (uint8[B*56], mi355_al_progress) as
    (status, num_iterations) int32 tensors without leaving the GPU."""
    import torch
    w = prog_bytes.view(torch.int32).view(-1, 14)
    return w[:, 0], w[:, 1]


class ShardedAugmentedLagrangian:
    """The constrained solves shard exactly like the unconstrained ones: rank r owns a contiguous range of start
    states, runs its own outer loop (no data-path collective: the iteration counts of different shards need not
    agree) and the ranks exchange only the 3-word record.  `solver` is a BatchedAugmentedLagrangian (GPU);
    `make_state(first, count)` builds the shard's (x, lambda, mu, penalty) tensors on the solver's device."""

    def __init__(self, solver, rank=0, world_size=1, group=None):
        self.solver = solver
        self.rank = rank
        self.world_size = world_size
        self.group = group

    def minimize_global(self, problem, B_global, make_state):
        lo, hi = shard_range(B_global, self.rank, self.world_size)
        x, lam, mu, penalty = make_state(lo, hi - lo)
        viol, kkt, prog = self.solver.minimize(problem, x, lam, mu, penalty)
        status, iters = al_progress_fields_device(prog)
        flag = allreduce_flag(local_counts(status, iters), self.group)
        return (lo, hi), (x, lam, mu, penalty, viol, kkt, prog), flag
