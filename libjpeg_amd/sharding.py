"""Image-granular sharding for the batch case (BASELINE config 4): frames are independent, so rank r simply
takes frames r, r + world, r + 2 world, ... and nothing is exchanged on the data path.  The only collectives
are the barriers around the timed region and the reduction of the two scalars a benchmark reports
(RCCL when the ranks hold GPUs -- backend "nccl" -- and gloo in the CPU tests)."""
from __future__ import annotations

import time


def frames_of_rank(n_frames: int, rank: int, world: int) -> list[int]:
    return list(range(rank, n_frames, world))


def _sync():
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass


def timed_region(work, dist=None):
    """barrier + device sync on both sides of work(); returns (seconds, whatever work() returned)."""
    _sync()
    if dist is not None:
        dist.barrier()
    _sync()
    t0 = time.perf_counter()
    units = work()
    _sync()
    if dist is not None:
        dist.barrier()
    _sync()
    return time.perf_counter() - t0, units


def reduce_result(units, elapsed, dist=None):
    """(sum of the units over ranks, max of the elapsed time over ranks)."""
    if dist is None:
        return units, elapsed
    import torch

    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    u = torch.tensor([float(units)], dtype=torch.float64, device=dev)
    t = torch.tensor([float(elapsed)], dtype=torch.float64, device=dev)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(round(float(u[0]))), float(t[0])
