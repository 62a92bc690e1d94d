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


def bind_to_gpu_node(device_index: int):
    """Host side of "one process per GPU": run this rank's threads (the library's worker pool is created by the first call that
    needs it and inherits the affinity) on the NUMA node its GPU hangs off -- the streams it parses, the pinned upload buffer and
    the PCIe root of the device then sit on one socket.  Returns {"node": n, "cpus": k} or None when the topology cannot be read
    (no sysfs, no PCI address, a single node) -- nothing is changed then."""
    import os

    try:
        import torch

        p = torch.cuda.get_device_properties(device_index)
        dom, bus, dev = getattr(p, "pci_domain_id", None), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None)
        if bus is None or dev is None:
            return None
        path = f"/sys/bus/pci/devices/{(dom or 0):04x}:{bus:02x}:{dev:02x}.0/numa_node"
        with open(path) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            spec = f.read().strip()
        cpus = set()
        for part in spec.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        allowed = os.sched_getaffinity(0)
        cpus &= allowed
        if not cpus or cpus == allowed or len(cpus) * 8 < len(allowed):  # (nothing to gain, or a sliver of what we may use)
            return None
        os.sched_setaffinity(0, cpus)
        return {"node": node, "cpus": len(cpus)}
    except Exception:  # noqa: BLE001 -- a hint, never a reason to fail
        return None
