"""Data-parallel harness for the encoder forward: one process per GPU, images sharded, NO data-path collective.

Inference over a batch of pages is embarrassingly parallel: every rank holds a full weight replica
(343 MB fp32 for ViT-B - nothing against 288 GB of HBM) and a contiguous slice of the batch.  ``torch.distributed``
(backend ``nccl`` = RCCL over xGMI on the GPU box, ``gloo`` in the CPU tests) is used only for the control plane:
the start/stop barrier and the max-over-ranks of the measured time.  The reference has no distributed code at all
(ref README.md:59 lists it as a TODO), so there is no NCCL call pattern to mirror.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.distributed as dist


@dataclass
class Rank:
    rank: int
    world: int
    local_rank: int
    backend: str

    @property
    def is_main(self) -> bool:
        return self.rank == 0


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous ``[lo, hi)`` slice of ``total`` items owned by ``rank``; sizes differ by at most one."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request total={total} rank={rank} world={world}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init(backend: str | None = None, force_group: bool = False) -> Rank:
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (as ``torch.distributed.run`` sets them).
    A single process (no WORLD_SIZE, or 1) needs no process group; ``force_group`` creates one anyway (the 1-GPU box
    rehearses the RCCL control plane that way, tests/test_gpu_dp.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if (world > 1 or force_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return Rank(rank=rank, world=world, local_rank=local, backend=backend)


def _cpulist(text: str) -> set:
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_to_gpu_numa(local_rank: int, local_world: int) -> str:
    """Pin this process to the host CPUs of its GPU's NUMA node ("one process per GPU pinned to its NUMA
    node"): the node is read from sysfs through the device's PCI address; if the platform does not say (node -1, no sysfs,
    no affinity API) the CPUs this process may use are split evenly over the local ranks instead.  The mask only ever
    shrinks the inherited one (a cgroup / taskset limit stays in force).  Returns a short description for the bench line."""
    if not hasattr(os, "sched_getaffinity"):
        return "unpinned (no affinity API)"
    allowed = set(os.sched_getaffinity(0))
    how, want = "", set()
    try:
        p = torch.cuda.get_device_properties(local_rank)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node >= 0:
            want = _cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read()) & allowed
            how = f"NUMA node {node} of GPU {bdf}"
    except (OSError, ValueError, AttributeError, RuntimeError):
        pass
    if not want and local_world > 1:
        cpus = sorted(allowed)
        per = max(1, len(cpus) // local_world)
        want = set(cpus[local_rank * per: (local_rank + 1) * per]) or allowed
        how = f"even split of {len(cpus)} allowed CPUs over {local_world} local ranks"
    if not want or want == allowed:
        return f"{len(allowed)} CPUs (inherited mask" + (f"; {how})" if how else ")")
    try:
        os.sched_setaffinity(0, want)
    except OSError:
        return f"{len(allowed)} CPUs (could not narrow the mask)"
    return f"{len(want)} CPUs: {how}"


def barrier(r: Rank) -> None:
    if r.world > 1 or dist.is_initialized():
        if r.backend == "nccl":
            dist.barrier(device_ids=[r.local_rank])
        else:
            dist.barrier()


def max_over_ranks(r: Rank, value: float) -> float:
    if r.world == 1 and not dist.is_initialized():
        return float(value)
    dev = torch.device("cuda", r.local_rank) if r.backend == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(r: Rank, value: float) -> float:
    if r.world == 1 and not dist.is_initialized():
        return float(value)
    dev = torch.device("cuda", r.local_rank) if r.backend == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(r: Rank, values) -> list:
    """Every rank contributes the same number of floats; every rank gets ``[world][len(values)]`` back (control plane only:
    the per-rank throughput and MFMA utilisation that ``bench.py --gpus N`` reports - a slow rank must be visible, not hidden
    behind the max-over-ranks time)."""
    vals = [float(v) for v in values]
    if r.world == 1 and not dist.is_initialized():
        return [vals]
    dev = torch.device("cuda", r.local_rank) if r.backend == "nccl" else torch.device("cpu")
    mine = torch.tensor(vals, dtype=torch.float64, device=dev)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [t.cpu().tolist() for t in out]


def finalize(r: Rank) -> None:
    if dist.is_initialized():
        dist.destroy_process_group()
