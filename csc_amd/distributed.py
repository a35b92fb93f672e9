"""Signal sharding over ranks + the single result gather.

The reference's only parallelism is ``joblib.Parallel`` over signals (reference
src/csc/dictionary.py:712,751,1106,1147): independent jobs, results collected in input order.  Here
the same independence becomes: rank r solves the contiguous slice ``shard_range(S, r, R)`` on its
own GPU with no traffic during the solve, and ONE ``torch.distributed.gather`` of fixed-size
per-signal records assembles the batch on rank 0 (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  Without an initialised process group everything degenerates to rank 0 of 1.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized()


def rank() -> int:
    return dist.get_rank() if is_distributed() else 0


def world_size() -> int:
    return dist.get_world_size() if is_distributed() else 1


def shard_range(n: int, r: int, world: int):
    """Contiguous [lo, hi) of rank r: the first n % world ranks hold one extra signal."""
    base, extra = divmod(n, world)
    lo = r * base + min(r, extra)
    return lo, lo + base + (1 if r < extra else 0)


def shard_sizes(n: int, world: int):
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def gather_records(local: np.ndarray, total: int, dst: int = 0, device=None):
    """One collective: rows of ``local`` ([n_local, rec_bytes] uint8, this rank's shard in input
    order) -> [total, rec_bytes] on rank ``dst`` (None elsewhere).  Shards are padded to the
    largest shard so a plain equal-size gather is enough."""
    local = np.ascontiguousarray(local)
    assert local.dtype == np.uint8 and local.ndim == 2
    if not is_distributed():
        assert local.shape[0] == total
        return local
    R, r = world_size(), rank()
    sizes = shard_sizes(total, R)
    assert local.shape[0] == sizes[r], (local.shape, sizes, r)
    width = local.shape[1]
    cap = max(sizes)
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    if device is not None and on_gpu:
        dev = torch.device(device)
    send = torch.zeros((cap, width), dtype=torch.uint8, device=dev)
    if sizes[r]:
        send[:sizes[r]] = torch.from_numpy(local).to(dev)
    recv = [torch.empty_like(send) for _ in range(R)] if r == dst else None
    dist.gather(send, recv, dst=dst)
    if r != dst:
        return None
    parts = [recv[q][:sizes[q]].cpu().numpy() for q in range(R)]
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, width), np.uint8)


def barrier():
    if is_distributed():
        dist.barrier()
