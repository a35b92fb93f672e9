"""Signal sharding over ranks + the single result gather.

The reference's only parallelism is ``joblib.Parallel`` over signals (reference
src/csc/dictionary.py:712,751,1106,1147): independent jobs, results collected in input order.  Here
the same independence becomes: rank r solves the contiguous slice ``shard_range(S, r, R)`` on its
own GPU with no traffic during the solve, and ONE ``torch.distributed.gather`` of fixed-size
per-signal records assembles the batch on rank 0 (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  The records stay on the device from the kernel that writes them to the
collective that sends them (no host hop).  Without an initialised process group everything
degenerates to rank 0 of 1.

Device placement: one process per GPU.  With a process group initialised, ``local_device()`` selects
``cuda:LOCAL_RANK`` (torchrun's variable; ``rank % device_count`` without it) for this process before
the first GPU call of csc_amd, and the first gather checks that no two ranks of a host share a GPU.
"""
from __future__ import annotations

import os
import socket

import numpy as np
import torch
import torch.distributed as dist

_device_checked = None      # the default process group the one-rank-per-GPU check last passed for


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized()


def rank() -> int:
    return dist.get_rank() if is_distributed() else 0


def world_size() -> int:
    return dist.get_world_size() if is_distributed() else 1


def local_device():
    """The GPU of this process (None without one).  Under an initialised multi-rank process group the
    choice is made HERE, not left to the caller: cuda:LOCAL_RANK, made current for the process (a side
    effect on purpose: ``torch.cuda.set_device`` -- every later allocation of the process lands there)."""
    if not torch.cuda.is_available():
        return None
    if is_distributed() and world_size() > 1:
        n = torch.cuda.device_count()
        lr = os.environ.get("LOCAL_RANK")
        idx = int(lr) if lr is not None else rank() % max(1, n)
        if idx >= n:
            raise RuntimeError(f"rank {rank()}: LOCAL_RANK={idx} but only {n} GPU(s) are visible "
                               "(one process per GPU: launch with --nproc-per-node <= number of GPUs)")
        if torch.cuda.current_device() != idx:
            torch.cuda.set_device(idx)
        return torch.device("cuda", idx)
    return torch.device("cuda", torch.cuda.current_device())


def _check_one_rank_per_device(dev):
    """Once per process group: fail loudly if two ranks on one host use the same GPU (RCCL would hang or
    refuse later, with a less readable message)."""
    global _device_checked
    if not is_distributed():
        _device_checked = None                    # a process group initialised later is checked again
        return
    group = dist.distributed_c10d._get_default_group()
    if _device_checked is group or world_size() == 1 or dev is None or dev.type != "cuda":
        return
    mine = (socket.gethostname(), int(dev.index if dev.index is not None else torch.cuda.current_device()))
    seen = [None] * world_size()
    dist.all_gather_object(seen, mine)
    if len(set(seen)) != len(seen):
        raise RuntimeError(f"csc_amd: two ranks share a GPU {sorted(seen)}; run one process per GPU "
                           "(torchrun --nproc-per-node N sets LOCAL_RANK, which csc_amd follows)")
    _device_checked = group


def shard_range(n: int, r: int, world: int):
    """Contiguous [lo, hi) of rank r: the first n % world ranks hold one extra signal."""
    base, extra = divmod(n, world)
    lo = r * base + min(r, extra)
    return lo, lo + base + (1 if r < extra else 0)


def shard_sizes(n: int, world: int):
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


class PendingRows:
    """A gather in flight (``gather_rows(..., async_op=True)``): ``wait()`` makes the current stream wait for the
    collective and returns what ``gather_rows`` returns.  The send / receive buffers stay referenced until then."""

    def __init__(self, work, recv, send, sizes, dst_is_me):
        self._work, self._recv, self._send, self._sizes, self._me = work, recv, send, sizes, dst_is_me

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        self._send = None
        if not self._me:
            return None
        return _assemble(self._recv, self._sizes)


def _assemble(recv, sizes):
    cap, R = max(sizes), len(sizes)
    if all(n == cap for n in sizes):
        return torch.cat(recv, dim=0) if R > 1 else recv[0]
    return torch.cat([recv[q][:sizes[q]] for q in range(R)], dim=0)


def gather_rows(local: torch.Tensor, sizes, dst: int = 0, out=None, async_op: bool = False):
    """THE collective of the path.  ``local``: this rank's [sizes[rank], width] uint8 rows, in input order, on
    this rank's GPU (nccl) or on the host (gloo).  Returns the [sum(sizes), width] tensor on rank ``dst`` (same
    device kind), None elsewhere.  Shards are padded to the largest so that one equal-size ``dist.gather``
    does it; ``out`` (rank dst: a list of world [cap, width] buffers) lets a caller reuse the receive
    buffers from call to call -- the result is then assembled from those buffers and, with one rank, IS
    ``out[0]`` (no copy): the next call with the same ``out`` overwrites it, so consume or clone it first.
    ``async_op=True``: the collective is only enqueued (the solve of the next batch can run under it) and a
    ``PendingRows`` is returned; its ``wait()`` gives the rows."""
    assert local.dtype == torch.uint8 and local.dim() == 2
    if not is_distributed():
        assert local.shape[0] == sum(sizes)
        return PendingRows(None, [local], None, [local.shape[0]], True) if async_op else local
    R, r = world_size(), rank()
    assert len(sizes) == R and local.shape[0] == sizes[r], (tuple(local.shape), sizes, r)
    _check_one_rank_per_device(local.device)
    cap, width = max(sizes), local.shape[1]
    if sizes[r] == cap:
        send = local.contiguous()
    else:
        send = torch.zeros((cap, width), dtype=torch.uint8, device=local.device)
        send[:sizes[r]] = local
    recv = None
    if r == dst:
        recv = out if out is not None else [torch.empty_like(send) for _ in range(R)]
    work = dist.gather(send, recv, dst=dst, async_op=async_op)
    if async_op:
        return PendingRows(work, recv, send, list(sizes), r == dst)
    if r != dst:
        return None
    return _assemble(recv, sizes)


def gather_records(local: np.ndarray, total: int, dst: int = 0, device=None):
    """NumPy front of ``gather_rows`` for host-resident rows ([n_local, rec_bytes] uint8 -> [total, rec_bytes]
    on rank ``dst``, None elsewhere); used where the rows are produced on the host (the MP pipelines)."""
    local = np.ascontiguousarray(local)
    assert local.dtype == np.uint8 and local.ndim == 2
    if not is_distributed():
        assert local.shape[0] == total
        return local
    sizes = shard_sizes(total, world_size())
    t = torch.from_numpy(local)
    if dist.get_backend() == "nccl":
        t = t.to(torch.device(device) if device is not None else local_device())
    got = gather_rows(t, sizes, dst)
    return None if got is None else got.cpu().numpy()


def barrier():
    if is_distributed():
        dist.barrier()
