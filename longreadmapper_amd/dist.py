"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards by independent units (reads); the only collective is the one-time broadcast of
the index image over xGMI.  No data-path collective exists: every rank maps its own contiguous
slice of the batch and results are concatenated in input order."""
import os

import numpy as np


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_process_group(backend=None):
    import torch
    import torch.distributed as dist
    rank, world, local = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            # LRM_DIST_BACKEND=gloo: rehearsal of the N>1 control flow on a box with fewer GPUs than ranks
            backend = os.environ.get("LRM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def partition_by_bases(lens, world):
    """Contiguous slices [lo, hi) per rank, balanced by cumulative bases rather than by read
    count (100 kbp reads next to 1 kbp reads would otherwise skew the ranks)."""
    lens = np.asarray(lens, dtype=np.int64)
    n = len(lens)
    cum = np.concatenate([[0], np.cumsum(lens)])
    total = int(cum[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        i = int(np.searchsorted(cum, target, side="left"))
        cuts.append(min(max(i, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


BCAST_CHUNK = 256 << 20     # >= 64 MiB pieces so RCCL pipelines across the xGMI links


def broadcast_blob(blob, device=None, src=0):
    """blob: torch uint8 tensor on `src` (None elsewhere).  Returns the full image on every rank,
    on `device` (a torch.device; CPU for the gloo tests)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return blob
    via_host = dist.get_backend() == "gloo" and device is not None and torch.device(device).type == "cuda"
    cdev = torch.device("cpu") if via_host else device          # gloo rehearsal: stage through host memory
    size = torch.zeros(1, dtype=torch.int64, device=cdev)
    if rank == src:
        size[0] = blob.numel()
    dist.broadcast(size, src=src)
    n = int(size.item())
    if rank == src:
        buf = blob.to(cdev) if via_host else blob
    else:
        buf = torch.empty(n, dtype=torch.uint8, device=cdev)
    for lo in range(0, n, BCAST_CHUNK):
        dist.broadcast(buf[lo:min(lo + BCAST_CHUNK, n)], src=src)
    return buf.to(device) if via_host else buf


def gather_in_order(local_arrays, slices):
    """all_gather of per-rank numpy result arrays, concatenated in input (rank) order."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_arrays
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, local_arrays)
    keys = out[0].keys()
    return {k: np.concatenate([o[k] for o in out]) for k in keys}
