"""Voxel sharding across the GPUs of one node (one process per GPU, torch.distributed).

The path is embarrassingly parallel over voxels (reference: mp.Pool over voxels, mf.py:978-1009):
rank r takes a contiguous ROI-order block, the dictionary tables are broadcast once from rank 0
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests), results are gathered by
concatenation.  No collective runs inside the voxel loop.
"""
import numpy as np


def shard_range(V, rank, world):
    """Contiguous [lo, hi) block of rank `rank`; sizes differ by at most one voxel."""
    base, rem = divmod(int(V), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_interpolator(ms, scheme, src=0, device=None, group=None):
    """Broadcast (MultiShellInterpolator, scheme) from `src`; returns them on every rank.

    Tensors travel through the process group (RCCL when `device` is a CUDA device); the small
    shape header rides along as a pickled object."""
    import torch
    import torch.distributed as dist
    from .mf_utils import MultiShellInterpolator
    rank = dist.get_rank(group)
    if rank == src:
        hdr, flat = ms.pack()
        meta = [hdr, int(flat.size), tuple(np.asarray(scheme).shape)]
    else:
        meta = [None, None, None]
    dist.broadcast_object_list(meta, src=src, group=group)
    hdr, nflat, sshape = meta
    dev = device if device is not None else torch.device("cpu")
    t_flat = torch.empty(nflat, dtype=torch.float64, device=dev)
    t_sch = torch.empty(sshape, dtype=torch.float64, device=dev)
    if rank == src:
        t_flat.copy_(torch.from_numpy(flat))
        t_sch.copy_(torch.from_numpy(np.ascontiguousarray(scheme, dtype=np.float64)))
    dist.broadcast(t_flat, src=src, group=group)
    dist.broadcast(t_sch, src=src, group=group)
    if rank == src:
        return ms, np.asarray(scheme, dtype=np.float64)
    return MultiShellInterpolator.unpack(hdr, t_flat.cpu().numpy()), t_sch.cpu().numpy()


def gather_rows(local_rows, V, group=None, device=None):
    """All-gather per-rank row blocks (shard_range order) into the full [V, P] array on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    P = local_rows.shape[1]
    dev = device if device is not None else torch.device("cpu")
    sizes = [shard_range(V, r, world)[1] - shard_range(V, r, world)[0] for r in range(world)]
    mx = max(sizes)
    buf = torch.zeros((mx, P), dtype=torch.float64, device=dev)
    buf[:local_rows.shape[0]] = torch.as_tensor(np.ascontiguousarray(local_rows), dtype=torch.float64)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    return np.concatenate([outs[r][:sizes[r]].cpu().numpy() for r in range(world)], axis=0)
