"""Voxel sharding across the GPUs of one node (one process per GPU, torch.distributed).

The path is embarrassingly parallel over voxels (reference: mp.Pool over voxels, mf.py:978-1009):
rank r takes a contiguous ROI-order block, the dictionary tables are broadcast once from rank 0
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests), results are gathered by
concatenation.  No collective runs inside the voxel loop.  When the voxel classes are mixed (numfasc, CSF and EAR
masks: the per-voxel cost differs by up to 40x between classes) `balanced_shard_indices` deals the voxels of every
class round-robin over the ranks instead, and `gather_rows_indexed` puts the rows back in ROI order.
"""
import numpy as np


def shard_range(V, rank, world):
    """Contiguous [lo, hi) block of rank `rank`; sizes differ by at most one voxel."""
    base, rem = divmod(int(V), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def balanced_shard_indices(numfasc, csf, ear, rank, world):
    """ROI indices (ascending) of rank `rank` when the voxels of each class (numfasc, csf, ear) are dealt round-robin
    over the ranks, so that every rank gets the same mix of classes (the per-voxel cost depends on the
    class); the start of the deal rotates from class to class so that no rank collects all the remainders."""
    numfasc = np.asarray(numfasc).astype(np.int64).ravel()
    V = numfasc.size
    c = np.zeros(V, np.int64) if csf is None else np.asarray(csf).astype(np.int64).ravel()
    e = np.zeros(V, np.int64) if ear is None else np.asarray(ear).astype(np.int64).ravel()
    key = numfasc * 4 + (c != 0) * 2 + (e != 0)
    parts, shift = [], 0
    for k in np.unique(key):
        members = np.flatnonzero(key == k)
        parts.append(members[(rank - shift) % world::world])
        shift = (shift + members.size) % world
    return np.sort(np.concatenate(parts)) if parts else np.zeros(0, np.int64)


def gather_rows_indexed(local_rows, local_idx, V, group=None, device=None):
    """All-gather row blocks that carry their ROI indices (balanced_shard_indices) into the full [V, P] array."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    P = local_rows.shape[1]
    n = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=dev)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    mx = max(int(x.item()) for x in ns)
    buf = torch.zeros((mx, P + 1), dtype=torch.float64, device=dev)   # last column: the ROI index (exact in float64)
    buf[:local_rows.shape[0], :P] = torch.as_tensor(np.ascontiguousarray(local_rows), dtype=torch.float64)
    buf[:local_rows.shape[0], P] = torch.as_tensor(np.asarray(local_idx, dtype=np.float64))
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    full = np.zeros((int(V), P))
    for r in range(world):
        blk = outs[r][:int(ns[r].item())].cpu().numpy()
        full[blk[:, P].astype(np.int64)] = blk[:, :P]
    return full


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
