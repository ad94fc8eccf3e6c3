"""Multi-GPU sharding of the path's independent units: one process per GPU,
torch.distributed ("nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path has no reduction across units -- U-Net patches of a frame, frames, and the <= 20
(t1 -> t2) matches of an ensemble prediction are mutually independent -- so the only collectives
are gathers of results: centre crops / probability volumes, centroid sets, ensemble predictions.
"""
from __future__ import annotations

from typing import Callable, List, Sequence


def dist_info():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block partition: [begin, end) of `n_items` owned by `rank` (sizes differ by <= 1)."""
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_list(items: Sequence, rank: int, world: int) -> List:
    b, e = shard_range(len(items), rank, world)
    return list(items[b:e])


def all_gather_varlen(local, counts: Sequence[int]):
    """All-gather tensors whose leading dimension differs per rank (counts known from shard_range).
    Pads to the maximum count so a single all_gather (one RCCL call) suffices."""
    import torch
    import torch.distributed as dist
    rank, world = dist_info()
    if world == 1:
        return local
    mx = max(counts)
    tail = tuple(local.shape[1:])
    buf = torch.zeros((mx, *tail), dtype=local.dtype, device=local.device)
    if local.shape[0]:
        buf[:local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def sharded_map_gather(fn: Callable, items: Sequence, tail_shape=None, dtype=None, device=None):
    """Apply `fn(item) -> tensor` to this rank's share of `items` and all-gather the stacked results
    in item order: [len(items), ...] on every rank."""
    import torch
    rank, world = dist_info()
    b, e = shard_range(len(items), rank, world)
    mine = [fn(it) for it in items[b:e]]
    if mine:
        local = torch.stack(mine)
    else:
        if tail_shape is None:
            raise ValueError("a rank without work needs tail_shape/dtype/device to build its empty contribution")
        local = torch.zeros((0, *tail_shape), dtype=dtype, device=device)
    if world == 1:
        return local
    counts = [shard_range(len(items), r, world)[1] - shard_range(len(items), r, world)[0] for r in range(world)]
    if not mine and tail_shape is None:
        raise ValueError("empty shard")
    return all_gather_varlen(local, counts)


def predict_volume_sharded(model, vol, shrink=(24, 24, 2)):
    """One frame's U-Net patches split over the ranks (BASELINE config 3): every rank runs its
    contiguous patch range and the per-rank partial volumes (disjoint centre crops, zeros
    elsewhere) are summed with one all-reduce... which is exactly a gather because the supports
    are disjoint.  Returns the full probability volume on every rank."""
    import torch
    import torch.distributed as dist
    from .unet3d import tile_plan
    rank, world = dist_info()
    centre, grid = tile_plan(tuple(vol.shape), model.arch.input_shape, shrink)
    total = grid[0] * grid[1] * grid[2]
    b, e = shard_range(total, rank, world)
    out = torch.zeros_like(vol)
    if e > b:
        model.predict_volume_device(vol, shrink, p_begin=b, n=e - b, out=out)
    if world > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
    return out


def gather_centroids(local_coords, cap: int = 4096):
    """All-gather per-frame centroid sets of different sizes: fixed-capacity (cap x 3 fp64 + count)
    buffers, one all_gather.  Returns a list of (n_r, 3) tensors, one per rank."""
    import torch
    import torch.distributed as dist
    rank, world = dist_info()
    if world == 1:
        return [local_coords]
    n = local_coords.shape[0]
    if n > cap:
        raise ValueError(f"{n} centroids exceed the gather capacity {cap}")
    buf = torch.zeros((cap + 1, 3), dtype=torch.float64, device=local_coords.device)
    buf[:n] = local_coords
    buf[cap, 0] = float(n)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return [o[:int(o[cap, 0].item())] for o in out]
