"""CPU-only, world_size 2 over gloo: the sharding / gather layer used for patches, frames and ensemble
matches (parallel.py).  The same code runs over RCCL ("nccl") on the GPUs."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

par = importlib.import_module("3deecelltracker_amd.parallel")


def test_shard_range_partitions():
    for n in (0, 1, 7, 20, 75, 88):
        for world in (1, 2, 3, 4, 8):
            ranges = [par.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert [par.shard_range(75, r, 8)[1] - par.shard_range(75, r, 8)[0] for r in range(8)] == [10, 10, 10, 9, 9, 9, 9, 9]
    assert [len(par.shard_list(list(range(20)), r, 8)) for r in range(8)] == [3, 3, 3, 3, 2, 2, 2, 2]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert par.dist_info() == (rank, world)
        items = list(range(5))                                   # 5 ensemble matches over 2 ranks: 3 + 2

        def fn(i):
            return torch.full((4, 3), float(i), dtype=torch.float64) + torch.arange(3, dtype=torch.float64)
        stack = par.sharded_map_gather(fn, items)
        assert tuple(stack.shape) == (5, 4, 3)
        for i in items:
            assert torch.equal(stack[i], fn(i))
        # rank with an empty shard: with and without the tail_shape hint (the product callers at early time points of an
        # ensemble have world_size > len(items); round-1 code raised on the empty rank while the other blocked in all_gather)
        one = par.sharded_map_gather(fn, [7], tail_shape=(4, 3), dtype=torch.float64, device="cpu")
        assert tuple(one.shape) == (1, 4, 3) and torch.equal(one[0], fn(7))
        one = par.sharded_map_gather(fn, [7])
        assert tuple(one.shape) == (1, 4, 3) and one.dtype == torch.float64 and torch.equal(one[0], fn(7))
        f32 = par.sharded_map_gather(lambda i: torch.full((2,), float(i), dtype=torch.float32), [3], chains=4)
        assert tuple(f32.shape) == (1, 2) and f32.dtype == torch.float32 and float(f32[0, 0]) == 3.0
        none = par.sharded_map_gather(fn, [], tail_shape=(4, 3), dtype=torch.float64, device="cpu")
        assert tuple(none.shape) == (0, 4, 3)
        # the frames mode's gather: a batch of tracked sets per rank in ONE all_gather_into_tensor
        g = par.TrackedSetGather()
        kept = []
        for frames in (3, 1, 3):
            mine = [torch.full((5, 3), 10.0 * rank + f, dtype=torch.float64) for f in range(frames)]
            buf = g(mine)
            kept.append(buf)
            assert tuple(buf.shape) == (2, frames, 5, 3)
            for r in range(2):
                for f in range(frames):
                    assert torch.equal(buf[r, f], torch.full((5, 3), 10.0 * r + f, dtype=torch.float64))
        assert g.gathered == 2 * (3 + 1 + 3) and len(g.bufs) == 2          # buffers are kept per batch shape ...
        assert kept[0].data_ptr() != kept[2].data_ptr()                    # ... and rotate: the next call of the same shape leaves the last result alone
        again = g([torch.zeros((5, 3), dtype=torch.float64) for _ in range(3)])
        assert again.data_ptr() == kept[0].data_ptr() and torch.equal(kept[2][1 - rank, 1], torch.full((5, 3), 10.0 * (1 - rank) + 1, dtype=torch.float64))
        # variable-size centroid sets
        local = torch.arange((rank + 2) * 3, dtype=torch.float64).reshape(-1, 3) + 100 * rank
        sets = par.gather_centroids(local, cap=16)
        assert [s.shape[0] for s in sets] == [2, 3] and torch.equal(sets[rank], local)
        assert torch.equal(sets[1 - rank], torch.arange((1 - rank + 2) * 3, dtype=torch.float64).reshape(-1, 3) + 100 * (1 - rank))
        # slab gather of predict_volume_sharded: 5 units over 2 ranks (3 + 2), slabs padded to the maximum count
        total, per = 5, 4
        mx = -(-total // world)
        b, e = par.shard_range(total, rank, world)
        slab = torch.zeros((mx, per), dtype=torch.float32)
        slab[:e - b] = (torch.arange(b, e, dtype=torch.float32) + 1)[:, None]
        allslabs = torch.empty((world, mx, per), dtype=torch.float32)
        dist.all_gather_into_tensor(allslabs.view(-1), slab.view(-1))
        got = torch.cat([allslabs[r, :par.shard_range(total, r, world)[1] - par.shard_range(total, r, world)[0]] for r in range(world)])
        assert torch.equal(got, (torch.arange(total, dtype=torch.float32) + 1)[:, None].expand(total, per))
        q.put((rank, "ok"))
    except Exception as ex:   # pragma: no cover
        q.put((rank, repr(ex)))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_chain_map_keeps_order_and_falls_back_without_gpu():
    par = importlib.import_module("3deecelltracker_amd.parallel")
    assert par.chain_map(lambda v: v * v, [3, 1, 2], chains=3) == [9, 1, 4]
    assert par.chain_map(lambda v: v, [], chains=3) == []


# ------------------------------------------------------------------------------------ world_size 8 (one node of MI355X), SURVEY 8e's table
def _worker8(rank, world, port, q):
    """Every sharded path of SURVEY 8e at the node's real width, on CPU tensors over gloo: the shard arithmetic, the padded slab gather
    of the patch mode, the ensemble gather with uneven and EMPTY shards, the frames mode's TrackedSetGather with its buffer rotation, and the
    variable-size centroid gather.  What differs on the GPUs is the backend string only."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert par.dist_info() == (rank, world) and not par._solo(world)
        # ---- (1) patches of one 512x512x32 frame: 75 units -> 10/10/10/9/9/9/9/9, slabs padded to 10, ONE all_gather_into_tensor
        total, per = 75, 6                                          # (per = voxels of a centre crop; 6 stands in for 112*112*12)
        counts = [par.shard_range(total, r, world)[1] - par.shard_range(total, r, world)[0] for r in range(world)]
        assert counts == [10, 10, 10, 9, 9, 9, 9, 9] and sum(counts) == total
        mx = -(-total // world)
        b, e = par.shard_range(total, rank, world)
        assert e - b == counts[rank]
        crop = lambda p: torch.arange(per, dtype=torch.float32) + 1000.0 * p      # what patch p's centre crop holds
        slab = torch.zeros((mx, per), dtype=torch.float32)
        for i, p in enumerate(range(b, e)):
            slab[i] = crop(p)
        allslabs = torch.empty((world, mx, per), dtype=torch.float32)
        dist.all_gather_into_tensor(allslabs.view(-1), slab.view(-1))
        vol = torch.full((total, per), -1.0)
        for r in range(world):                                      # predict_volume_sharded's unpack loop (ct_tile_unpack_crops per source rank)
            rb, re_ = par.shard_range(total, r, world)
            vol[rb:re_] = allslabs[r, :re_ - rb]
        assert torch.equal(vol, torch.stack([crop(p) for p in range(total)]))
        assert all(torch.equal(allslabs[r, counts[r]:], torch.zeros((mx - counts[r], per))) for r in range(world))    # the pad rows stay zero

        # ---- (2) ensemble: 20 (t1 -> t2) matches -> 3/3/3/3/2/2/2/2, gathered in ITEM order, the trimmed mean identical on every rank
        members = list(range(20, 80, 3))                            # get_volumes_list(80, ...): [20, 23, ..., 77]
        assert len(members) == 20 and [len(par.shard_list(members, r, world)) for r in range(world)] == [3, 3, 3, 3, 2, 2, 2, 2]
        ran = []

        def predict(t1):
            ran.append(t1)
            g = torch.Generator().manual_seed(t1)
            return torch.rand((113, 3), generator=g, dtype=torch.float64) + t1
        preds = par.sharded_map_gather(predict, members)
        assert ran == par.shard_list(members, rank, world)          # a rank runs ITS members only
        assert tuple(preds.shape) == (20, 113, 3)
        want = torch.stack([torch.rand((113, 3), generator=torch.Generator().manual_seed(t1), dtype=torch.float64) + t1 for t1 in members])
        assert torch.equal(preds, want)
        from scipy.stats import trim_mean
        tm = torch.from_numpy(trim_mean(preds.numpy(), 0.1, axis=0))
        alltm = [torch.empty_like(tm) for _ in range(world)]
        dist.all_gather(alltm, tm)
        assert all(torch.equal(a, tm) for a in alltm)
        # the rank's whole share as one batched chain (batch_fn), same gather
        preds_b = par.sharded_map_gather(None, members, batch_fn=lambda its: [torch.full((4, 3), float(i), dtype=torch.float64) for i in its])
        assert tuple(preds_b.shape) == (20, 4, 3) and torch.equal(preds_b[:, 0, 0], torch.tensor(members, dtype=torch.float64))

        # ---- (3) fewer items than ranks: 3 over 8 (ranks 3..7 empty), 1 over 8, 0 over 8; with and without the shape hint
        fn = lambda i: torch.full((5, 3), float(i), dtype=torch.float64) + torch.arange(3, dtype=torch.float64)
        assert [len(par.shard_list([0, 1, 2], r, world)) for r in range(world)] == [1, 1, 1, 0, 0, 0, 0, 0]
        for hint in (dict(), dict(tail_shape=(5, 3), dtype=torch.float64, device="cpu")):
            three = par.sharded_map_gather(fn, [4, 5, 6], **hint)
            assert tuple(three.shape) == (3, 5, 3) and all(torch.equal(three[k], fn(4 + k)) for k in range(3))
            one = par.sharded_map_gather(fn, [9], **hint)
            assert tuple(one.shape) == (1, 5, 3) and torch.equal(one[0], fn(9))
        none = par.sharded_map_gather(fn, [], tail_shape=(5, 3), dtype=torch.float64, device="cpu")
        assert tuple(none.shape) == (0, 5, 3)
        f32 = par.sharded_map_gather(lambda i: torch.full((2,), float(i), dtype=torch.float32), [3, 4], chains=4)
        assert f32.dtype == torch.float32 and tuple(f32.shape) == (2, 2) and f32[:, 0].tolist() == [3.0, 4.0]

        # ---- (4) frames mode: every rank tracks its own frames, the corrected sets leave as one gather per batch; depth rotation
        for depth in (2, 3):
            g = par.TrackedSetGather(depth=depth)
            got = []
            for call in range(2 * depth + 1):
                mine = [torch.full((600, 3), 100.0 * call + 10.0 * rank + f, dtype=torch.float64) for f in range(4)]
                buf = g(mine)
                got.append(buf)
                assert tuple(buf.shape) == (world, 4, 600, 3)
                for r in range(world):
                    for f in range(4):
                        assert float(buf[r, f, 0, 0]) == 100.0 * call + 10.0 * r + f and float(buf[r, f, -1, -1]) == 100.0 * call + 10.0 * r + f
                if call >= 1:                                        # the buffer returned one call ago is still intact ...
                    assert float(got[call - 1][world - 1, 3, 0, 0]) == 100.0 * (call - 1) + 10.0 * (world - 1) + 3
                if call >= depth:                                    # ... and the one `depth` calls ago is the one just overwritten
                    assert got[call - depth].data_ptr() == buf.data_ptr()
            assert len(g.bufs) == 1 and len(next(iter(g.bufs.values()))) == depth
            assert g.gathered == world * 4 * (2 * depth + 1)
        assert par.TrackedSetGather()([]) is None                   # nothing tracked: no collective

        # ---- (5) centroid sets of different sizes per rank (rank 5 found nothing)
        n_local = 0 if rank == 5 else 590 + rank
        local = (torch.arange(n_local * 3, dtype=torch.float64).reshape(-1, 3) + 10000.0 * rank)
        sets = par.gather_centroids(local, cap=1024)
        assert [s.shape[0] for s in sets] == [0 if r == 5 else 590 + r for r in range(world)]
        for r in range(world):
            assert torch.equal(sets[r], torch.arange(sets[r].shape[0] * 3, dtype=torch.float64).reshape(-1, 3) + 10000.0 * r)
        try:
            par.gather_centroids(torch.zeros((1025, 3), dtype=torch.float64), cap=1024)
            raise AssertionError("a set larger than the gather capacity must raise")
        except ValueError:
            pass
        # ---- (6) all_gather_varlen directly, ragged incl. zero
        cnt = [r % 3 for r in range(world)]
        mine = torch.full((cnt[rank], 2), float(rank))
        cat = par.all_gather_varlen(mine, cnt)
        assert cat[:, 0].tolist() == [float(r) for r in range(world) for _ in range(cnt[r])]
        q.put((rank, "ok"))
    except Exception as ex:   # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


def test_world_size_8_gloo():
    """One node = 8 ranks: the only multi-GPU evidence obtainable without the hardware."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=300) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert sorted(res) == [(r, "ok") for r in range(world)], [r for r in res if r[1] != "ok"][:2]
