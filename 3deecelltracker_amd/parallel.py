"""Multi-GPU sharding of the path's independent units: one process per GPU,
torch.distributed ("nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path has no reduction across units -- U-Net patches of a frame, frames, and the <= 20
(t1 -> t2) matches of an ensemble prediction are mutually independent -- so the only collectives
are gathers of results: centre crops / probability volumes, centroid sets, ensemble predictions.
"""
from __future__ import annotations

import os
from typing import Callable, List, Sequence


def dist_info():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


def _solo(world: int) -> bool:
    """One rank: the sharded entry points take their single-process shortcuts.  CT_FORCE_COLLECTIVES=1 with a process group of ONE rank
    (tests): every collective is issued for real -- the way to execute the RCCL calls, with their stream and event ordering, on a one-GPU box."""
    if world != 1:
        return False
    if os.environ.get("CT_FORCE_COLLECTIVES") != "1":
        return True
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized())


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
    if _solo(world):
        return local
    mx = max(counts)
    tail = tuple(local.shape[1:])
    buf = torch.zeros((mx, *tail), dtype=local.dtype, device=local.device)
    if local.shape[0]:
        buf[:local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def chain_map(fn: Callable, items: Sequence, chains: int = 3):
    """[fn(it) for it in items] with up to `chains` items in flight, each on its own host thread and HIP stream.

    A PR-GLS match is a chain of dependent tiny kernels that leaves most of the GPU idle (37 ms alone, 13.5 ms per match
    with three chains in flight); the matches of an ensemble prediction are independent, so their chains
    interleave.  `fn` must be thread-safe (the TrackerLite / Tracker match paths are: all scratch is allocated per
    call, the models are read-only).  Results are identical to the sequential map."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    import torch
    items = list(items)
    if chains <= 1 or len(items) <= 1 or not torch.cuda.is_available():
        return [fn(it) for it in items]
    device = torch.cuda.current_device()
    n = min(int(chains), len(items))
    streams = [torch.cuda.Stream(device=device) for _ in range(n)]
    free = list(range(n)); lock = threading.Lock()
    ready = torch.cuda.Event(); ready.record()               # inputs produced on the caller's stream
    caller = torch.cuda.current_stream(device)

    def hand_over(out):
        """Results were allocated on a pooled side stream and are consumed on the caller's: tell the caching allocator, or a later
        chain on the same side stream may get the block while the caller's consumer (torch.stack, a collective) is still pending."""
        if isinstance(out, torch.Tensor):
            if out.is_cuda:
                out.record_stream(caller)
        elif isinstance(out, (list, tuple)):
            for o in out:
                hand_over(o)
        elif isinstance(out, dict):
            for o in out.values():
                hand_over(o)

    def run(it):
        with lock:
            idx = free.pop()
        try:
            torch.cuda.set_device(device)
            st = streams[idx]
            st.wait_event(ready)
            with torch.cuda.stream(st):
                out = fn(it)
            st.synchronize()
            hand_over(out)
            return out
        finally:
            with lock:
                free.append(idx)
    with ThreadPoolExecutor(max_workers=n) as pool:
        return list(pool.map(run, items))


_DTYPES = ("float64", "float32", "int32", "int64", "uint8", "float16")


def _comm_device(device=None):
    """Device collectives of the active backend expect: the rank's GPU for nccl (= RCCL), whatever is given (or CPU) for gloo."""
    import torch
    import torch.distributed as dist
    if device is not None:
        return torch.device(device)
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def _agree_on_tail(local, tail_shape, dtype, device):
    """Every rank learns the per-item shape / dtype of the gathered result, also ranks whose shard is empty (world_size >
    len(items): early time points of an ensemble, skipped volumes).  One tiny all_gather of int64 metadata; a rank with work
    contributes what it produced, a rank without adopts the first contribution (or the caller's tail_shape / dtype hint)."""
    import torch
    import torch.distributed as dist
    rank, world = dist_info()
    dev = _comm_device(device if local is None else local.device)
    meta = torch.zeros(8, dtype=torch.int64, device=dev)
    if local is not None:
        tail = tuple(local.shape[1:])
        if len(tail) > 5:
            raise ValueError("sharded_map_gather: results with more than 5 trailing dimensions are not supported")
        meta[0] = 1; meta[1] = len(tail)
        for i, d in enumerate(tail):
            meta[2 + i] = d
        meta[7] = _DTYPES.index(str(local.dtype).replace("torch.", ""))
    elif tail_shape is not None:
        meta[0] = 2; meta[1] = len(tail_shape)
        for i, d in enumerate(tail_shape):
            meta[2 + i] = d
        meta[7] = _DTYPES.index(str(dtype or torch.float64).replace("torch.", ""))
    allm = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(allm, meta)
    allm = torch.stack(allm).cpu()
    src = [r for r in range(world) if int(allm[r, 0]) == 1] or [r for r in range(world) if int(allm[r, 0]) == 2]
    if not src:
        return None, None, dev
    m = allm[src[0]]
    return tuple(int(v) for v in m[2:2 + int(m[1])]), getattr(torch, _DTYPES[int(m[7])]), dev


def sharded_map_gather(fn: Callable, items: Sequence, tail_shape=None, dtype=None, device=None, chains: int = 1, batch_fn=None):
    """Apply `fn(item) -> tensor` to this rank's share of `items` and all-gather the stacked results
    in item order: [len(items), ...] on every rank.  chains > 1: the rank's own items run `chains` at a time
    (chain_map).  Ranks whose share is empty take part in the collective with a zero-length contribution whose
    trailing shape is agreed on first (`tail_shape` / `dtype` are only hints for the case that *no* rank has work)."""
    import torch
    rank, world = dist_info()
    items = list(items)
    b, e = shard_range(len(items), rank, world)
    # batch_fn(list of items) -> list of tensors: the rank's whole share at once (e.g. one batched chain of launches)
    mine = (batch_fn(items[b:e]) if e > b else []) if batch_fn is not None else chain_map(fn, items[b:e], chains)
    local = torch.stack(mine) if mine else None
    if _solo(world):
        if local is None:
            if tail_shape is None:
                raise ValueError("no items and no tail_shape: cannot build an empty result")
            return torch.zeros((0, *tail_shape), dtype=dtype or torch.float64, device=_comm_device(device))
        return local
    tail, dt, dev = _agree_on_tail(local, tail_shape, dtype, device)
    if tail is None:
        raise ValueError("sharded_map_gather: no rank had work and no tail_shape was given")
    if local is None:
        local = torch.zeros((0, *tail), dtype=dt, device=dev)
    counts = [shard_range(len(items), r, world)[1] - shard_range(len(items), r, world)[0] for r in range(world)]
    return all_gather_varlen(local, counts)


def predict_volume_sharded(model, vol, shrink=(24, 24, 2), src: int | None = 0, comm_stream=None):
    """One frame's U-Net patches split over the ranks (BASELINE config 3; replaces the sequential patch loop of
    unet3d.py:246-254): the input volume is broadcast from rank `src` (None: every rank already holds it), every rank runs
    its contiguous patch range, packs the centre crops it produced into a dense slab (ct_tile_pack_crops, ~4.2 MB per rank at
    512x512x32 on 8 ranks) and ONE all_gather_into_tensor of the slabs (RCCL over xGMI) gives every rank all crops, which it
    places into its volume (ct_tile_unpack_crops).  No reduction, no zero-fill of a full volume.  Returns the full
    probability volume on every rank.  The collective runs on `comm_stream` (default: a side stream of the current one),
    ordered after the producing stream by an event and waited for before the crops are placed."""
    import torch
    import torch.distributed as dist
    from . import _lib
    from .unet3d import tile_plan
    rank, world = dist_info()
    centre, grid = tile_plan(tuple(vol.shape), model.arch.input_shape, shrink)
    total = grid[0] * grid[1] * grid[2]
    out = torch.empty_like(vol)
    if _solo(world):
        return model.predict_volume_device(vol, shrink, out=out)
    L = _lib.lib()
    cur = torch.cuda.current_stream(vol.device)
    if src is not None:
        dist.broadcast(vol, src=src)                    # one-time input distribution (async on the current stream for nccl)
    b, e = shard_range(total, rank, world)
    if e > b:
        model.predict_volume_device(vol, shrink, p_begin=b, n=e - b, out=out)
    per = centre[0] * centre[1] * centre[2]
    mx = -(-total // world)
    slab = torch.zeros((mx, per), dtype=torch.float32, device=vol.device)
    vs, ns, sh = _lib.ivec(vol.shape), _lib.ivec(model.arch.input_shape), _lib.ivec(shrink)
    if e > b:
        _lib.check(L.ct_tile_pack_crops(out.data_ptr(), vs, ns, sh, b, e - b, slab.data_ptr(), cur.cuda_stream), "ct_tile_pack_crops")
    allslabs = torch.empty((world, mx, per), dtype=torch.float32, device=vol.device)
    comm = comm_stream if comm_stream is not None else torch.cuda.Stream(device=vol.device)
    produced = torch.cuda.Event(); produced.record(cur)
    comm.wait_event(produced)
    with torch.cuda.stream(comm):
        dist.all_gather_into_tensor(allslabs.view(-1), slab.view(-1))
        gathered = torch.cuda.Event(); gathered.record(comm)
    cur.wait_event(gathered)
    allslabs.record_stream(comm); slab.record_stream(comm)
    for r in range(world):
        rb, re_ = shard_range(total, r, world)
        if r != rank and re_ > rb:
            _lib.check(L.ct_tile_unpack_crops(allslabs[r].data_ptr(), vs, ns, sh, rb, re_ - rb, out.data_ptr(), cur.cuda_stream),
                       "ct_tile_unpack_crops")
    return out


def gather_centroids(local_coords, cap: int = 4096):
    """All-gather per-frame centroid sets of different sizes: fixed-capacity (cap x 3 fp64 + count)
    buffers, one all_gather.  Returns a list of (n_r, 3) tensors, one per rank."""
    import torch
    import torch.distributed as dist
    rank, world = dist_info()
    if _solo(world):
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


class TrackedSetGather:
    """The "gather of centroid sets" of the frames mode: the tracked (cells, 3) sets of one match batch leave as ONE
    all_gather_into_tensor ([frames][cells][3] per rank -> [world][frames][cells][3]) on a communication stream ordered after the
    producer.  `gathered` counts the sets received (world x frames per call): a SCALE record shows that the collective saw every rank.
    CPU tensors + gloo in the tests, RCCL on the GPUs.

    Contract of the returned buffer: the CALLER's current stream already waits for the collective (an event recorded on the communication
    stream), so kernels enqueued after the call may read it; `last_event` is that event for consumers on other streams.  Receive buffers
    are kept per batch shape and used in rotation (`depth` of them, default 2): a returned buffer stays untouched until `depth` further
    calls of the same shape have been made.  Every rank must pass the same number of frames and the same cell count -- it is an
    all_gather_into_tensor; ragged tails go through gather_centroids."""

    def __init__(self, comm_stream=None, depth: int = 2):
        self.comm = comm_stream
        self.depth = max(1, int(depth))
        self.bufs = {}
        self.turn = {}
        self.gathered = 0
        self.last_event = None

    def __call__(self, tracked):
        import torch
        import torch.distributed as dist
        rank, world = dist_info()
        tracked = list(tracked)
        if _solo(world) or not tracked:
            return None
        cuda = tracked[0].is_cuda
        side = cuda and self.comm is not None
        ctx = torch.cuda.stream(self.comm) if side else _Null()
        cur = torch.cuda.current_stream(tracked[0].device) if cuda else None
        if side:
            self.comm.wait_stream(cur)
        with ctx:
            send = torch.stack(tracked)                      # [frames][cells][3], this rank's frames
            key = (len(tracked), tuple(send.shape[1:]), send.dtype)
            ring = self.bufs.setdefault(key, [])
            k = self.turn.get(key, 0)
            if len(ring) <= k:
                ring.append(torch.empty((world, *send.shape), dtype=send.dtype, device=send.device))
            buf = ring[k]
            self.turn[key] = (k + 1) % self.depth
            dist.all_gather_into_tensor(buf.view(-1, *send.shape[1:]), send)   # dim-0 concatenation: the form gloo accepts too
            if side:
                send.record_stream(self.comm)
                self.last_event = torch.cuda.Event(); self.last_event.record(self.comm)
        if side:
            cur.wait_event(self.last_event)                  # consumers on the caller's stream are ordered after the collective
        self.gathered += world * len(tracked)
        return buf


class _Null:
    def __enter__(self): return self
    def __exit__(self, *a): return False


class FramePipeline:
    """Intra-GPU overlap of the two halves of the path across frames.

    One PR-GLS match is a chain of ~10^3 dependent tiny kernels (latency-bound, ~5 us dispatch floor each), the
    U-Net a stream of chip-filling kernels.  On this GPU plain streams do not interleave them (the dispatcher
    drains the conv workgroups first), so the device is split with CU-masked streams: the U-Net gets
    `n_cu - match_cus` CUs, and `workers` host threads each drive the match chain of a *different* frame on their
    own stream inside the remaining `match_cus` CUs (ctypes releases the GIL while a chain runs).  Frames are
    independent units, so results are identical to processing them one after another.
    `priority=True` replaces the partition by stream priorities (the faster arrangement, see below; bench.py's default).
    """

    def __init__(self, device: int = 0, match_cus: int = 32, workers: int = 3, disjoint: bool = False, priority: bool = False):
        import ctypes as C
        from concurrent.futures import ThreadPoolExecutor
        import threading
        import torch
        from . import _lib
        L = _lib.lib()
        n_cu = C.c_int(0)
        _lib.check(L.ct_device_info(device, C.byref(n_cu), None, None, 0), "ct_device_info")
        self.device = device
        self.n_cu = n_cu.value
        self.match_cus = max(4, min(int(match_cus), self.n_cu // 2))
        self.workers = max(1, int(workers))
        self._handles = []
        _lib.check_hw_queues(f"FramePipeline with {self.workers} match workers")   # the multi-chain mode is where stream aliasing was measured

        def cu_stream(first, count):
            h = C.c_void_p()
            _lib.check(L.ct_stream_create_cu_range(device, first, count, C.byref(h)), "ct_stream_create_cu_range")
            self._handles.append(h)
            return torch.cuda.ExternalStream(h.value, device=f"cuda:{device}")
        if priority:
            # no CU partition: the U-Net on a normal-priority full-chip stream, the match chains on high-priority streams.  Every
            # tiny kernel of a chain then waits for a workgroup slot to drain (~10 us), and the U-Net keeps all 256 CUs: 134 vs 126
            # volumes/s for the benchmark's 364-iteration matches, 170 vs 116 for 10-iteration ones.  (Round 2 first measured 63 for
            # the former: the FFN scores were wrong whenever conv and match waves shared a SIMD - DESIGN.md section 5 - and the
            # matches ran 440-460 iterations on garbage priors.)
            self.seg_stream = torch.cuda.Stream(device=f"cuda:{device}", priority=0)
            self._match_streams = [torch.cuda.Stream(device=f"cuda:{device}", priority=-1) for _ in range(self.workers)]
            self.match_cus = 0
        else:
            self.seg_stream = cu_stream(self.match_cus, self.n_cu - self.match_cus)
        if priority:
            pass
        elif disjoint and self.match_cus >= 2 * self.workers:
            # every chain gets its own slice of the match partition (streams sharing one CU mask were observed to
            # advance in lock-step, i.e. serialised)
            per = self.match_cus // self.workers
            self._match_streams = [cu_stream(i * per, per if i < self.workers - 1 else self.match_cus - i * per)
                                   for i in range(self.workers)]
        else:
            self._match_streams = [cu_stream(0, self.match_cus) for _ in range(self.workers)]
        # pre-processing of the NEXT frame (LCN: HBM-bound, 0.3 ms) can ride on the match partition while the U-Net of the current
        # frame owns the big one: `with torch.cuda.stream(pipe.prep_stream)` + an event the seg stream waits on
        # (created on first use: every extra stream takes a hardware queue, and an idle fifth one measurably slows the others)
        self._prep_stream = None
        self._make_prep = (lambda: torch.cuda.Stream(device=f"cuda:{device}", priority=-1)) if priority else (lambda: cu_stream(0, self.match_cus))
        self._tls = threading.local()
        self._free = list(range(self.workers))
        self._lock = threading.Lock()
        self._pool = ThreadPoolExecutor(max_workers=self.workers)
        self._inflight = []

    @property
    def prep_stream(self):
        if self._prep_stream is None:
            self._prep_stream = self._make_prep()
        return self._prep_stream

    def _run(self, fn, args):
        import torch
        with self._lock:
            idx = self._free.pop()
        try:
            torch.cuda.set_device(self.device)
            s = self._match_streams[idx]
            with torch.cuda.stream(s):
                out = fn(*args)
            s.synchronize()
            return out
        finally:
            with self._lock:
                self._free.append(idx)

    def submit_match(self, fn, *args):
        """Run `fn(*args)` (a device match, e.g. trackerlite.match_device) on a worker; at most `workers` in flight."""
        while len(self._inflight) >= self.workers:
            self._inflight.pop(0).result()
        fut = self._pool.submit(self._run, fn, args)
        self._inflight.append(fut)
        return fut

    def drain(self):
        for f in self._inflight:
            f.result()
        self._inflight.clear()

    def close(self):
        self.drain()
        self._pool.shutdown(wait=True)
