"""The whole per-frame chain on one HIP stream, TrackerLite dialect (what a user of the reference runs for every volume):

    raw uint16 stack                          (already in HBM, or uploaded from a pinned host buffer)
      -> ct_normalize_image                   preprocess._normalize_image            (preprocess.py:170-188)
      -> ct_unet_predict_volume               unet3d.unet3_prediction                (unet3d.py:203-256)
      -> ct_watershed_segment                 prob map -> regions -> centres: the reference's marker watershed (tracker.py:636-648,
                                              671-684 = watershed.py:16-108; seg/coords%06d.npy).  region_method="cc" selects the cheap
                                              variant instead (ct_segment_centroids: threshold + connected components)
      -> normalise / kNN features / FFN / greedy prior / PR-GLS / de-normalise       (trackerlite.py:70-109)
      -> ct_accurate_correction               CoordsToImageTransformer.accurate_correction (coord_image_transformer.py:406-489)

Only (n, 3) coordinate arrays and a few scalars (region count, convergence flags) visit the host.  `FrameChain.synthetic`
builds a self-consistent synthetic sequence (two consecutive frames of moving blobs, pass-through U-Net weights so that the
probability map has cell-like regions, the synthetic-trained FFN) for the benchmark's `chained` pass and scripts/microbench.py.
"""
from __future__ import annotations

import collections

import numpy as np

from . import _dev
from .coord_image_transformer import Coordinates, CoordsToImageTransformer
from .preprocess import normalize_image_device
from .segment import segment_centroids_device, watershed_centroids_device, watershed_centroids_enqueue
from .trackerlite import match_device


class FrameChain:
    def __init__(self, unet_model, ffn_model, transformer: CoordsToImageTransformer, noise_level: float, shrink=(24, 24, 2),
                 min_size: int = 20, beta: float = 3.0, lambda_: float = 3.0, ensemble: bool = True, region_method: str = "watershed",
                 prefetch_ref: bool = True):
        if region_method not in ("watershed", "cc"):
            raise ValueError(f"unknown region_method {region_method!r}: use 'watershed' or 'cc'")
        self.region_method = region_method
        self.prefetch_ref = bool(prefetch_ref)
        self.lcn_beside_unet = True                              # run_sequence: a frame's LCN on the watershed's stream (False: in front of its U-Net)
        # run_sequence: HIP priorities of the U-Net / watershed / match + correction streams (-1 high, 0 normal, 1 low); CT_SEQ_PRIO="s,w,t" overrides
        import os
        try:
            self.seq_priorities = tuple(int(v) for v in os.environ.get("CT_SEQ_PRIO", "0,-1,-1").split(","))
        except ValueError:
            self.seq_priorities = ()
        if len(self.seq_priorities) != 3 or any(v not in (-1, 0, 1) for v in self.seq_priorities):
            raise ValueError(f"CT_SEQ_PRIO must be three comma-separated stream priorities out of -1, 0, 1 (U-Net, watershed, match); got "
                             f"{os.environ.get('CT_SEQ_PRIO')!r}")
        # run_sequence: CUs reserved for the match + correction stream (the U-Net stream is masked off them; 0 = no reservation).  A prior that
        # needs many PR-GLS iterations makes that stream -- hundreds of dependent small launches, each waiting for a workgroup slot beside the
        # U-Net -- the loop's critical path; on CUs of its own they start at once.  Costs the U-Net its share of the chip on EVERY frame.
        self.seq_match_cus = int(os.environ.get("CT_SEQ_MATCH_CUS", "0"))
        self._side = None
        self._seq = None
        self.unet_model = unet_model
        self.ffn_model = ffn_model
        self.transformer = transformer
        self.noise_level = float(noise_level)
        self.shrink = tuple(shrink)
        self.min_size = int(min_size)
        self.beta, self.lambda_ = float(beta), float(lambda_)
        self.ensemble = ensemble
        self._events = None
        self._prob = None
        self.raw_t2 = None
        self.seg_real_t1 = None
        self.confirmed_real_t1 = None

    def normalized(self, raw_d):
        """raw stack -> the LCN-normalised volume the U-Net takes (asynchronous)."""
        return normalize_image_device(raw_d, self.noise_level, (27, 27, 1), mode=0, subtract_median=True)

    def probability_map(self, raw_d, out=None, norm=None):
        """raw stack -> prob fp32 [x,y,z] on the device (LCN + U-Net; asynchronous).  norm: the stack already normalised (normalized())."""
        if norm is None:
            norm = self.normalized(raw_d)
        self._mark("lcn")
        if out is None:
            if self._prob is None or self._prob.shape != norm.shape:
                self._prob = _dev.torch().empty_like(norm)
            out = self._prob
        prob = self.unet_model.predict_volume_device(norm, self.shrink, out=out)
        self._mark("unet")
        return prob

    def regions_enqueue(self, prob):
        """The watershed of `prob` enqueued on the current stream without its host round trip -> object whose .result()[1] are the centres."""
        vs = self.transformer.voxel_size
        return watershed_centroids_enqueue(prob, float(vs[2]) / float(vs[0]), "min_size", self.min_size, 0, want_labels=False)

    def regions(self, prob):
        """prob map -> centres fp64 [n,3] (voxel units) on the device."""
        if self.region_method == "watershed":                   # Tracker._segment's default (tracker.py:636-648): z_xy_ratio = voxel z / voxel x
            vs = self.transformer.voxel_size
            _, centres, _, _, _ = watershed_centroids_device(prob, float(vs[2]) / float(vs[0]), "min_size", self.min_size, 0, want_labels=False)
        else:
            _, centres, _ = segment_centroids_device(prob, 0.5, 1, self.min_size, want_labels=False)
        self._mark("regions")
        return centres

    def segment(self, raw_d):
        """raw stack -> (prob fp32 [x,y,z], centres fp64 [n,3] voxel units), both on the device."""
        prob = self.probability_map(raw_d)
        return prob, self.regions(prob)

    def run(self, raw_d=None, seg_real_t1=None, confirmed_real_t1=None):
        """One frame: segment `raw_d` (t2), match it against frame t1's segmentation, move t1's confirmed cells, correct them on
        t2's probability map.  Coordinates are real units (numpy or device fp64 [n, 3]).  Returns a dict with the corrected
        `Coordinates`, the number of segmented cells and the iteration counts."""
        t = _dev.torch()
        raw_d = self.raw_t2 if raw_d is None else raw_d
        seg_real_t1 = self.seg_real_t1 if seg_real_t1 is None else seg_real_t1
        confirmed_real_t1 = self.confirmed_real_t1 if confirmed_real_t1 is None else confirmed_real_t1
        self._mark(None)
        main = t.cuda.current_stream()
        conf_d = _dev.points_dev(confirmed_real_t1, raw_d.device)
        s1_d = _dev.points_dev(seg_real_t1, raw_d.device)
        if self.prefetch_ref:
            if self._side is None:
                self._side = t.cuda.Stream(device=raw_d.device)
                self._inputs_ready = t.cuda.Event()
            self._inputs_ready.record(main)                      # (the side stream must not wait for the U-Net: only for the two point sets)
        prob = self.probability_map(raw_d)
        # what the match needs of frame t1 alone (normalisation, the Gram matrix of its segmentation and that matrix's low-rank factor) is
        # enqueued on a second stream -- after the U-Net's launches, so that the host's enqueue time is hidden too -- and runs beside the
        # U-Net: ~0.35 ms of mostly single-workgroup work leave the frame's dependent chain
        if self.prefetch_ref:
            self._side.wait_event(self._inputs_ready)
            with t.cuda.stream(self._side):
                conf_n, para = _dev.normalize_points(conf_d)
                s1, _ = _dev.normalize_points(s1_d, apply_para=para)
                prepared = _dev.prgls_prepare_ref(s1, self.beta)
        if self.prefetch_ref:
            main.wait_stream(self._side)
            for x in (conf_d, s1_d, conf_n, para, s1):
                x.record_stream(main)
            return self.track(prob, s1_d, conf_d, pre=(conf_n, para, s1, prepared))
        return self.track(prob, s1_d, conf_d)

    def track(self, prob, seg_real_t1, confirmed_real_t1, pre=None, centres=None):
        """The part of a frame behind the U-Net, on the current stream: regions -> centres of `prob` (unless given), match against frame t1's
        segmentation, move t1's confirmed cells, correct them on `prob`.  `pre`: (normalised confirmed set, its parameters, normalised t1
        segmentation, PreparedRef) when they were made ahead."""
        t = _dev.torch()
        if centres is None:
            centres = self.regions(prob)
        vs = t.as_tensor(np.asarray(self.transformer.voxel_size, dtype=np.float64), device=centres.device)
        seg_real_t2 = centres * vs
        if pre is not None:
            conf_n, para, s1, prepared = pre
        else:
            conf_n, para = _dev.normalize_points(_dev.points_dev(confirmed_real_t1, centres.device))
            s1, _ = _dev.normalize_points(_dev.points_dev(seg_real_t1, centres.device), apply_para=para)
            prepared = None
        s2, _ = _dev.normalize_points(seg_real_t2, apply_para=para)
        _dev.check_match_sizes(s1.shape[0], s2.shape[0], 20, "FrameChain")
        tracked_n, iters = match_device(self.ffn_model, s1, s2, conf_n, self.beta, self.lambda_, prepared=prepared)
        tracked = _dev.denormalize_points(tracked_n, para)
        self._mark("match")
        coords = Coordinates(tracked.cpu().numpy(), self.transformer.interpolation_factor, self.transformer.voxel_size, dtype="real")
        corrected = self.transformer.accurate_correction(prob, coords, ensemble=self.ensemble)
        self._mark("correction")
        return {"coords": corrected, "n_segmented": int(centres.shape[0]), "prgls_iterations": int(iters),
                "correction_rounds": int(self.transformer.last_iterations), "seg_real_t2": seg_real_t2}

    def run_sequence(self, raws, seg_real_t0, confirmed_real_t0):
        """A sequence of frames as the reference's loop over volumes sees them (TrackerLite tracks from segmentations made beforehand,
        trackerlite.py:33-109: a volume's segmentation does not depend on the tracking of the volumes before it): every frame runs the whole
        chain with ITS OWN predecessor's results -- frame i is matched against frame i-1's segmentation and moves frame i-1's corrected
        cells.  Three streams, one host thread: the LCN + U-Net of frame i+2 (stream S) and the marker watershed of frame i+1 (stream W,
        enqueued without its host round trip) run beside the match + correction of frame i (stream T, whose steps the host synchronises
        with); three probability-map buffers.  `raws`: the stacks as the reference's loop finds them -- on the HOST (tracker.py:605-650 reads every volume
        from disk): numpy arrays or CPU tensors (pinned ones copy asynchronously), uploaded inside the loop on a copy stream of their own, frame i+3's
        upload beside everything else (a ring of four device buffers) -- or device uint16 / float stacks that are already resident.  Yields run()'s
        dict per frame; same values as calling run(raw_i, seg_{i-1}, corrected_{i-1}) frame after frame."""
        t = _dev.torch()
        raws = [t.from_numpy(np.ascontiguousarray(r)) if isinstance(r, np.ndarray) else r for r in raws]
        if not raws:
            return
        from . import _lib
        on_host = [not r.is_cuda for r in raws]
        if any(on_host):
            dev = next((r.device for r in raws if r.is_cuda), None)
            if dev is None:
                m = self.unet_model
                dev = t.device("cuda", m._device if getattr(m, "_device", None) is not None else t.cuda.current_device())
        else:
            dev = raws[0].device
        NB, NR = 3, 4
        key = (tuple(raws[0].shape), str(dev))
        if any(tuple(r.shape) != key[0] or (r.is_cuda and r.device != dev) for r in raws):
            raise ValueError("run_sequence: every volume of a sequence must have the same shape and live on the host or on the same device")
        if any(h and r.dtype != raws[0].dtype for h, r in zip(on_host, raws)):
            raise ValueError("run_sequence: host volumes of one sequence must share a dtype")
        if self._seq is None or self._seq["key"] != key:       # (streams and probability-map buffers belong to one volume shape on one device)
            self._release_seq_streams()
            ps, pw, pt = self.seq_priorities
            S_, T_ = t.cuda.Stream(device=dev, priority=ps), t.cuda.Stream(device=dev, priority=pt)
            if self.seq_match_cus > 0:                         # CU-masked streams (library-owned, kept for the chain's lifetime)
                import ctypes as C
                L = _lib.lib()
                n_cu = C.c_int(0)
                _lib.check(L.ct_device_info(dev.index or 0, C.byref(n_cu), None, None, 0), "ct_device_info")
                r = max(2, min(self.seq_match_cus, n_cu.value // 2))
                hs, ht = C.c_void_p(), C.c_void_p()
                _lib.check(L.ct_stream_create_cu_range(dev.index or 0, r, n_cu.value - r, C.byref(hs)), "ct_stream_create_cu_range")
                _lib.check(L.ct_stream_create_cu_range(dev.index or 0, 0, r, C.byref(ht)), "ct_stream_create_cu_range")
                S_, T_ = t.cuda.ExternalStream(hs.value, device=dev), t.cuda.ExternalStream(ht.value, device=dev)
                self._seq_handles = [hs, ht]
            self._seq = {"key": key, "S": S_, "W": t.cuda.Stream(device=dev, priority=pw),
                         "T": T_,
                         "prob": [t.empty(key[0], dtype=t.float32, device=dev) for _ in range(NB)],
                         "ready": [t.cuda.Event() for _ in range(NB)]}
        q = self._seq
        if any(on_host):
            hdt = next(r.dtype for h, r in zip(on_host, raws) if h)
            if q.get("raw_dtype") != hdt:                        # upload ring + copy stream: made once per (shape, device, dtype)
                q["raw"] = [t.empty(key[0], dtype=hdt, device=dev) for _ in range(NR)]
                q["raw_dtype"] = hdt
                q["C"] = t.cuda.Stream(device=dev)
            q["raw_free"] = [None] * NR
            q["raw_up"] = [None] * NR
        uploaded = [-1]
        q["free"] = [None] * NB
        q["spans"] = collections.deque(maxlen=3 * 512)           # (stream-local spans of the last sequence's last 512 frames: sequence_spans())
        S, W, T = q["S"], q["W"], q["T"]
        entry = t.cuda.current_stream(dev)
        for st in (S, W, T):
            st.wait_stream(entry)
        pending = {}

        def span(name, stream, fn):
            e0 = t.cuda.Event(enable_timing=True); e0.record(stream)
            r = fn()
            e1 = t.cuda.Event(enable_timing=True); e1.record(stream)
            q["spans"].append((name, e0, e1))
            return r, e1

        lcn_on_w = self.lcn_beside_unet and self.region_method == "watershed"

        def upload_through(k_last):
            # host stacks: frame k's copy goes out on the copy stream as soon as the ring slot's previous tenant (frame k - NR) has been normalised
            for k in range(uploaded[0] + 1, min(k_last, len(raws) - 1) + 1):
                uploaded[0] = k
                if not on_host[k]:
                    continue
                rb = k % NR
                Cs = q["C"]
                if q["raw_free"][rb] is not None:
                    Cs.wait_event(q["raw_free"][rb])
                else:
                    Cs.wait_stream(entry)
                with t.cuda.stream(Cs):
                    q["raw"][rb].copy_(raws[k], non_blocking=True)
                    q["raw_up"][rb] = t.cuda.Event(); q["raw_up"][rb].record(Cs)

        def raw_of(j, stream):
            """Frame j's stack on the device, `stream` ordered behind its upload."""
            if not on_host[j]:
                return raws[j]
            stream.wait_event(q["raw_up"][j % NR])
            return q["raw"][j % NR]

        def raw_done(j, stream):
            if on_host[j]:
                q["raw_free"][j % NR] = t.cuda.Event(); q["raw_free"][j % NR].record(stream)

        def enqueue_unet(j):
            b = j % NB
            upload_through(j + 1)                                # (frame j's own upload was queued one enqueue_unet ago: it runs a frame ahead)
            norm = None
            if lcn_on_w:
                # the LCN (bandwidth-bound sweeps) of frame j on the watershed's stream, which has ~3 ms of slack per frame, beside the
                # power-bound U-Net of the frame before: the U-Net stream is the pipeline's bottleneck and loses 0.25 ms per frame
                with t.cuda.stream(W):
                    norm = self.normalized(raw_of(j, W))
                    ev = t.cuda.Event(); ev.record(W)
                    raw_done(j, W)
                S.wait_event(ev)
                norm.record_stream(S)
            if q["free"][b] is not None:
                S.wait_event(q["free"][b])                       # frame j-3's correction has read this buffer
            with t.cuda.stream(S):
                span("unet", S, lambda: self.probability_map(None if norm is not None else raw_of(j, S), out=q["prob"][b], norm=norm))
                if norm is None:
                    raw_done(j, S)
                q["ready"][b].record(S)

        def enqueue_regions(j):
            if self.region_method != "watershed":
                return                                           # (the cheap variant has its round trip inside: it runs with the match)
            with t.cuda.stream(W):
                W.wait_event(q["ready"][j % NB])
                pending[j], _ = span("regions", W, lambda: self.regions_enqueue(q["prob"][j % NB]))

        seg_prev, conf_prev = seg_real_t0, confirmed_real_t0
        try:
            enqueue_unet(0)
            if len(raws) > 1:
                enqueue_unet(1)
            enqueue_regions(0)
            for i in range(len(raws)):
                if i + 2 < len(raws):
                    enqueue_unet(i + 2)                          # everything asynchronous is queued before this frame's host-synchronous steps
                if i + 1 < len(raws):
                    enqueue_regions(i + 1)
                b = i % NB
                with t.cuda.stream(T):
                    T.wait_event(q["ready"][b])
                    centres = pending.pop(i).result()[1] if i in pending else None
                    out, ev = span("match+correction", T, lambda: self.track(q["prob"][b], seg_prev, conf_prev, centres=centres))
                    q["free"][b] = ev
                seg_prev, conf_prev = out["seg_real_t2"], out["coords"].real
                yield out
        finally:
            # also when the consumer stops early or a frame raises (GeneratorExit / the exception passes through here): the U-Nets and
            # watersheds already enqueued keep writing the cached buffers, so the caller's stream is ordered behind all three streams
            for st in (S, W, T) + ((q["C"],) if any(on_host) else ()):
                entry.wait_stream(st)

    def _release_seq_streams(self):
        """The CU-masked streams of run_sequence are library-owned (ct_stream_create_cu_range): give them back when the sequence key changes
        or the chain goes away, after whatever they still hold has finished."""
        handles, self._seq_handles = getattr(self, "_seq_handles", []), []
        if handles:
            from . import _lib
            L = _lib.lib()
            if self._seq is not None:
                for k in ("S", "T"):
                    self._seq[k].synchronize()
            self._seq = None
            for h in handles:
                L.ct_stream_destroy(h)

    def close(self):
        self._release_seq_streams()

    def __del__(self):
        try:
            self._release_seq_streams()
        except Exception:
            pass

    def sequence_spans(self):
        """Mean ms of the LCN + U-Net spans (stream S) and of the regions/match/correction spans (stream T) of the last run_sequence."""
        _dev.torch().cuda.synchronize()
        acc = {}
        for name, e0, e1 in (self._seq or {}).get("spans", []):
            acc.setdefault(name, []).append(e0.elapsed_time(e1))
        return {k: sum(v) / len(v) for k, v in acc.items()}

    # ---- optional per-stage timing (HIP events on the current stream)
    def enable_timing(self, on=True):
        self._events = [] if on else None

    def _mark(self, name):
        if self._events is None:
            return
        t = _dev.torch()
        ev = t.cuda.Event(enable_timing=True); ev.record()
        self._events.append((name, ev))

    def stage_times(self):
        """ms per stage, averaged over the runs recorded since enable_timing()."""
        t = _dev.torch()
        t.cuda.synchronize()
        acc, cnt = {}, {}
        ev = self._events or []
        for (_, e0), (n1, e1) in zip(ev, ev[1:]):
            if n1 is None:
                continue
            acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1); cnt[n1] = cnt.get(n1, 0) + 1
        return {k: acc[k] / cnt[k] for k in acc}

    # ---- synthetic sequence
    @classmethod
    def synthetic(cls, shape=(512, 512, 32), n_cells=600, seed=0, ffn_weights=None, device=None, factor=5, region_method="watershed",
                  prefetch_ref=True):
        """Two consecutive synthetic frames: blobs at c1 (frame t1) and at c1 + smooth motion (frame t2)."""
        from pathlib import Path
        from . import synth, unet3d
        from .ffn import FFN
        t = _dev.torch()
        rng = np.random.default_rng(seed)
        stack1, c1 = synth.make_stack(shape, n_cells, seed)
        A = (rng.uniform(0, 1, (3, 3)) - 0.5) * 0.02
        ctr = np.asarray(shape) / 2.0
        c2 = c1 + (c1 - ctr) @ A + rng.normal(0, 0.3, c1.shape) * np.array([1, 1, 0.3]) + np.array([1.5, -1.0, 0.0])
        c2 = np.clip(c2, [6, 6, 1], [shape[0] - 6, shape[1] - 6, shape[2] - 1.5])
        stack2 = _render(shape, c2, rng)
        vs = np.array([1.0, 1.0, 4.0])
        subregions = []
        for c in c1:
            r = np.array([4, 4, max(2, int(1.5 * factor))])
            ci = np.array([c[0], c[1], c[2] * factor + factor // 2])
            lo = np.maximum(np.floor(ci - r).astype(int), 0); hi = np.minimum(np.ceil(ci + r).astype(int) + 1, (shape[0], shape[1], shape[2] * factor))
            g = np.meshgrid(*(np.arange(lo[a], hi[a]) for a in range(3)), indexing="ij")
            sub = sum(((g[a] - ci[a]) / r[a]) ** 2 for a in range(3)) <= 1.0
            subregions.append((tuple(slice(int(lo[a]), int(hi[a])) for a in range(3)), sub))
        vol1 = Coordinates(c1.astype(np.float32), factor, vs, "raw")
        tr = CoordsToImageTransformer(shape, vs, factor, subregions, vol1)
        model = unet3d.unet3_a(device=device).set_weights_dict(synth.make_passthrough_unet_weights("unet3_a", seed))
        if ffn_weights is None:
            ffn_weights = synth.load_trained_ffn()                  # package data (3deecelltracker_amd/data/)
        ffn = FFN(device=device).set_weights_dict(ffn_weights)
        chain = cls(model, ffn, tr, noise_level=100.0, region_method=region_method, prefetch_ref=prefetch_ref)
        dev = "cuda" if device is None else f"cuda:{device}"
        chain.raw_t1 = t.from_numpy(stack1).to(dev)
        chain.raw_t2 = t.from_numpy(stack2).to(dev)
        _, cen1 = chain.segment(chain.raw_t1)                       # frame t1's segmentation = what seg/coords%06d.npy would hold
        chain.seg_real_t1 = (cen1 * t.as_tensor(vs, device=cen1.device)).clone()
        chain.confirmed_real_t1 = vol1.real
        chain.true_t2 = c2
        return chain


def _render(shape, centres, rng):
    sx, sy, sz = shape
    img = rng.normal(100.0, 20.0, shape).astype(np.float32)
    np.clip(img, 0, None, out=img)
    amps = rng.uniform(400, 2000, len(centres))
    sig = np.array([3.0, 3.0, 1.0]); rad = np.array([9, 9, 3])
    for c, a in zip(centres, amps):
        c0 = np.maximum(np.floor(c - rad).astype(int), 0)
        c1 = np.minimum(np.ceil(c + rad).astype(int) + 1, shape)
        gx = np.exp(-0.5 * ((np.arange(c0[0], c1[0]) - c[0]) / sig[0]) ** 2)
        gy = np.exp(-0.5 * ((np.arange(c0[1], c1[1]) - c[1]) / sig[1]) ** 2)
        gz = np.exp(-0.5 * ((np.arange(c0[2], c1[2]) - c[2]) / sig[2]) ** 2)
        img[c0[0]:c1[0], c0[1]:c1[1], c0[2]:c1[2]] += a * gx[:, None, None] * gy[None, :, None] * gz[None, None, :]
    return np.clip(img, 0, 65535).astype(np.uint16)
