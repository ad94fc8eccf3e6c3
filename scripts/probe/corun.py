"""What the U-Net loses beside the other streams of the frame loop, part by part (round-4 verdict, "weak" 5).

The loop of frame.FrameChain.run_sequence with switches: the U-Net of frame i+2 on stream S, [the LCN of frame i+2 on W], [the marker watershed of
frame i+1 on W], [match + correction of frame i on T].  Every co-runner works on FIXED inputs (a probability map and point sets made beforehand) so
that a subset can run without the others; pacing is the real loop's (W and T wait for the U-Net events).  Per configuration: ms per frame of the
whole loop, the U-Net span on its own stream (HIP events), and package power / shader clock sampled through amdsmi every 10 ms.

    python scripts/probe/corun.py [--frames 48] [--mask-cus N]      (--mask-cus: W and T on a CU-masked stream of the first N CUs, S on the rest)
    python scripts/probe/corun.py --delays                           (the watershed / the match started D ms after the U-Net they run beside: does it
                                                                      matter WHICH conv layers a co-runner overlaps?  a spinning one-wave kernel is the delay)
"""
import importlib
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402
import torch  # noqa: E402

frame = importlib.import_module("3deecelltracker_amd.frame")
_lib = importlib.import_module("3deecelltracker_amd._lib")


class Sampler:
    """package power (W) and shader clock (MHz) every 10 ms in a thread (amdsmi; silent when it is missing)."""

    def __init__(self):
        self.rows, self._stop, self.ok = [], threading.Event(), False
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self.h = amdsmi.amdsmi_get_processor_handles()[0]
            self.smi = amdsmi
            self.ok = True
        except Exception as e:  # noqa: BLE001
            print("amdsmi unavailable:", repr(e)[:200])

    def _read(self):
        s = self.smi
        p = c = None
        try:
            pi = s.amdsmi_get_power_info(self.h)
            p = pi.get("current_socket_power") or pi.get("average_socket_power")
        except Exception:  # noqa: BLE001
            pass
        try:
            ci = s.amdsmi_get_clock_info(self.h, s.AmdSmiClkType.GFX)
            c = ci.get("clk") or ci.get("cur_clk")
        except Exception:  # noqa: BLE001
            pass
        return p, c

    def __enter__(self):
        self.rows = []
        self._stop.clear()
        if self.ok:
            def loop():
                while not self._stop.is_set():
                    self.rows.append(self._read()); time.sleep(0.01)
            self.th = threading.Thread(target=loop, daemon=True); self.th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self.ok:
            self.th.join()

    def summary(self):
        def med(i):
            v = [r[i] for r in self.rows if isinstance(r[i], (int, float))]
            return float(np.median(v)) if v else None
        return {"power_W": med(0), "sclk_MHz": med(1), "samples": len(self.rows)}


def main():
    frames = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 48
    mask = int(sys.argv[sys.argv.index("--mask-cus") + 1]) if "--mask-cus" in sys.argv else 0
    chain = frame.FrameChain.synthetic(shape=(512, 512, 32), n_cells=600, seed=0)
    dev = chain.raw_t2.device
    t = torch
    raw = chain.raw_t2
    norm = chain.normalized(raw)
    prob_fixed = chain.probability_map(raw, out=t.empty(tuple(raw.shape), dtype=t.float32, device=dev)).clone()
    centres_fixed = chain.regions(prob_fixed).clone()
    seg1, conf1 = chain.seg_real_t1, chain.confirmed_real_t1
    t.cuda.synchronize()
    if mask:
        import ctypes as C
        L = _lib.lib()
        n_cu = C.c_int(0)
        _lib.check(L.ct_device_info(0, C.byref(n_cu), None, None, 0), "ct_device_info")

        def cu_stream(first, count):
            h = C.c_void_p()
            _lib.check(L.ct_stream_create_cu_range(0, first, count, C.byref(h)), "ct_stream_create_cu_range")
            return t.cuda.ExternalStream(h.value, device=dev)
        S = cu_stream(mask, n_cu.value - mask) if "--mask-unet" in sys.argv else t.cuda.Stream(device=dev)
        W, T = cu_stream(0, mask), cu_stream(0, mask)
    else:
        S, W, T = t.cuda.Stream(device=dev), t.cuda.Stream(device=dev, priority=-1), t.cuda.Stream(device=dev, priority=-1)
    probs = [t.empty_like(prob_fixed) for _ in range(3)]
    smp = Sampler()

    # device-side delay: torch.cuda._sleep spins one wave for a number of counter ticks; calibrated here
    def sleep_ms_per_tick():
        e0 = t.cuda.Event(enable_timing=True); e1 = t.cuda.Event(enable_timing=True)
        t.cuda._sleep(1000); t.cuda.synchronize()
        e0.record(); t.cuda._sleep(2_000_000); e1.record(); t.cuda.synchronize()
        return e0.elapsed_time(e1) / 2_000_000
    tick_ms = sleep_ms_per_tick()
    delay = {"ws": 0.0, "match": 0.0}

    def spin(ms):
        if ms > 0:
            t.cuda._sleep(int(ms / tick_ms))

    def loop(parts, n):
        ev_u = [None] * n
        ev_s = [None] * n                                    # start of U-Net j on S (what a delayed co-runner counts from)
        pend = {}

        def enq_unet(j):
            if "lcn" in parts:
                with t.cuda.stream(W):
                    chain.normalized(raw)
            with t.cuda.stream(S):
                ev_s[j] = t.cuda.Event(); ev_s[j].record(S)
                if "unet" in parts:
                    chain.unet_model.predict_volume_device(norm, chain.shrink, out=probs[j % 3])
                ev_u[j] = t.cuda.Event(); ev_u[j].record(S)

        def enq_ws(j):
            if "ws" in parts:
                with t.cuda.stream(W):
                    W.wait_event(ev_u[j])
                    if delay["ws"] > 0 and j + 1 < n and ev_s[j + 1] is not None:
                        W.wait_event(ev_s[j + 1]); spin(delay["ws"])
                    pend[j] = chain.regions_enqueue(prob_fixed)
        e0 = t.cuda.Event(enable_timing=True); e1 = t.cuda.Event(enable_timing=True)
        t.cuda.synchronize(); t0 = time.perf_counter()
        e0.record(S)
        enq_unet(0); enq_unet(1); enq_ws(0)
        for i in range(n):
            if i + 2 < n:
                enq_unet(i + 2)
            if i + 1 < n:
                enq_ws(i + 1)
            if "match" in parts:
                with t.cuda.stream(T):
                    T.wait_event(ev_u[i])
                    if delay["match"] > 0 and i + 1 < n and ev_s[i + 1] is not None:
                        T.wait_event(ev_s[i + 1]); spin(delay["match"])
                    if i in pend:
                        pend.pop(i).result()
                    chain.track(prob_fixed, seg1, conf1, centres=centres_fixed)
            else:
                ev_u[i].synchronize()
                if i in pend:
                    pend.pop(i).result()
        e1.record(S)
        t.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        return dt * 1e3, e0.elapsed_time(e1) / n

    if "--delays" in sys.argv:
        print(f"device-side delay: {tick_ms * 1e6:.3f} ns per tick; frames per configuration {frames}")
        for parts, key in ((("unet", "ws"), "ws"), (("unet", "match"), "match"), (("unet", "lcn", "ws", "match"), "ws"), (("unet", "lcn", "ws", "match"), "match")):
            for d in (0.0, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0):
                delay["ws"] = delay["match"] = 0.0
                delay[key] = d
                loop(parts, 6)
                ms, span = loop(parts, frames)
                print(f"{'+'.join(parts):22s} {key} delayed by {d:3.1f} ms after the U-Net's start: loop {ms:6.2f} ms/frame   S-stream span {span:6.2f} ms/frame", flush=True)
        return
    configs = [("unet",), ("unet", "lcn"), ("unet", "ws"), ("unet", "lcn", "ws"), ("unet", "match"), ("unet", "lcn", "ws", "match"),
               ("lcn", "ws", "match"), ("ws",), ("match",), ("lcn",)]
    if "--only" in sys.argv:
        want = [tuple(c.split("+")) for c in sys.argv[sys.argv.index("--only") + 1].split(",")]
        configs = [c for c in configs if c in want] + [c for c in want if c not in configs]
    print(f"frames per configuration {frames}; mask {mask} CUs for W/T{' (U-Net on the rest)' if '--mask-unet' in sys.argv else ''}")
    for parts in configs:
        loop(parts, 6)
        with smp:
            ms, span = loop(parts, frames)
        s = smp.summary()
        print(f"{'+'.join(parts):22s} loop {ms:6.2f} ms/frame   S-stream span {span:6.2f} ms/frame   power {s['power_W']} W  sclk {s['sclk_MHz']} MHz ({s['samples']} samples)",
              flush=True)


if __name__ == "__main__":
    main()
