"""The frame loop with the U-Net of a frame as TWO patch ranges on two streams (every layer's tail under the other half's body) against the
shipped single launch sequence.  Alone the split recovers ~0.1 ms of 5.5 (scripts/probe/two_stream_unet.py); does the loop's U-Net span shrink?
    python scripts/probe/seq_split_unet.py [frames]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

frame = importlib.import_module("3deecelltracker_amd.frame")
unet3d = importlib.import_module("3deecelltracker_amd.unet3d")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64


class SplitUnet:
    """predict_volume_device as `parts` patch ranges: range 0 on the caller's stream, the others on helper streams forked from / joined into it"""

    def __init__(self, base, weights, parts):
        self.models = [base] + [unet3d.unet3_a().set_weights_dict(weights) for _ in range(parts - 1)]
        self.helpers = [torch.cuda.Stream() for _ in range(parts - 1)]
        self.parts = parts
        self.arch = base.arch

    def __getattr__(self, name):
        return getattr(self.models[0], name)

    def predict_volume_device(self, vol, shrink=(24, 24, 2), out=None, **kw):
        _, grid = unet3d.tile_plan(tuple(vol.shape), self.arch.input_shape, shrink)
        total = grid[0] * grid[1] * grid[2]
        bounds = [round(total * k / self.parts) for k in range(self.parts + 1)]
        if out is None:
            out = torch.zeros_like(vol)
        cur = torch.cuda.current_stream()
        ev = cur.record_event()
        for k in range(1, self.parts):
            st = self.helpers[k - 1]
            st.wait_event(ev)
            with torch.cuda.stream(st):
                self.models[k].predict_volume_device(vol, shrink, p_begin=bounds[k], n=bounds[k + 1] - bounds[k], out=out)
        self.models[0].predict_volume_device(vol, shrink, p_begin=0, n=bounds[1], out=out)
        for st in self.helpers:
            cur.wait_event(st.record_event())
        return out


def run(chain, label):
    raws = [chain.raw_t2, chain.raw_t1] * (n // 2)
    outs0 = list(chain.run_sequence(raws[:4], chain.seg_real_t1, chain.confirmed_real_t1))
    res = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = list(chain.run_sequence(raws, chain.seg_real_t1, chain.confirmed_real_t1))
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / len(raws) * 1e3)
    print(f"{label}: {' '.join(f'{r:.2f}' for r in res)} ms per frame", {k: round(v, 2) for k, v in chain.sequence_spans().items()}, flush=True)
    return outs0[-1]["coords"].real


chain = frame.FrameChain.synthetic(shape=(512, 512, 32), n_cells=600, seed=0)
base = chain.unet_model
ref = run(chain, "one launch sequence")
synth = importlib.import_module("3deecelltracker_amd.synth")
w = synth.make_passthrough_unet_weights("unet3_a", 0)
for parts in (2, 3):
    chain.unet_model = SplitUnet(base, w, parts)
    got = run(chain, f"{parts} patch ranges on {parts} streams")
    print("   same corrected coordinates:", bool((got == ref).all()))
chain.unet_model = base
run(chain, "one launch sequence (again)")
