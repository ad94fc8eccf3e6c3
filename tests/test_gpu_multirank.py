"""Multi-rank product paths on one GPU box: 2 ranks over gloo sharing cuda:0 (RCCL refuses two ranks per device; the
collectives used -- all_gather / all_reduce of device tensors -- are backend-agnostic).  Sharded results must equal the
single-process ones bit-for-bit."""
import importlib
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, tmp, q):
    try:
        sys.path.insert(0, str(REPO))
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        synth = importlib.import_module("3deecelltracker_amd.synth")
        unet3d = importlib.import_module("3deecelltracker_amd.unet3d")
        par = importlib.import_module("3deecelltracker_amd.parallel")
        tl = importlib.import_module("3deecelltracker_amd.trackerlite")
        cit = importlib.import_module("3deecelltracker_amd.coord_image_transformer")
        # --- patches of one frame sharded over the ranks (BASELINE config 2 pattern, small volume: 18 patches)
        model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", 0))
        vol = torch.from_numpy(np.random.default_rng(0).normal(size=(256, 256, 24)).astype(np.float32)).cuda()
        sharded = par.predict_volume_sharded(model, vol)
        whole = model.predict_volume_device(vol)
        ok_vol = bool(torch.equal(sharded, whole))
        # --- ensemble prediction sharded over the ranks (BASELINE config 3 pattern)
        vs = np.array([1.0, 1.0, 4.0])
        proof = cit.Coordinates(np.load(Path(tmp) / "seg" / "coords000001.npy"), 4, vs, "raw")
        trk = tl.TrackerLite(tmp, "synthetic", proof, basedir=str(Path(tmp) / "ffn_models"))
        ens = trk.predict_cell_positions_ensemble([], 6, proof, beta=3, lambda_=3, sampling_number=20).real
        q.put((rank, ok_vol, ens))
        dist.destroy_process_group()
    except Exception as ex:      # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))


def test_two_ranks_match_single_process(tmp_path):
    import torch
    import torch.multiprocessing as mp
    synth = importlib.import_module("3deecelltracker_amd.synth")
    ffn_mod = importlib.import_module("3deecelltracker_amd.ffn")
    tl = importlib.import_module("3deecelltracker_amd.trackerlite")
    cit = importlib.import_module("3deecelltracker_amd.coord_image_transformer")
    vs = np.array([1.0, 1.0, 4.0])
    (tmp_path / "seg").mkdir(); (tmp_path / "ffn_models").mkdir()
    ffn_mod.FFN().set_weights_dict(synth.make_ffn_weights(0, 6.0, -3.0)).save_weights(tmp_path / "ffn_models" / "synthetic.npz")
    rng = np.random.default_rng(5)
    base = rng.uniform(0, 1, (60, 3)) * np.array([168, 401, 32])
    (tmp_path / "track_results" / "coords_real").mkdir(parents=True)
    for t in range(1, 7):
        c = (base + rng.normal(0, 0.4, base.shape))[rng.permutation(60)].astype(np.float32)
        np.save(tmp_path / "seg" / f"coords{str(t).zfill(6)}.npy", c)
        np.save(tmp_path / "track_results" / "coords_real" / f"coords{str(t).zfill(6)}.npy", cit.Coordinates(c, 4, vs, "raw").real)
    proof = cit.Coordinates(np.load(tmp_path / "seg" / "coords000001.npy"), 4, vs, "raw")
    single = tl.TrackerLite(str(tmp_path), "synthetic", proof, basedir=str(tmp_path / "ffn_models")) \
        .predict_cell_positions_ensemble([], 6, proof, beta=3, lambda_=3, sampling_number=20).real
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
    for rank, ok_vol, ens in res:
        assert ok_vol is True, ens
        assert np.array_equal(ens, single), f"rank {rank}: sharded ensemble differs from the single-process result"


RCCL_CHILD = r"""
import importlib, os, sys
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[2]
os.environ["CT_FORCE_COLLECTIVES"] = "1"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
synth = importlib.import_module("3deecelltracker_amd.synth")
unet3d = importlib.import_module("3deecelltracker_amd.unet3d")
par = importlib.import_module("3deecelltracker_amd.parallel")
model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", 0))
vol = torch.from_numpy(np.random.default_rng(0).normal(size=(256, 256, 24)).astype(np.float32)).cuda()
whole = model.predict_volume_device(vol).clone()
# broadcast + patch range + pack + all_gather_into_tensor on a side stream + (nothing to unpack from other ranks)
sharded = par.predict_volume_sharded(model, vol)
torch.cuda.synchronize()
assert torch.equal(sharded, whole), "predict_volume_sharded over a one-rank RCCL group"
# the frames mode's gather of tracked sets on a communication stream, buffers in rotation
comm = torch.cuda.Stream()
g = par.TrackedSetGather(comm)
rng = np.random.default_rng(1)
for it in range(5):
    sets = [torch.from_numpy(rng.normal(size=(600, 3))).cuda() for _ in range(8)]
    buf = g(sets)
    got = buf.clone()                       # (ordered after the collective on the current stream by the gatherer's event)
    torch.cuda.synchronize()
    assert got.shape == (1, 8, 600, 3) and torch.equal(got[0], torch.stack(sets)), f"TrackedSetGather round {it}"
assert g.gathered == 40
# ragged centroid sets, sharded map, the bench's max-over-ranks reduction and barrier
c = torch.from_numpy(rng.normal(size=(571, 3))).cuda()
out = par.gather_centroids(c)
assert len(out) == 1 and torch.equal(out[0], c)
res = par.sharded_map_gather(lambda k: torch.full((4, 3), float(k), dtype=torch.float64, device="cuda"), list(range(7)))
assert res.shape == (7, 4, 3) and torch.equal(res[:, 0, 0].cpu(), torch.arange(7, dtype=torch.float64))
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
assert float(t.item()) == 1.25
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_one_rank_rccl_group_runs_every_collective_of_the_sharded_paths():
    """RCCL refuses two ranks on one device, so on a one-GPU box the nccl backend is exercised with a process group of ONE rank and
    CT_FORCE_COLLECTIVES=1: broadcast, all_gather, all_gather_into_tensor (on side streams, with the product's event ordering), all_reduce and
    barrier really go through RCCL kernels; results must equal the single-process ones."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", RCCL_CHILD, str(REPO), str(_free_port())], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
