"""bench.py contract and its sharding modes on the GPU box (1 GPU: 2 ranks over gloo share cuda:0; the nccl path runs when the
box has >= 2 devices)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline", "cpu_baseline", "value_spread"}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(cmd, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_modes_share_one_schema():
    base = [sys.executable, "bench.py", "--gpus", "1", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-realistic-pass"]
    frames = _run(base + ["--windows", "3"])
    patches = _run(base + ["--mode", "patches"])
    indep = _run(base + ["--mode", "independent"])
    for line in (frames, patches, indep):
        assert KEYS <= set(line) and line["n_gpus"] == 1 and line["steps"] == 8 and line["unit"] == "volumes/s"
        assert line["metric"].startswith("volumes/s segment+match") and "workload" in line["config"] and "model" not in line["config"]
        r = line["roofline"]
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r) and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert r["traffic"] is None or isinstance(r["traffic"], (int, float))          # bytes per launch (PMC), the contract's scalar
        # the block explains itself: the SURVEY 8(d) contract figure, the measured sustained matrix-pipe rate beside the datasheet peak
        assert 0.0 < r["hbm_contract_frac"] < 1.0 and r["hbm_contract_frac"] == r["conv_stack_hbm_frac"]
        assert r["sustained_peak"] < r["peak"] and abs(r["frac_of_sustained_peak"] - r["achieved"] / r["sustained_peak"]) < 1e-3
        assert any("pipe_busy" in ly for ly in line["layers"])                            # SQ digest merged per layer
        assert r["hbm_bound_kernel"]["kernel"].startswith(("conv_first_f16_kernel", "conv_l0l1_fused_kernel"))   # (the first conv runs inside the second's workgroups)
        assert 0 < r["algorithmic_frac"] <= r["frac"] and 0 < r["conv_stack_hbm_frac"] < 1
        assert line["config"]["rccl_ranks"]["world_size"] == 1
    # the contract line is the REAL frame: nothing excluded, every frame chained on its predecessor, several timed windows
    cfg = frames["config"]
    assert cfg["headline_excludes"] == [] and cfg["frames_timed"] == 8 and cfg["cells_segmented"][0] > 400 and cfg["prgls_iterations"] <= 30
    assert {"unet", "regions", "match+correction"} <= set(cfg["stream_spans_ms"])
    sp = frames["value_spread"]
    assert sp["windows"] == 3 and sp["min"] <= sp["median"] <= sp["max"] and sp["min"] <= frames["value"] <= sp["max"]
    assert abs(frames["value"] - 1e3 / frames["ms_per_step"]) < 0.01 * frames["value"]
    # the lines of rounds 1-4 still say what they leave out
    assert patches["config"]["headline_excludes"] and indep["config"]["headline_excludes"] and patches["value_spread"] is None
    assert frames["scaling"] == "weak" and indep["scaling"] == "weak" and patches["scaling"] == "strong"
    # at N = 1 the patches and independent modes do the same work: same frame rate within noise
    assert 0.5 < patches["value"] / indep["value"] < 2.0, (indep["value"], patches["value"])      # 8-step runs that end on a 364-iteration match chain: generous band
    ens = _run(base + ["--mode", "ensemble"])
    assert ens["unit"] == "predictions/s" and ens["value"] > 0 and ens["scaling"] == "strong"


def test_lcn_beside_the_match_chains_and_priority_pipeline_run():
    """The CU-partitioned pipeline (--partition) and the other placement of the LCN (--lcn-stream) of the independent-matches mode do the same
    work as its default (priority streams, LCN of frame t+1 on FramePipeline.prep_stream while the U-Net of frame t runs): same PR-GLS
    iteration count - the check that exposed the co-residency hazard of DESIGN.md section 5."""
    base = [sys.executable, "bench.py", "--gpus", "1", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-realistic-pass", "--mode", "independent"]
    ref = _run(base)
    for extra in (["--lcn-stream", "seg"], ["--partition"], ["--partition", "--lcn-stream", "match"]):
        line = _run(base + extra)
        assert KEYS <= set(line) and line["steps"] == 8
        assert line["config"]["prgls_iterations"] == ref["config"]["prgls_iterations"]
        assert 0.4 < line["value"] / ref["value"] < 1.6, (extra, ref["value"], line["value"])


def _two_ranks(backend, extra):
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", backend] + extra
    return _run(cmd, timeout=1500)


def test_two_ranks_gloo_same_device_runs_all_sharding_passes():
    line = _two_ranks("gloo", ["--same-device"])
    assert KEYS <= set(line) and line["n_gpus"] == 2 and line["scaling"] == "weak" and line["cpu_baseline"] is None
    cfg = line["config"]
    assert cfg["rccl_ranks"]["world_size"] == 2 and cfg["rccl_ranks"]["backend"] == "gloo"
    # every frame of every sequence (warm-up, the priming frames, 5 timed windows) from every rank
    assert cfg["rccl_ranks"]["tracked_sets_gathered"] == 2 * (line["warmup"] + min(4, line["steps"]) + 5 * line["steps"])
    assert cfg["patches_sharded"]["per_s"] > 0 and cfg["ensemble_sharded"]["per_s"] > 0
    assert cfg["independent_matches"]["with_discriminating_ffn"]["prgls_iterations"] <= 30 and cfg["headline_excludes"] == []


def test_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2 ...` as the driver would type it: bench.py starts its own ranks (torch.distributed.run)."""
    env_keys = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")
    saved = {k: os.environ.pop(k) for k in env_keys if k in os.environ}
    try:
        line = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--same-device",
                     "--no-realistic-pass"], timeout=1500)
    finally:
        os.environ.update(saved)
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"]["world_size"] == 2 and line["value"] > 0


def test_two_ranks_nccl():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one device per rank: this box has a single GPU")
    line = _two_ranks("nccl", [])
    assert line["n_gpus"] == 2 and line["config"]["patches_sharded"]["per_s"] > 0 and line["config"]["ensemble_sharded"]["per_s"] > 0
    for mode in ("patches", "ensemble"):
        port = _free_port()
        out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--mode", mode,
                    "--no-realistic-pass"], timeout=1500)
        assert out["scaling"] == "strong" and out["value"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("arrangement", ["prio", "plain"])
def test_lcn_regions_correction_are_unaffected_by_a_unet_sharing_their_cus(arrangement):
    """The other small kernels of the frame (LCN with both border modes, regions -> centres, accurate correction) on a high-priority /
    plain second stream while the U-Net's split conv kernels run: bit-identical to their stand-alone results, and the U-Net's output
    to its own (scripts/probe/coresident_probe.py; the match side has its own test in test_gpu_match.py)."""
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(repo / "scripts" / "probe" / "coresident_probe.py"), arrangement, "8"], capture_output=True,
                       text=True, timeout=600, cwd=repo)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "mismatching results: 0 of" in r.stdout, r.stdout[-1500:]


def test_chained_frame_runs_the_reference_region_step():
    """frame.FrameChain (bench.py's config.chained): the region step is the reference's marker watershed by default, the cheap connected-components
    variant on request; both chains track the same synthetic cells, the watershed splits touching cells the components leave joined."""
    import importlib
    import numpy as np
    frame = importlib.import_module("3deecelltracker_amd.frame")
    seg = importlib.import_module("3deecelltracker_amd.segment")
    res = {}
    for method in ("watershed", "cc"):
        chain = frame.FrameChain.synthetic(shape=(256, 256, 24), n_cells=150, seed=3, region_method=method)
        out = chain.run()
        prob, centres = chain.segment(chain.raw_t2)
        if method == "watershed":                         # the chain's centres ARE ct_watershed_segment's on the map it just produced
            _, want, _, _, _ = seg.watershed_centroids_device(prob, 4.0, "min_size", chain.min_size, 0, want_labels=False)
            assert np.array_equal(centres.cpu().numpy(), want.cpu().numpy())
        err = float(np.abs(out["coords"].real - chain.true_t2 * np.array([1.0, 1.0, 4.0])).max(axis=1).mean())
        res[method] = (out["n_segmented"], err)
    assert res["watershed"][0] >= res["cc"][0] >= 100
    assert res["watershed"][1] < 2.0 and res["cc"][1] < 2.0
    with pytest.raises(ValueError):
        frame.FrameChain(None, None, None, 100.0, region_method="stardist")


@pytest.mark.gpu
def test_chained_frame_with_the_reference_set_prepared_beside_the_unet_is_unchanged():
    """FrameChain enqueues what the match needs of frame t1 alone (normalisation, Gram matrix, low-rank factor: ct_prgls_prepare_ref) on a second
    stream beside the U-Net; the frame's results are those of the serial chain bit for bit."""
    import importlib
    import numpy as np
    frame = importlib.import_module("3deecelltracker_amd.frame")
    outs = []
    for prefetch in (True, False):
        chain = frame.FrameChain.synthetic(shape=(256, 256, 24), n_cells=150, seed=4, prefetch_ref=prefetch)
        a = chain.run(); b = chain.run()
        assert np.array_equal(a["coords"].real, b["coords"].real)
        outs.append(b)
    assert outs[0]["prgls_iterations"] == outs[1]["prgls_iterations"] and outs[0]["n_segmented"] == outs[1]["n_segmented"]
    assert np.array_equal(outs[0]["coords"].real, outs[1]["coords"].real)


@pytest.mark.gpu
@pytest.mark.parametrize("region_method", ("watershed", "cc"))
def test_frame_sequence_with_overlapped_unet_equals_the_serial_frames(region_method):
    """FrameChain.run_sequence: every frame is matched against ITS predecessor's segmentation and moves its predecessor's corrected cells; the
    LCN + U-Net of frame i+2 and the watershed of frame i+1 run on their own streams beside frame i's match + correction (three probability
    buffers; with the connected-components variant the region step stays with the match).  Same values as run() frame after frame."""
    import importlib
    import numpy as np
    frame = importlib.import_module("3deecelltracker_amd.frame")
    chain = frame.FrameChain.synthetic(shape=(256, 256, 24), n_cells=150, seed=5, region_method=region_method)
    raws = [chain.raw_t2, chain.raw_t1, chain.raw_t2, chain.raw_t1, chain.raw_t2]
    seg, conf = chain.seg_real_t1, chain.confirmed_real_t1
    want = []
    for r in raws:
        o = chain.run(r, seg, conf)
        want.append(o); seg, conf = o["seg_real_t2"], o["coords"].real
    for _ in range(2):                                            # (the second pass reuses the streams and buffers of the first)
        got = list(chain.run_sequence(raws, chain.seg_real_t1, chain.confirmed_real_t1))
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g["n_segmented"] == w["n_segmented"] and g["prgls_iterations"] == w["prgls_iterations"]
            assert np.array_equal(g["seg_real_t2"].cpu().numpy(), w["seg_real_t2"].cpu().numpy())
            assert np.array_equal(g["coords"].real, w["coords"].real)
    assert len({o["n_segmented"] for o in want}) >= 1 and want[0]["n_segmented"] >= 100
    assert list(chain.run_sequence([], seg, conf)) == []
    # the stacks as the reference's loop finds them: on the HOST (numpy, pinned tensors, or a mix with resident ones) -- uploaded inside the
    # loop on the sequence's copy stream through a ring of four device buffers (more frames than ring slots: every slot is reused); same values
    import torch
    long_raws = raws + raws[:4]                                   # 9 frames
    want9 = list(chain.run_sequence(long_raws, chain.seg_real_t1, chain.confirmed_real_t1))
    pinned = [r.cpu().pin_memory() for r in long_raws]
    as_numpy = [r.cpu().numpy() for r in long_raws]
    mixed = [p if i % 3 else r for i, (p, r) in enumerate(zip(pinned, long_raws))]
    for variant in (pinned, as_numpy, mixed, pinned):
        got = list(chain.run_sequence(variant, chain.seg_real_t1, chain.confirmed_real_t1))
        assert len(got) == len(want9)
        for g, w in zip(got, want9):
            assert g["n_segmented"] == w["n_segmented"] and g["prgls_iterations"] == w["prgls_iterations"]
            assert np.array_equal(g["coords"].real, w["coords"].real)
    with pytest.raises(ValueError):
        list(chain.run_sequence([pinned[0], pinned[1][:, :, :8]], chain.seg_real_t1, chain.confirmed_real_t1))


@pytest.mark.gpu
def test_frame_sequence_survives_an_early_stop_and_another_volume_shape():
    """run_sequence is a generator with streams and probability-map buffers cached on the chain: a consumer that stops after two frames must
    leave the caller's stream ordered behind the frames already enqueued (try / finally joins the three streams), the next sequence on the
    same chain gives the values of a fresh one, and a sequence of another volume shape rebuilds the cached buffers instead of handing the
    U-Net buffers of the wrong shape (round-4 advisor: _seq is keyed on shape and device)."""
    import importlib
    import numpy as np
    import torch
    frame = importlib.import_module("3deecelltracker_amd.frame")
    chain = frame.FrameChain.synthetic(shape=(256, 256, 24), n_cells=150, seed=6)
    raws = [chain.raw_t2, chain.raw_t1, chain.raw_t2, chain.raw_t1]
    want = list(chain.run_sequence(raws, chain.seg_real_t1, chain.confirmed_real_t1))
    gen = chain.run_sequence(raws, chain.seg_real_t1, chain.confirmed_real_t1)
    first = next(gen); next(gen)
    gen.close()                                                   # GeneratorExit inside the loop: the finally clause joins S / W / T
    torch.cuda.synchronize()
    assert np.array_equal(first["coords"].real, want[0]["coords"].real)
    again = list(chain.run_sequence(raws, chain.seg_real_t1, chain.confirmed_real_t1))
    for g, w in zip(again, want):
        assert g["n_segmented"] == w["n_segmented"] and np.array_equal(g["coords"].real, w["coords"].real)
    assert chain._seq["key"][0] == (256, 256, 24)
    other = frame.FrameChain.synthetic(shape=(192, 160, 16), n_cells=60, seed=7)
    small = [other.raw_t2, other.raw_t1]
    # the same chain fed another geometry: U-Net and watershed run on REBUILT buffers of the new shape; the chain's transformer belongs to the
    # first volume, so the correction refuses the map (ValueError) -- inside the generator, whose finally clause still joins the streams
    with pytest.raises(ValueError, match="shape"):
        list(chain.run_sequence(small, other.seg_real_t1, other.confirmed_real_t1))
    torch.cuda.synchronize()
    assert chain._seq["key"][0] == (192, 160, 16)
    back = list(chain.run_sequence(raws, chain.seg_real_t1, chain.confirmed_real_t1))
    assert chain._seq["key"][0] == (256, 256, 24)
    for g, w in zip(back, want):
        assert g["n_segmented"] == w["n_segmented"] and np.array_equal(g["coords"].real, w["coords"].real)
    with pytest.raises(ValueError, match="same shape"):
        list(chain.run_sequence([chain.raw_t2, other.raw_t2], chain.seg_real_t1, chain.confirmed_real_t1))       # mixed shapes in one sequence


def test_rccl_selftest_runs_the_frame_loops_gather_through_rccl_on_one_rank():
    """--rccl-selftest: a one-rank nccl process group inside the bench process; the frame loop's gather of corrected centroid sets (every 8
    frames, on its communication stream) and predict_volume_sharded's collectives go through RCCL; the contract line itself is unchanged."""
    line = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "8", "--warmup", "2", "--windows", "1", "--no-cpu-baseline", "--no-realistic-pass",
                 "--rccl-selftest"])
    assert KEYS <= set(line) and line["config"]["rccl_ranks"]["world_size"] == 1
    r = line["config"]["rccl_single_rank"]
    assert "error" not in r, r
    assert r["backend"] == "nccl" and r["world_size"] == 1 and r["patches_sharded_equals_single_process"] is True
    assert r["tracked_sets_gathered"] == r["frames"] == 32 and r["volumes_per_s"] > 0.5 * line["value"]


def test_the_json_line_is_the_last_line_on_stdout_when_rccl_has_printed_its_banner():
    """RCCL writes a version banner to the C library's stdout when its first communicator is created; with stdout on a pipe it used to come out at
    process exit, after the JSON line.  bench.py flushes C stdio and prints its line last."""
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "4", "--warmup", "1", "--windows", "1", "--no-cpu-baseline",
                          "--no-realistic-pass", "--rccl-selftest"], cwd=REPO, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert lines[-1].startswith("{") and json.loads(lines[-1])["config"]["rccl_single_rank"]["backend"] == "nccl", lines[-3:]
