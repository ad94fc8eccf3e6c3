"""CPU-only: the C-ABI library builds/loads, exports every symbol include/ctamd.h declares, and its
host-only entry points (no kernel launches) behave."""
import ctypes as C
import importlib
import re
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
_lib = importlib.import_module("3deecelltracker_amd._lib")
arch_mod = importlib.import_module("3deecelltracker_amd.arch")


@pytest.fixture(scope="module")
def L():
    if not _lib.LIB_PATH.exists():
        importlib.import_module("__graft_entry__").build()
    return _lib.lib()


def _declared_symbols():
    text = (REPO / "include" / "ctamd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ct_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(L):
    names = _declared_symbols()
    assert len(names) >= 35
    assert _lib.MISSING == []
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/ctamd.h but not exported by libctamd.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes prototype in _lib.SIGNATURES"
    assert sorted(_lib.SIGNATURES) == names


def test_host_only_entry_points(L):
    assert L.ct_version() >= 100
    assert L.ct_error_string(0) == b"ok" and L.ct_error_string(-3) == b"workspace too small"
    for a in arch_mod.ARCHS.values():
        convs = a.conv_layers()
        want = sum(27 * ci * co + 5 * co for ci, co in convs) + a.out[1] + 1
        assert L.ct_unet_num_weights(a.arch_id) == want
        s = (C.c_int * 3)()
        assert L.ct_unet_patch_shape(a.arch_id, s) == 0 and tuple(s) == a.input_shape
    assert L.ct_unet_num_weights(7) == 0
    assert L.ct_unet_num_weights(0) == 510393 + 1632          # conv params + BN params
    assert L.ct_ffn_num_weights() == 61 * 512 + 1024 * 512 + 513 + 2 * 4 * 512


def test_tile_plan_matches_reference_counts(L):
    centre, grid = (C.c_int * 3)(), (C.c_int * 3)()
    for vol, want in (((64, 64, 16), 2), ((256, 256, 24), 18), ((512, 512, 32), 75), ((168, 401, 128), 88)):
        rc = L.ct_tile_plan(_lib.ivec(vol), _lib.ivec((160, 160, 16)), _lib.ivec((24, 24, 2)), centre, grid)
        assert rc == 0 and tuple(centre) == (112, 112, 12) and int(np.prod(tuple(grid))) == want
    assert L.ct_tile_plan(_lib.ivec((64, 64, 16)), _lib.ivec((40, 40, 16)), _lib.ivec((24, 24, 2)), centre, grid) == -2
    assert L.ct_tile_plan(_lib.ivec((0, 64, 16)), _lib.ivec((160, 160, 16)), _lib.ivec((24, 24, 2)), centre, grid) == -1


def test_argument_validation_without_a_gpu(L):
    assert L.ct_knn_features(None, 10, 20, None, None) == -1
    assert L.ct_greedy_workspace_bytes(0, 5) == 0 and L.ct_greedy_workspace_bytes(600, 600) > 0
    assert L.ct_prgls_workspace_bytes(600, 600, 600) > 3 * 600 * 600 * 8
    assert L.ct_ffn_workspace_bytes(600, 600) >= 2 * 1200 * 512 * 4
    assert L.ct_unet_workspace_bytes(None, 3) == 0


def test_product_raises_without_library_or_gpu(monkeypatch):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    unet3d = importlib.import_module("3deecelltracker_amd.unet3d")
    synth = importlib.import_module("3deecelltracker_amd.synth")
    with pytest.raises(_lib.CtamdError):            # no silent CPU fallback
        unet3d.unet3_c().set_weights_dict(synth.make_unet_weights("unet3_c", 0))
    monkeypatch.setattr(_lib, "LIB_PATH", Path("/nonexistent/libctamd.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.CtamdError):
        _lib.lib()


def test_no_packed_fp32_outside_the_conv_kernels(L):
    """Build gate: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on LDS-fed operands returned wrong values beside the
    split conv kernels' waves.  Every device object except ct_unet must be free of them -- checked on the disassembly of what was
    built, so a compiler bump or a float2 edit cannot reintroduce them silently -- and only the C ABI leaves the library."""
    import subprocess
    import sys
    sys.path.insert(0, str(REPO / "scripts"))
    try:
        import check_packed_fp32 as gate
    finally:
        sys.path.pop(0)
    objs = sorted((REPO / "3deecelltracker_amd" / "csrc").glob("*.o"))
    assert {o.stem for o in objs} >= {"ct_unet", "ct_match", "ct_preprocess", "ct_segment", "ct_correct"}
    assert gate.check(objs) == {}
    assert sum(gate.packed_fp32_by_kernel(REPO / "3deecelltracker_amd" / "csrc" / "ct_unet.o").values()) > 0   # the detector sees them
    nm = subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = [l.split()[-1] for l in nm.splitlines() if l.strip()]
    assert exported and all(n.startswith("ct_") for n in exported), [n for n in exported if not n.startswith("ct_")][:5]


def test_package_leaves_the_environment_alone_and_warns_about_hardware_queues():
    """FramePipeline keeps five to six HIP streams busy; on the runtime's default of four hardware queues they alias.  The package
    must not set GPU_MAX_HW_QUEUES behind the host application's back (a process-global side effect that is silently ineffective once HIP is
    initialised): importing it leaves the environment alone, and _lib.check_hw_queues() -- what FramePipeline (the multi-chain mode) calls -- warns once
    when the variable is missing or too small."""
    import os
    import subprocess
    code = ("import sys, os, importlib, warnings; sys.path.insert(0, sys.argv[1]); L = importlib.import_module('3deecelltracker_amd._lib'); "
            "print('Q=' + os.environ.get('GPU_MAX_HW_QUEUES', 'unset'))\n"
            "with warnings.catch_warnings(record=True) as w:\n"
            "    warnings.simplefilter('always'); ok1 = L.check_hw_queues(); ok2 = L.check_hw_queues()\n"
            "print('OK=%s/%s W=%d' % (ok1, ok2, sum('GPU_MAX_HW_QUEUES' in str(x.message) for x in w)))")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    r = subprocess.run([sys.executable, "-c", code, str(REPO)], capture_output=True, text=True, timeout=120, env=env)
    assert "Q=unset" in r.stdout and "OK=False/False W=1" in r.stdout, r.stdout + r.stderr[-300:]
    r = subprocess.run([sys.executable, "-c", code, str(REPO)], capture_output=True, text=True, timeout=120, env=dict(env, GPU_MAX_HW_QUEUES="6"))
    assert "Q=6" in r.stdout and "OK=False/False W=1" in r.stdout, r.stdout + r.stderr[-300:]
    r = subprocess.run([sys.executable, "-c", code, str(REPO)], capture_output=True, text=True, timeout=120, env=dict(env, GPU_MAX_HW_QUEUES="16"))
    assert "Q=16" in r.stdout and "OK=True/True W=0" in r.stdout, r.stdout + r.stderr[-300:]
