"""`python bench.py --gpus N` without a launcher must start N ranks itself (the driver's N = 1 command with a larger N): bench.py
re-executes under torch.distributed.run.  `--launch-check` stops after the rendezvous, so this runs without a GPU (gloo); the full
2-rank run of the same command line is tests/test_gpu_bench.py::test_two_ranks_without_a_launcher."""
import json
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def _line(cmd, env=None):
    env = dict(os.environ if env is None else env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_starts_two_ranks():
    line = _line([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "1", "--launch-check"])
    assert line["world_size"] == 2 and line["gpus"] == 2 and line["backend"] == "gloo" and line["ranks"] == [0, 1]


def test_gpus_1_does_not_relaunch():
    line = _line([sys.executable, "bench.py", "--gpus", "1", "--launch-check"])
    assert line["world_size"] == 1 and line["backend"] is None


def test_gpus_8_rendezvous_of_a_whole_node():
    """The driver's N = 8 command line (one rank per GPU of one node): the rendezvous, the rank -> device mapping and the line's shape."""
    line = _line([sys.executable, "bench.py", "--gpus", "8", "--backend", "gloo", "--same-device", "--steps", "1", "--launch-check"])
    assert line["world_size"] == 8 and line["gpus"] == 8 and line["ranks"] == list(range(8))
