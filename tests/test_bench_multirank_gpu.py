"""bench.py's own multi-rank control flow, executed once before a driver with 8 GPUs does (reference: train.py:123-127 starts one process
per GPU under DDPStrategy; bench.py --gpus N does the same through launch_ranks()).

A one-GPU box cannot form a multi-rank RCCL group, so the ranks run in REHEARSAL mode (TCVN_BENCH_REHEARSAL=1: every rank on cuda:0,
exchange over gloo) -- everything else is the real path: launch_ranks() starting the children before any GPU call, the world-size check,
enable_data_parallel() (state broadcast + overlapped arena all-reduce hooks), the barriers around the timed region, the MAX all-reduce
of the elapsed time, rank 0 printing the one JSON line, the exit code of the worst child.  No scaling figure is asserted."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = dict(os.environ, TCVN_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):      # no launcher around us: bench.py starts its ranks itself
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-fp32",
           "--no-sdxl", "--no-batch8"] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    print(p.stdout[-3000:], p.stderr[-3000:])
    assert p.returncode == 0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                      # rank 0 alone prints, once
    return json.loads(lines[0])


def test_bench_two_ranks_weak_scaling_line():
    out = _run([])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 64 and out["config"]["parallelism"] == "dp2"
    assert out["steps"] == 2 and out["warmup"] == 1
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert abs(out["value"] - 64 / out["ms_per_step"] * 1000) <= 1e-2 * out["value"]      # whole-job events per second
    assert out["loss"] == out["loss"] and abs(out["loss"]) < 1e3                           # finite
    assert out["library"] == "libtcvn_hip.so"


def test_bench_two_ranks_strong_scaling_line():
    out = _run(["--global-batch", "16"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["global_batch"] == 16
    assert "batch 8/GPU" in out["config"]["workload"]
    assert out["value"] > 0 and out["loss"] == out["loss"] and abs(out["loss"]) < 1e3

