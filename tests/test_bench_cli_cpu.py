"""bench.py refuses a launcher whose world size differs from --gpus (the line's n_gpus would not be what was asked for) -- before any GPU call,
so this runs on the CPU box (reference: train.py:123-127 takes its rank count from options.num_gpu alone)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
