"""bench.py --gpus N without a launcher around it starts its own ranks (the driver's 1-GPU command form is
plain `python bench.py --gpus 1 ...`). Without a GPU the ranks can only get as far as "bench.py needs an MI355X" -- which is
exactly what shows that N children were started through torch.distributed.run and that their status comes back."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_2_without_launcher_starts_two_ranks_and_relays_their_status():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-side check; the GPU box runs tests/test_gpu_sharded.py::test_bench_gpus_2_spawns_its_own_ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0                                  # the children's failure is the parent's exit status
    assert r.stderr.count("bench.py needs an MI355X") >= 2    # both ranks ran bench.py's main() under the launcher
    assert "launch through torch.distributed.run" not in r.stderr


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr
