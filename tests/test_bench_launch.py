"""bench.py's launch contract (no GPU needed): `--gpus N` either runs N ranks or fails; it never prints a line labelled
with a GPU count it did not use."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SIMPB_BENCH_DEVICE")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_flag_disagreeing_with_the_launcher_fails():
    r = _run(["--gpus", "8"], dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout) and '"n_gpus"' not in r.stdout
    r = _run(["--gpus", "1"], dict(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and '"n_gpus"' not in r.stdout


def test_gpus_flag_without_launcher_refuses_a_node_with_fewer_gpus():
    """No launcher, --gpus 8, fewer than 8 GPUs visible (none in the build container): a non-zero exit with the reason,
    not a one-GPU benchmark labelled n_gpus = 1 (on an 8-GPU node the same call starts the 8 ranks itself)."""
    import torch
    if torch.cuda.device_count() >= 8:
        import pytest
        pytest.skip("an 8-GPU node: this call would start the real benchmark")
    r = _run(["--gpus", "8"], {})
    assert r.returncode != 0 and "GPU(s)" in (r.stderr + r.stdout) and '"n_gpus"' not in r.stdout
