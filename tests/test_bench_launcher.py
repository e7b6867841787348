"""bench.py's multi-GPU launch: `python bench.py --gpus N` must run N ranks or fail, never report fewer.

The reference shares a render over num_cpus::get() threads (src/render.rs:128-151); bench.py shares it over N GPUs with one
rank process per GPU.  These tests drive the launcher alone (no rendering): on this CPU-only host the ranks count each other
with one gloo all-reduce.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout)


def test_launcher_starts_the_ranks_itself():
    """No WORLD_SIZE in the environment: bench.py becomes the parent of a torch.distributed.run child with 2 ranks."""
    r = run(["--gpus", "2", "--launch-check", "--backend", "gloo"], {"RR_BENCH_ASSUME_DEVICES": "2"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out == {"launch_check": True, "n_gpus": 2, "n_ranks_seen": 2, "world_size": 2}


def test_launcher_refuses_when_fewer_devices_are_visible_than_asked_for():
    r = run(["--gpus", "8", "--launch-check", "--backend", "gloo"], {"RR_BENCH_ASSUME_DEVICES": "2"})
    assert r.returncode != 0
    assert "--gpus 8 asked for, 2 GPU(s) visible" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU host: the real device count is tested by the gpu tests")
def test_a_real_run_without_enough_gpus_fails_before_touching_anything():
    r = run(["--gpus", "2"])
    assert r.returncode != 0
    assert "refusing to measure fewer GPUs than asked" in r.stderr


def test_a_world_that_is_not_the_number_of_gpus_asked_for_is_an_error():
    """Started under a launcher with another world size than --gpus: bench.py exits instead of reporting one as the other."""
    r = run(["--gpus", "8", "--launch-check"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr
    assert not r.stdout.strip()
