"""bench.py --gpus N must start its own N ranks when no launcher is around it: the driver calls
`python bench.py --gpus N ...` exactly as it does for N = 1.  No GPU is needed for the launcher itself."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("n", [2, 4])
def test_bench_self_launches_its_ranks(n):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1",
                        "--launch-only"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert sorted(r["rank"] for r in rows) == list(range(n))
    assert all(r["world"] == n and r["launch_only"] for r in rows)
    assert sorted(r["local_rank"] for r in rows) == list(range(n))
    assert all(r["master"].startswith("127.0.0.1:") for r in rows)


def test_bench_under_an_external_launcher_does_not_relaunch():
    """the driver's N > 1 form: torch.distributed.run around bench.py -- the ranks must run, not spawn again"""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--launch-only"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert sorted(r["rank"] for r in rows) == [0, 1] and all(r["world"] == 2 for r in rows)


def test_world_size_mismatch_is_an_error():
    env = _clean_env()
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-only"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
