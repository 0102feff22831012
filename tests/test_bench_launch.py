"""bench.py's process plumbing, without a GPU: `--gpus N` must start N ranks itself (before any GPU call), and a
torchrun environment whose WORLD_SIZE disagrees with --gpus must be refused."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout, cwd=ROOT)


def test_gpus_n_refuses_a_node_with_fewer_gpus():
    """No GPU here: --gpus 2 must fail loudly (not run one rank and print n_gpus: 1)."""
    r = _run(["--gpus", "2"])
    assert r.returncode != 0
    assert "shows 0 GPU" in r.stderr


def test_gpus_n_starts_n_ranks_before_any_gpu_call():
    """With the one-device rehearsal switch the parent launches torch.distributed.run with 2 ranks; on this GPU-less host
    each rank announces itself and then stops at 'needs an MI355X' — which proves rank processes were started with
    WORLD_SIZE=2 and that the parent itself never needed a GPU to get there (torchrun ends the sibling of the first rank
    to fail, so one announcement is enough)."""
    r = _run(["--gpus", "2", "--workload", "cfg2"], {"NCF_BENCH_SINGLE_DEVICE": "1", "NCF_BENCH_BACKEND": "gloo", "NCF_BENCH_ANNOUNCE": "1"})
    assert r.returncode != 0
    assert " of 2 (local rank" in r.stderr and "bench.py needs an MI355X" in r.stderr, r.stderr[-2000:]


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "--gpus 4 but WORLD_SIZE=2" in r.stderr
