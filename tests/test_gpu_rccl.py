"""GPU: a one-rank RCCL (``nccl`` backend) process group beside the engine: what one GPU can prove about the
multi-GPU path -- see tests/helpers/rccl_one_rank.py.  Runs in a child process under a timeout (a communicator that hangs must not
hang the suite) on its own rendezvous port."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_one_rank_rccl_group_runs_the_dist_collectives_beside_the_engine():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_one_rank.py"), "nccl"], env=env,
                       capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    print(out)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "rccl_one_rank.json"), "w"))
    assert out["rccl_one_rank"] == "ok" and out["librccl_loaded"] and out["libplangen_loaded"]


def _bench(*argv, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_two_ranks_on_one_gpu_shard_check():
    """The N > 1 path of bench.py with REAL engines: two rank processes share this box's GPU (gloo transport, PG_FORCE_DEVICE: RCCL
    refuses two ranks on one device), rank 0 collates and broadcasts, each rank generates its contiguous shard, tokens are gathered to
    rank 0, and the shard check regenerates rank 1's shard on rank 0: zero mismatches, both ranks timed."""
    rc, out, err = _bench("--gpus", "2", "--batch", "2", "--tokens", "24", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-roofline",
                          env={"PG_FORCE_DEVICE": "0", "PG_DIST_BACKEND": "gloo", "PG_DIST_TIMEOUT_S": "300"})
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["dist_backend"] == "gloo" and len(out["rank_ms_per_step"]) == 2
    assert out["config"]["global_batch"] == 4 and out["shard_check"]["mismatches"] == 0 and out["shard_check"]["tokens_compared"] == 2 * 24


def test_launcher_refuses_more_ranks_than_gpus():
    """`--gpus 4` on a one-GPU box: the launcher counts the devices in a throw-away child (it never touches the GPU itself) and says so,
    instead of spawning ranks that die with "invalid device ordinal" inside a collective."""
    import torch
    n = torch.cuda.device_count()
    rc, out, err = _bench("--gpus", str(n + 3), "--steps", "1", "--warmup", "0", timeout=600)
    assert rc == 2 and out is None and f"this box exposes {n} GPU" in err, (rc, err[-500:])
