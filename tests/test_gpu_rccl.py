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
