"""CPU: the v4 decode GEMM's hand-counted `s_waitcnt vmcnt(N)` and its LDS ring discipline, checked against the code hipcc
actually emits for gfx950 (tools/sk4_isa_check.py walks `hipcc -S` output of gemm.hip: every barrier is preceded by a wait that
retires the chunk's x DMA pieces, every MFMA reads W fragments from retired loads, every x fragment read sits between the barrier
of its chunk and the next).  A compiler that reorders a load across one of the counted waits -- what caused the round-2 cold-launch
failure -- fails here without a GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="hipcc not available")
def test_sk4_counted_waits_match_the_compiled_code(tmp_path):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sk4_isa_check.py")], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]
    assert "0 failed" in p.stdout and "instantiations checked" in p.stdout
    assert int(p.stdout.strip().split()[0]) >= 40           # production + bench instantiations
