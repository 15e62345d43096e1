"""CPU: the LDS-DMA staging protocols of every global_load_lds kernel, checked against the code hipcc actually emits for gfx950.

tools/dma_isa_check.py replays each kernel's instruction stream (VMEM operations retire in issue order): the prefill flash
attention, the 256x256 GEMM and the halo convolution's weight ring must retire a staged tile ONE BARRIER BEFORE the phase that
reads it and re-stage a slot only behind a barrier that follows its last read; tools/sk4_isa_check.py (called from it) walks the
114 decode-GEMM instantiations with their hand-counted `s_waitcnt vmcnt(N)`.  A compiler that reorders a load across a counted
wait -- what caused the round-2 cold-launch failure -- or a new LDS-DMA kernel without a protocol spec fails here without a GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="hipcc not available")
def test_lds_dma_protocols_match_the_compiled_code():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dma_isa_check.py")], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]
    out = p.stdout
    assert out.strip().endswith("0 failed")
    assert out.count("fifo protocol holds in the strict form") >= 4           # flash2 + three gemm256 instantiations (plain, RoPE epilogue, conv)
    assert out.count(": weight ring strict") == 2                             # production (staggered) halo convolution, plain + upsample
    assert out.count("double buffer, strict form holds") == 2                 # the 128x128 GEMM (plain / implicit-im2col loaders), two barriers per K tile
    assert out.count("lock-step weight ring strict") == 2                     # conv_halo=2 option kernels
    assert out.count("patch retired") == 1                                    # conv_out halo kernel
    assert "weak" not in out and "legacy" not in out                          # no LDS-DMA kernel is left on the same-phase form
    sk4 = [l for l in out.splitlines() if l.startswith("sk4:")][0]
    assert "0 failed" in sk4 and int(sk4.split()[1]) >= 40                    # production + bench instantiations of the decode GEMM
