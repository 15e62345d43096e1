"""CPU: the LDS-DMA staging protocols of every global_load_lds kernel, checked against the code hipcc actually emits for gfx950.

tools/dma_isa_check.py replays each kernel's instruction stream (VMEM operations retire in issue order; round 6: the two-phase 256x256 GEMM's
WAR margin with its wave groups one barrier apart, and the v5 decode GEMM's whole hand-counted stream, tools/sk5_isa_check.py): the prefill flash
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
    import json
    line = [l for l in p.stdout.splitlines() if l.startswith("SUMMARY ")][-1]
    r = json.loads(line[len("SUMMARY "):])                                    # machine-readable result; the report's wording is free
    assert r["failed"] == 0
    v = r["verified"]
    assert v.get("fifo", 0) >= 4             # flash2 + three gemm256 instantiations (plain, RoPE epilogue, conv)
    assert v.get("halo_stag", 0) == 8        # production (staggered) halo convolution: plain (generic + 4 fast epilogues) and upsample (generic + 2)
    assert v.get("big", 0) == 2              # the 128x128 GEMM (plain / implicit-im2col loaders), two barriers per K tile
    assert v.get("halo_lock", 0) == 2        # conv_halo=2 option kernels
    assert v.get("once", 0) == 1             # conv_out halo kernel
    assert v.get("sink", 0) >= 6             # run-ahead weight prefetcher instantiations: LDS-DMA into a sink that is never read
    assert v.get("fifo_weak", 0) <= 1        # only the diagnostics library's big-wave GEMM experiment (diag_gemm_bw.hip) reads behind its retiring barrier
    assert set(v) <= {"fifo", "fifo_weak", "halo_stag", "big", "halo_lock", "once", "sink"}  # every PRODUCT kernel is on a STRICT spec (no same-phase form left)
    assert r["sk4"]["rc"] == 0 and r["sk4"]["failed"] == 0 and r["sk4"]["checked"] >= 40   # production + bench instantiations of the decode GEMM
    assert r["sk5"]["failed"] == 0 and r["sk5"]["checked"] >= 4                             # round 6: the v5 wide-N kernel (8 / 16 chunks x slab / SwiGLU epilogue), exact stream replay


def test_inline_asm_stores_carry_their_wait_states():
    """hipcc treats an asm statement as opaque: it neither keeps a 96/128-bit store's data registers alive nor pads the store-data
    hazard, so the wait states must sit INSIDE the statement (CDNA guide 5.7).  The missing `s_nop 1` behind the decode GEMM's
    write-through `global_store_dwordx4` was the root cause of the rounds 2-3 "stale x piece" failures.
    Source-level rule, checked for every inline-asm store in csrc/: a *_store_dwordx3 / x4 is followed by an s_nop >= 1 in the
    same string."""
    import glob
    import re
    bad, seen = [], 0
    for f in sorted(glob.glob(os.path.join(ROOT, "plangen_amd", "csrc", "*.h*"))):
        src = open(f).read()
        for m in re.finditer(r'asm\s+volatile\s*\(\s*((?:"(?:[^"\\]|\\.)*"\s*)+)', src):
            text = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', m.group(1)))
            for st in re.finditer(r"(global|buffer|flat|scratch)_store_dwordx[34]", text):
                seen += 1
                tail = text[st.end():]
                nop = re.search(r"s_nop\s+(\d+)", tail)
                if not nop or int(nop.group(1)) < 1:
                    bad.append((os.path.basename(f), src.count("\n", 0, m.start()) + 1))
    assert seen >= 2, "the write-through epilogue stores were not found: update this check with the code"
    assert not bad, f"inline-asm wide stores without their wait states: {bad}"
