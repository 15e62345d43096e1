#!/usr/bin/env python3
"""Static check of the v4 decode GEMM's hand-counted waits against the COMPILED code.

For every gemm_sk4_kernel instantiation (no ablation bits; bit 64 = early retire, checked with the stricter rule) in `hipcc -S` output of gemm.hip: walk the VMEM instructions in program order
(global_load_lds = x DMA piece, `global_load_dwordx4 ... nt` = W fragment load, stores), and at every `s_waitcnt vmcnt(N)`
  * in front of the s_barrier of chunk c: every DMA piece of x chunk c must be among the retired ops (ops issued - N >= index of its last piece);
  * every MFMA: the W fragment registers it reads must come from retired loads;
  * every DMA piece of x chunk k is issued after barrier k - XD + 1 (the readers of the slot it overwrites are past their reads);
  * every ds_read_b128 of the x ring sits between the barrier of the chunk whose slot it addresses and the next barrier
    (direct-store epilogues only: the LDS-transposed epilogues reuse the ring).
VMEM ops retire in issue order (tools/dma_order_probe.py), so this is exactly the condition the kernel relies on.
usage: sk4_isa_check.py [gemm.s]   (default: compiles plangen_amd/csrc/gemm.hip to /tmp/sk4_check.s)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/sk4_check.s"
if len(sys.argv) < 2:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-S", "--cuda-device-only",
                           os.path.join(ROOT, "plangen_amd/csrc/gemm.hip"), "-I", os.path.join(ROOT, "plangen_amd/csrc"), "-o", path],
                          stderr=subprocess.DEVNULL)
lines = open(path).read().splitlines()
name_re = re.compile(r"^_Z15gemm_sk4_kernelI((?:Li\d+E)+)E")
bad = checked = 0
i = 0
while i < len(lines):
    m = name_re.match(lines[i])
    if not m or ": ; @" not in lines[i]:
        i += 1; continue
    p = [int(x) for x in re.findall(r"Li(\d+)E", m.group(1))]
    MT, NCK, XD, WD, EPI, OCC = p[:6]
    ABL = p[6] if len(p) > 6 else 0
    MS = p[7] if len(p) > 7 else 1
    NWN = p[8] if len(p) > 8 else 4
    j = i + 1
    ev = []
    while j < len(lines) and "s_endpgm" not in lines[j]:
        t = lines[j].strip()
        if t.startswith("global_load_lds"): ev.append(("X", None))
        elif t.startswith("global_load_dwordx4") and t.endswith("nt"): ev.append(("W", re.search(r"(v\[\d+:\d+\])", t).group(1)))
        elif t.startswith(("global_load", "buffer_load")): ev.append(("L", None))
        elif t.startswith(("global_store", "buffer_store")): ev.append(("S", None))
        elif t.startswith("s_waitcnt") and "vmcnt(" in t: ev.append(("wait", int(re.search(r"vmcnt\((\d+)\)", t).group(1))))
        elif t.startswith("s_barrier"): ev.append(("B", None))
        elif t.startswith("ds_read_b128"):
            mo = re.search(r"offset:(\d+)", t)
            ev.append(("R", int(mo.group(1)) if mo else 0))
        elif t.startswith("v_mfma"): ev.append(("M", re.findall(r"(v\[\d+:\d+\])", t)))
        j += 1
    i = j
    if ABL & ~64:
        continue                                      # timing ablations / stamped variants
    AH = 1 if ABL & 64 else 0                         # early retire: chunk c+1 must be retired at barrier c (read one barrier after its retirement)
    XPW = MT * 4 // (NWN * MS)
    issued = 0; retired = 0; xs = []; wreg = {}        # wreg: W destination register range -> issue index of the load that last wrote it
    chunk = -1; ok = True; why = ""
    for kind, val in ev:
        if kind in ("X", "W", "L", "S"):
            issued += 1
            if kind == "X":
                xs.append(issued)
                kx = (len(xs) - 1) // XPW                   # x chunk this piece belongs to; it overwrites the ring slot of chunk kx - XD
                if kx - (XD - 1) > chunk:                   # ... whose last readers are only known to be done after barrier kx - XD + 1
                    ok = False; why = f"x piece of chunk {kx} issued before barrier {kx - XD + 1} (only {chunk + 1} barriers passed): ring slot may still be read"; break
            if kind == "W": wreg[val] = issued
        elif kind == "wait":
            retired = max(retired, issued - val)          # in-order retirement: all but the `val` youngest ops are done
        elif kind == "B":
            chunk += 1
            if chunk < NCK:
                tgt = min(chunk + AH, NCK - 1)
                if len(xs) < (tgt + 1) * XPW: ok = False; why = f"chunk {chunk}: x pieces of chunk {tgt} were not issued before the barrier"; break
                need = xs[(tgt + 1) * XPW - 1]
                if retired < need: ok = False; why = f"barrier {chunk}: x piece #{need} of chunk {chunk} may still be in flight (retired {retired})"; break
        elif kind == "R":
            # x fragment read: must sit between barrier c and barrier c+1 of the chunk whose ring slot it addresses (a read that sank
            # below the next barrier could see the slot re-staged; one hoisted above its barrier could see pieces still landing)
            XB = MT * 16 * 256
            if EPI >= 2 and XB * XD <= 65536 and chunk >= 0 and chunk < NCK and (val // XB) % XD != chunk % XD:       # (slot visible in the 16-bit offset immediate only when the ring fits it: MT <= 4)
                ok = False; why = f"ds_read of ring slot {(val // XB) % XD} between barrier {chunk} and {chunk + 1} (chunk {chunk} lives in slot {chunk % XD})"; break
            if EPI >= 2 and chunk < 0: ok = False; why = "x fragment read in front of the first barrier"; break
        elif kind == "M":
            for r in val[1:3]:                            # A / B operands
                if r in wreg and retired < wreg[r]:
                    ok = False; why = f"MFMA reads {r} (W load #{wreg[r]}) with only {retired} ops retired"; break
            if not ok: break
    checked += 1
    if not ok or chunk + 1 < NCK:
        bad += 1
        print(f"FAIL gemm_sk4_kernel<{','.join(map(str, p))}>: {why or f'only {chunk + 1} barriers for {NCK} chunks'}")
print(f"{checked} instantiations checked, {bad} failed")
sys.exit(1 if bad else 0)
