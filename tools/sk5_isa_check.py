#!/usr/bin/env python3
"""Static check of the v5 decode GEMM (gemm_sk5_kernel, gemm_skinny.h) against the code hipcc EMITS: the kernel's waits are hand-counted
`s_waitcnt vmcnt(N)` over ONE in-order counter shared by the x tile's LDS-DMA pieces and the W fragment loads, so the issue order in the
instruction stream must be exactly the order sk5_wait_x / sk5_wait_w simulate at compile time.  For every instantiation in gemm.hip's listing:

  prologue   2 plain loads (row-scale partials, oldest), X(0..XD-2) = 4 DMA pieces each, W(0..WD-1) = 8 loads each,
             vmcnt(ops after X(0)) + barrier                      (X(0) retired one barrier before its first read)
  chunk c    vmcnt(ops after X(min(c+1, NH-1))) + barrier          (X(c+1) retired one barrier before ITS first read: the strict staging rule)
             X(c+XD-1) issued behind that barrier                  (into the slot of X(c-1), whose readers passed the barrier)
             vmcnt(ops after W(c)); 16 fragment reads; 32 MFMAs; W(c+WD) issued
  epilogue   vmcnt(0) + barrier, 4 exchange writes, barrier, vmcnt(0), 4 exchange reads, 4 stores

The expected event sequence is rebuilt here from (NCK, XD, WD) with its own op counting and compared EXACTLY with the compiled one: a compiler that
moves a load across a counted wait, drops a barrier, or re-orders the DMA against the W ring fails the check.   usage: sk5_isa_check.py [gemm.s]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def listing(path=None):
    if path:
        return open(path).read().splitlines()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dma_isa_check as D                      # content-keyed -S cache with the Makefile's code-generation flags
    return D.compile_s("gemm")


def events(lines, i):
    ev, j = [], i + 1
    while j < len(lines) and "s_endpgm" not in lines[j]:
        t = lines[j].strip()
        if t.startswith("global_load_lds"): ev.append("X")
        elif t.startswith("global_load_dwordx4") and t.endswith("nt"): ev.append("W")
        elif t.startswith(("global_load", "buffer_load", "flat_load")): ev.append("L")
        elif t.startswith(("global_store", "buffer_store", "flat_store")): ev.append("S")
        elif t.startswith("s_waitcnt") and "vmcnt(" in t: ev.append("w%d" % int(re.search(r"vmcnt\((\d+)\)", t).group(1)))
        elif t.startswith("s_barrier"): ev.append("|")
        elif t.startswith("v_mfma"): ev.append("m")
        elif t.startswith("ds_read"): ev.append("r")
        elif t.startswith("ds_write"): ev.append("d")
        elif t.startswith("scratch_"): ev.append("!")
        j += 1
    return ev


def expected(NCK, XD, WD):
    NH = NCK // 2
    ev, ops, lastX, lastW = ["L", "L"], 0, {}, {}
    for p in range(min(XD - 1, NH)):
        ev += ["X"] * 4; ops += 4; lastX[p] = ops
    for p in range(min(WD, NH)):
        ev += ["W"] * 8; ops += 8; lastW[p] = ops
    ev += ["w%d" % (ops - lastX[0]), "|"]
    for c in range(NH):
        ev += ["w%d" % (ops - lastX[min(c + 1, NH - 1)]), "|"]
        if c + XD - 1 < NH:
            ev += ["X"] * 4; ops += 4; lastX[c + XD - 1] = ops
        ev += ["w%d" % (ops - lastW[c])] + ["r"] * 16 + ["m"] * 32
        if c + WD < NH:
            ev += ["W"] * 8; ops += 8; lastW[c + WD] = ops
    ev += ["w0", "|"] + ["d"] * 4 + ["|", "w0"] + ["r"] * 4           # __syncthreads() drains vmcnt (nothing is outstanding any more), then the exchange
    return ev


def main():
    lines = listing(sys.argv[1] if len(sys.argv) > 1 else None)
    checked = bad = 0
    for i, l in enumerate(lines):
        m = re.match(r"^_Z15gemm_sk5_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)EE\S*: ; @", l)
        if not m:
            continue
        NCK, XD, WD, EPI = (int(v) for v in m.groups())
        got = events(lines, i)
        want = expected(NCK, XD, WD)
        checked += 1
        body, tail = got[:len(want)], got[len(want):]
        ok = body == want and "!" not in got and all(e in ("S", "L") or e.startswith("w") for e in tail) and tail.count("S") >= 4
        if not ok:
            bad += 1
            k = next((k for k, (a, b) in enumerate(zip(got, want)) if a != b), min(len(got), len(want)))
            print(f"FAIL gemm_sk5_kernel<{NCK},{XD},{WD},{EPI}>: first difference at event {k}: compiled {' '.join(got[max(0, k - 6):k + 6])} | expected {' '.join(want[max(0, k - 6):k + 6])}")
        else:
            print(f"ok   gemm_sk5_kernel<{NCK},{XD},{WD},{EPI}>: {len(want)} events in the simulated order (waits {[e for e in want if e.startswith('w')][:6]} ...), no scratch")
    print(f"{checked} instantiations checked, {bad} failed")
    return 1 if bad or not checked else 0


if __name__ == "__main__":
    sys.exit(main())
