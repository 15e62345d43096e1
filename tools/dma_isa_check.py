#!/usr/bin/env python3
"""Static check of the LDS-DMA (global_load_lds) staging protocols against the code hipcc EMITS for gfx950.

Rule being checked (CDNA guide; the same-phase form of the decode GEMM read a stale DMA piece in ~0.5 % of cold launches,
DESIGN 4.1): a staged tile is RETIRED -- the issuing wave's `s_waitcnt vmcnt(N)` followed by a block barrier -- one barrier
BEFORE the phase that first reads it, never in the phase its retiring barrier opens; and a ring slot is re-staged only after
a barrier that follows its last read.

VMEM operations retire in issue order (tools/dma_order_probe.py), so a kernel's protocol can be replayed from the instruction
stream alone: walk prologue + R copies of the main loop, number the global_load_lds instructions, and at every
`s_waitcnt vmcnt(N)` mark all but the N youngest VMEM operations retired.  Every kernel that contains a global_load_lds must
have a spec below (a new LDS-DMA kernel without one fails the check):

  fifo    tiles are consumed in the order they were issued; the spec names which LDS-read instruction class consumes which
          tile of the stream in loop iteration i.  Checked per read: the tile was retired as of the barrier BEFORE the most
          recent one (strict form), and per DMA: the tile that last lived in its slot was read before an earlier barrier.
  sk4     gemm_sk4_kernel: tools/sk4_isa_check.py (own walker: W register ring + x ring share one vmcnt counter).
  big / halo_lock / halo_stag / once / probe: per-kernel replays of the same rule (128x128 GEMM, the halo convolution's two wave schedules),
          the conv_out halo (staged once per tile) and the measurement probe are listed.

usage: dma_isa_check.py            (compiles plangen_amd/csrc/*.hip with -S into a content-keyed cache under $TMPDIR/dma_isa_<uid>/)"""
import os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "plangen_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# the code-generation flags of plangen_amd/csrc/Makefile (the listing checked here must be the code that ships)
ISA_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-amdgpu-kernarg-preload-count=16"]
_S_PATH = {}


def _compiler_id():
    try:
        return subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.replace("\n", " ")
    except OSError:
        return "unknown"


def compile_s(name):
    """hipcc -S of one source into a cache keyed on the CONTENT of the source + headers and on the compiler version (not on
    mtimes: a checkout or a compiler change must not reuse a stale listing); written under a temporary name and renamed, so
    concurrent runs cannot read a half-written file."""
    import hashlib, tempfile
    src = os.path.join(CSRC, name + ".hip")
    deps = [src] + sorted(os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h"))
    hsh = hashlib.sha256((_compiler_id() + " ".join(ISA_FLAGS)).encode())
    for d in deps:
        hsh.update(open(d, "rb").read())
    out = os.path.join(tempfile.gettempdir(), f"dma_isa_{os.getuid()}")
    os.makedirs(out, exist_ok=True)
    dst = os.path.join(out, f"{name}.{hsh.hexdigest()[:16]}.s")
    if not os.path.exists(dst):
        fd, tmp = tempfile.mkstemp(suffix=".s", dir=out)
        os.close(fd)
        subprocess.check_call([HIPCC, *ISA_FLAGS, "-S", "--cuda-device-only", src, "-I", CSRC, "-o", tmp], stderr=subprocess.DEVNULL)
        os.replace(tmp, dst)
        for f in os.listdir(out):                                  # drop older listings of the same source
            if f.startswith(name + ".") and f.endswith(".s") and os.path.join(out, f) != dst:
                try:
                    os.remove(os.path.join(out, f))
                except OSError:
                    pass
    _S_PATH[name] = dst
    return open(dst).read().splitlines()


def functions(lines):
    """name -> list of (kind, payload) events in program order; labels and branches kept for loop detection."""
    out, i = {}, 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):\s*; @", lines[i])
        if not m:
            i += 1
            continue
        name, ev, j = m.group(1), [], i + 1
        while j < len(lines) and "s_endpgm" not in lines[j]:
            t = lines[j].strip()
            if re.match(r"^\.LBB\d+_\d+:", t): ev.append(("label", t.split(":")[0]))
            elif t.startswith(("s_cbranch", "s_branch")): ev.append(("branch", t.split()[-1]))
            elif t.startswith("global_load_lds") or (t.startswith("buffer_load") and " lds" in t): ev.append(("dma", None))
            elif t.startswith(("global_load", "buffer_load", "global_store", "buffer_store", "global_atomic", "flat_load", "flat_store")): ev.append(("vmem", None))
            elif t.startswith("s_waitcnt") and ("vmcnt(" in t or "lgkmcnt(0)" in t):
                if "lgkmcnt(0)" in t: ev.append(("drain", None))          # every LDS read issued so far has returned
                if "vmcnt(" in t: ev.append(("wait", int(re.search(r"vmcnt\((\d+)\)", t).group(1))))
            elif t.startswith("s_barrier"): ev.append(("bar", None))
            elif t.startswith("ds_read"): ev.append(("read", t.split()[0]))
            j += 1
        out[name] = ev
        i = j
    return out


def main_loop(ev):
    """(prologue, body): the innermost backward-branch loop that contains both DMA issues and LDS reads."""
    pos = {p: k for k, (kind, p) in enumerate(ev) if kind == "label"}
    best = None
    for k, (kind, p) in enumerate(ev):
        if kind == "branch" and p in pos and pos[p] < k:
            body = ev[pos[p]:k]
            if any(e[0] == "dma" for e in body) and any(e[0] == "read" for e in body):
                if best is None or len(body) < best[1] - best[0]:
                    best = (pos[p], k)
    if best is None:
        return None, None
    return ev[:best[0]], ev[best[0]:best[1]]


def replay(prologue, body, spec, R=6):
    """Replays prologue + R x body.  spec: g (DMA instructions per tile per wave), need(i, cls) -> tile index consumed by read
    class cls in iteration i (None: not a staged read), slot_reuse (a tile's slot is reused by tile + slot_reuse(cls-of-tile)),
    tile_cls(n) -> class of stream tile n, strict (tile retired one barrier early) else weak."""
    g = spec["g"]
    issued = retired = dma_n = 0
    bars = 0
    snap = [0]                      # snap[b] = VMEM ops retired when barrier b (1-based) was passed
    dma_index = []                  # VMEM issue index of every DMA instruction
    last_read_bar = {}              # tile -> number of barriers passed at its last read
    drained_bar = {}                # tile -> number of barriers passed when an lgkmcnt(0) behind its last read was executed
    errs = []
    stream = [("p", e) for e in prologue]
    for i in range(R):
        stream += [(i, e) for e in body]
    pend_wait = None
    for it, (kind, val) in stream:
        if kind == "wait":
            pend_wait = val if pend_wait is None else max(pend_wait, val)       # waits in exclusive tail branches: weakest one
            continue
        if pend_wait is not None and kind in ("dma", "vmem", "bar", "read"):
            retired = max(retired, issued - pend_wait)
            pend_wait = None
        if kind in ("dma", "vmem"):
            issued += 1
            if kind == "dma":
                tile = dma_n // g
                dma_n += 1
                dma_index.append(issued)
                prev = tile - spec["slot_reuse"](spec["tile_cls"](tile))
                if prev >= 0 and prev in last_read_bar and not (bars > last_read_bar[prev]):
                    errs.append(f"iteration {it}: tile {tile} staged into the slot of tile {prev} with no barrier since that tile's last read")
                elif prev >= 0 and prev in last_read_bar and spec.get("stagger"):
                    # wave groups run one barrier apart: a read of the LAGGING group at its local barrier count r sits between block-wide
                    # barrier events r+1 and r+2, a re-stage by the LEADING group at its local count d behind event d.  The replay walks ONE
                    # instruction stream (both groups run the same one), so with nb = d - r: nb >= 3 is safe; nb == 2 is safe only when the
                    # read was drained (s_waitcnt lgkmcnt(0)) in FRONT of the barrier that follows it; nb < 2 never.
                    nb = bars - last_read_bar[prev]
                    if nb < 3 and not (nb == 2 and drained_bar.get(prev) == last_read_bar[prev]):
                        errs.append(f"iteration {it}: staggered wave groups: tile {tile} staged into the slot of tile {prev} {nb} barrier(s) after its last "
                                    f"read (one fewer for the lagging group) and that read was not drained (s_waitcnt lgkmcnt(0)) in front of its barrier")
        elif kind == "bar":
            bars += 1
            snap.append(retired)
        elif kind == "drain":
            for t_, b_ in last_read_bar.items():
                if b_ == bars: drained_bar[t_] = bars              # reads of the current barrier interval are now in registers
        elif kind == "read" and it != "p":
            tiles = spec["need"](it, val)
            if tiles is None:
                continue
            if not isinstance(tiles, (tuple, list)): tiles = (tiles,)
            for t_ in tiles:
                last_read_bar[t_] = bars
                drained_bar.pop(t_, None)
            tile = max(tiles)                                       # FIFO: the youngest of them retired => all retired
            last = (tile + 1) * g
            if last > len(dma_index):
                errs.append(f"iteration {it}: {val} needs tile {tile} which was never issued"); continue
            need_ops = dma_index[last - 1]
            back = 1 if spec.get("strict", True) else 0
            if bars - back < 1 or snap[bars - back] < need_ops:
                errs.append(f"iteration {it}: {val} reads tile {tile} (VMEM op #{need_ops}) but only {snap[max(bars - back, 0)]} ops were retired "
                            f"{'one barrier before' if back else 'at'} the barrier that opens this phase")
    return errs, dma_n


# ---- specs -------------------------------------------------------------------------------------------------------------------
def flash2_need(i, cls):
    # stream: K0 V0 K1 | V(i+1) K(i+2) per iteration; QK^T reads (ds_read_b128) consume K(i), PV transpose reads consume V(i)
    if cls == "ds_read_b128": return 2 * i
    if cls == "ds_read_b64_tr_b16": return 2 * i + 1
    return None


def gemm_bw_need_factory():
    # diag_gemm_bw.hip: one K tile (32) per iteration, 8 DMA instructions per wave per tile; per iteration 8 fragment reads of (tile i, k-step 1)
    # in front of the barrier and 8 of (tile i+1, k-step 0) behind it
    state = {"i": -1, "n": 0}

    def need(i, cls):
        if cls != "ds_read_b128": return None
        if state["i"] != i: state.update(i=i, n=0)
        n = state["n"]; state["n"] += 1
        return i if n < 8 else i + 1
    return need


def gemm256_need_factory():
    # stream per K tile: HA0 HB0 HB1 HA1; phase 1 reads HA0 + HB0 (12 ds_read_b128), phase 2 HB1 (4), phase 3 HA1 (8)
    state = {"i": -1, "n": 0}

    def need(i, cls):
        if cls != "ds_read_b128": return None
        if state["i"] != i: state.update(i=i, n=0)
        n = state["n"]; state["n"] += 1
        if n < 8: return 4 * i              # HA0
        if n < 12: return 4 * i + 1         # HB0
        if n < 16: return 4 * i + 2         # HB1
        return 4 * i + 3                    # HA1
    return need


def gemm256_ph2_need_factory():
    # two-phase form (template argument PH = 2): phase A issues 16 fragment reads of HB0 + HA0 + HB1 (any order the compiler likes: they are
    # attributed to all three half-tiles), phase B 2 x MT1 reads of HA1
    state = {"i": -1, "n": 0}

    def need(i, cls):
        if cls != "ds_read_b128": return None
        if state["i"] != i: state.update(i=i, n=0)
        n = state["n"]; state["n"] += 1
        if n < 16: return (4 * i, 4 * i + 1, 4 * i + 2)
        return 4 * i + 3
    return need


SPECS = [
    ("attn_prefill", r"attn_prefill_flash2_kernel", dict(kind="fifo", g=4, need=flash2_need, tile_cls=lambda n: n & 1, slot_reuse=lambda c: 4, strict=True,
                                                         why="two barriers per tile: A(t) retires V(t), B(t) retires K(t+1)")),
    ("gemm256", r"gemm256_kernel.*ELi2EEv", dict(kind="fifo", g=2, need="ph2", tile_cls=lambda n: n & 3, slot_reuse=lambda c: 8, strict=True, stagger=True,
                                                 why="two-phase form: half-tiles retired by vmcnt(8) / vmcnt(6) one phase before they are read; a slot is re-staged one barrier "
                                                     "after its last read with the read drained (lgkmcnt(0)) in front of that barrier (wave groups one barrier apart)")),
    ("gemm256", r"gemm256_kernel", dict(kind="fifo", g=2, need=None, tile_cls=lambda n: n & 3, slot_reuse=lambda c: 8, strict=True, stagger=True,
                                        why="four-phase form: half-tiles retired by vmcnt(8) in the phase before they are read; a slot is re-staged >= 2 phases after its last read")),
    ("diag_gemm_bw", r"gemm_bw_kernel", dict(kind="fifo", g=8, need="bw", tile_cls=lambda n: 0, slot_reuse=lambda c: 4, strict=False,
                                            why="libplangen_diag.so experiment: 4-stage ring, tile t+1 retired by vmcnt(8) + the barrier of iteration t and first read behind that barrier (weak form)")),
    ("gemm", r"gemm_sk5_kernel", dict(kind="sk5")),
    ("diag_gemm", r"gemm_sk5_kernel", dict(kind="sk5")),
    ("gemm", r"gemm_sk4_kernel", dict(kind="sk4")),
    ("diag_gemm", r"gemm_sk4_kernel", dict(kind="sk4")),
    ("bench_kernels", r"gemm_sk4_kernel", dict(kind="sk4")),
    ("gemm", r"gemm_big_kernel", dict(kind="big", why="128x128 double buffer: vmcnt(0) + barrier retire tile t+1, a second barrier opens the phase that reads it "
                                      "(round 3; a third LDS slot would halve the kernel's residency, the extra barrier measured free)")),
    ("conv_halo", r"conv3x3_halo_kernelI.*Lb1ELb[01]E", dict(kind="halo_stag", why="production: 4-slot weight ring, W(t+1) retired by vmcnt(2) in phase t and read in phase t+1; "
                                                             "halo + W(0) retired two barriers before the first read")),
    ("conv_halo", r"conv3x3_halo_kernelI.*Lb0ELb[01]E", dict(kind="halo_lock", why="conv_halo=2 option (lock-step waves): vmcnt(4) + two barriers per K tile")),
    ("bench_kernels", r"dma_order_kernel", dict(kind="probe", why="measurement probe (tools/dma_order_probe.py), not on the product path")),
    ("conv_halo", r"conv3x3_out2_kernel", dict(kind="fifo", g=7, need="all", tile_cls=lambda n: 0, slot_reuse=lambda c: 2, strict=True,
                                               why="conv_out, 4 x 32 tiles: two patches ping-pong (7 DMA instructions per wave and tile); vmcnt(7 + Cout) + two barriers retire a patch "
                                                   "before its first read, a barrier behind the last read frees it for the fill after next")),
    ("conv_halo", r"conv3x3_out_halo_kernel", dict(kind="once", why="halo patch staged once per tile (conv_out, 128 -> 3 channels)")),
    ("bench_kernels", r"weight_prefetch_kernel", dict(kind="sink", why="background-load stressor of libplangen_diag.so (round 4's run-ahead weight prefetcher): the 1 KiB per-wave LDS sink is written by LDS-DMA and never read")),
    ("chain", r"chain_skel_kernel", dict(kind="probe", why="measured skeleton of the persistent decode chain (tools/chain_skel.py), not on the product path: gathers behind vmcnt(0) + s_barrier, LDS read is a stand-in")),
]


def main():
    import collections, json
    bad = 0
    seen = set()
    okc = collections.Counter()          # kernels whose protocol was verified, by spec kind (the machine-readable result)
    report = []
    names = sorted(f[:-4] for f in os.listdir(CSRC) if f.endswith(".hip")
                   and re.search(r"glds16|global_load_lds|gemm_skinny\.h", open(os.path.join(CSRC, f)).read()))      # gemm_skinny.h: the v4 decode GEMM template (gemm.hip = production, diag_gemm.hip = sweep / forensics instantiations)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=4) as ex:
        compiled = dict(zip(names, ex.map(compile_s, names)))
    for fname in names:
        funcs = functions(compiled[fname])
        for name, ev in funcs.items():
            if not any(k == "dma" for k, _ in ev):
                continue
            spec = next((s for f, rx, s in SPECS if f == fname and re.search(rx, name)), None)
            short = re.sub(r"^_Z\d+", "", name)[:70]
            if spec is None:
                bad += 1
                report.append(f"FAIL {fname}:{short}: LDS-DMA kernel without a protocol spec in tools/dma_isa_check.py")
                continue
            seen.add((fname, spec.get("kind")))
            if spec["kind"] == "fifo":
                pro, body = main_loop(ev)
                if body is None:
                    bad += 1; report.append(f"FAIL {fname}:{short}: no main loop with DMA + LDS reads found"); continue
                sp = dict(spec)
                if sp["need"] is None: sp["need"] = gemm256_need_factory()
                elif sp["need"] == "ph2": sp["need"] = gemm256_ph2_need_factory()
                elif sp["need"] == "bw": sp["need"] = gemm_bw_need_factory()
                elif sp["need"] == "all": sp["need"] = lambda i, cls: i if cls == "ds_read_b128" else None     # every fragment read of iteration i needs patch i (the weights were written by ds_write, retired before the loop)
                errs, n = replay(pro, body, sp)
                if errs:
                    bad += 1
                    report.append(f"FAIL {fname}:{short}: " + "; ".join(errs[:3]))
                else:
                    okc[spec["kind"] if sp.get("strict", True) else "fifo_weak"] += 1
                    report.append(f"ok   {fname}:{short}: fifo protocol holds in the {'strict' if sp.get('strict', True) else 'weak'} form over 6 replayed iterations ({n} DMA instructions)")
            elif spec["kind"] in ("sk4", "sk5"):
                continue
            elif spec["kind"] == "halo_stag":
                # steady state of the tap loop (two K tiles per iteration): [stage W(t+2) (round 6: first thing in the phase), 16 fragment reads, vmcnt(2), barrier, MFMAs, barrier];
                # replayed behind the protocol's own prologue (halo = tile 0 as one group, W(0), W(1), vmcnt(2), two barriers)
                pro, body = main_loop(ev)
                flat = [e for e in ev if e[0] in ("dma", "wait", "bar", "read")]
                first_read = next(k for k, e in enumerate(flat) if e[0] == "read")
                w2 = max(k for k, e in enumerate(flat[:first_read]) if e == ("wait", 2))
                nbar = sum(1 for e in flat[w2:first_read] if e[0] == "bar")
                nj = sum(1 for e in flat[:w2] if e[0] == "dma") - 4
                state = {"n": 0}

                def need(i, cls, state=state):
                    t = state["n"] // 16; state["n"] += 1
                    return 1 + t                                   # W(t) (the halo, tile 0, is older: FIFO)
                synth = [("dma", None)] * 6 + [("wait", 2), ("bar", None), ("bar", None)]      # halo as one 2-instruction group + W0 + W1
                errs, n = ([], 0) if body is None else replay(synth, body, dict(g=2, need=need, tile_cls=lambda n: 1 if n else 0,
                                                                                 slot_reuse=lambda c: 4 if c else 10 ** 6, strict=True))
                if body is None or errs or nbar < 2 or nj not in (4, 12):          # 12: ten patch rows x nine 4-pixel instructions over eight waves (round 6: rows padded to 36 pixels); 4: the upsample form
                    bad += 1
                    report.append(f"FAIL {fname}:{short}: " + ("; ".join(errs[:3]) if errs else f"prologue: {nbar} barrier(s) between the retiring vmcnt(2) and the first read, {nj} halo DMAs"))
                else:
                    okc[spec["kind"]] += 1; report.append(f"ok   {fname}:{short}: weight ring strict over 6 replayed iterations; halo ({nj} DMAs) + W(0) retired {nbar} barriers before the first read")
            elif spec["kind"] == "halo_lock":
                pro, body = main_loop(ev)
                state = {"n": 0}

                def need(i, cls, state=state):
                    t = state["n"] // 16; state["n"] += 1
                    return 1 + t
                synth = [("dma", None)] * 8                                  # halo as one 2-instruction group + W0, W1, W2
                errs, n = (["no main loop found"], 0) if body is None else replay(synth, body, dict(g=2, need=need, tile_cls=lambda n: 1 if n else 0,
                                                                                                    slot_reuse=lambda c: 4 if c else 10 ** 6, strict=True))
                if errs:
                    bad += 1; report.append(f"FAIL {fname}:{short}: " + "; ".join(errs[:3]))
                else:
                    okc[spec["kind"]] += 1; report.append(f"ok   {fname}:{short}: lock-step weight ring strict over 6 replayed iterations -- {spec['why']}")
            elif spec["kind"] == "big":
                # stage(t+1) | read tile t | vmcnt(0) + barrier (retires t+1) | barrier
                pro, body = main_loop(ev)
                if body is not None:
                    kinds = [k for k, _ in body]
                    if "dma" in kinds and "read" in kinds and kinds.index("dma") > kinds.index("read"):
                        k = kinds.index("dma")                  # hipcc rotated the loop (stage(t+1) sits behind the back edge's target and the
                        body = body[k:] + body[:k]              # prologue jumps into it): replay in execution order -- stage(t+1), read tile t, wait, barrier
                errs, n = (["no main loop found"], 0) if body is None else replay(pro, body, dict(g=8, need=lambda i, cls: i if cls == "ds_read_b128" else None,
                                                                                              tile_cls=lambda n: 0, slot_reuse=lambda c: 2, strict=True))
                if errs:
                    bad += 1
                    report.append(f"FAIL {fname}:{short}: " + "; ".join(errs[:3]))
                else:
                    okc[spec["kind"]] += 1; report.append(f"ok   {fname}:{short}: double buffer, strict form holds ({n} DMA instructions replayed) -- {spec['why']}")
            elif spec["kind"] == "sink":
                nread = sum(1 for e in ev if e[0] == "read")
                if nread:
                    bad += 1; report.append(f"FAIL {fname}:{short}: {nread} LDS read(s) in a kernel whose LDS-DMA target is declared a write-only sink")
                else:
                    okc[spec["kind"]] += 1; report.append(f"ok   {fname}:{short}: LDS-DMA sink never read -- {spec['why']}")
            elif spec["kind"] == "once":
                # per tile: fill (DMA) -> vmcnt(0) -> barrier -> barrier -> fragment reads -> barrier (patch free).  Linear walk of the tile loop.
                flat = [e for e in ev if e[0] in ("dma", "wait", "bar", "read")]
                last_dma = max(k for k, e in enumerate(flat) if e[0] == "dma")
                first_read = next(k for k, e in enumerate(flat) if e[0] == "read" and k > last_dma)
                between = flat[last_dma + 1:first_read]
                w0 = [k for k, e in enumerate(between) if e == ("wait", 0)]
                nbar = sum(1 for e in between[w0[-1]:] if e[0] == "bar") if w0 else 0
                after = flat[first_read:]
                if not w0 or nbar < 2 or not any(e[0] == "bar" for e in after):
                    bad += 1; report.append(f"FAIL {fname}:{short}: {nbar} barrier(s) between the retiring vmcnt(0) and the first read of the patch")
                else:
                    okc[spec["kind"]] += 1; report.append(f"ok   {fname}:{short}: patch retired {nbar} barriers before its first read, released by a barrier after the last -- {spec['why']}")
            else:
                report.append(f"note {fname}:{short}: {spec['kind']} -- {spec['why']}")
    sk4_checked = sk4_failed = 0; sk4_rc = 0
    for unit in ("gemm", "diag_gemm"):          # production instantiations (libplangen_hip.so) and the diagnostics library's variant table
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sk4_isa_check.py"), _S_PATH[unit]], capture_output=True, text=True)
        last = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else "no output"
        report.append(f"sk4 ({unit}.hip): " + last)
        if p.returncode != 0:
            bad += 1; sk4_rc = p.returncode
            report += p.stdout.strip().splitlines()[:5]
        m = re.match(r"(\d+) .*?(\d+) failed", last)
        if m:
            sk4_checked += int(m.group(1)); sk4_failed += int(m.group(2))
        else:
            sk4_checked = -1
    # v5 decode GEMM (round 6): exact replay of its hand-counted stream (tools/sk5_isa_check.py), production + diagnostics instantiations
    sk5_checked = sk5_failed = 0
    for unit in ("gemm", "diag_gemm"):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sk5_isa_check.py"), _S_PATH[unit]], capture_output=True, text=True)
        last = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else "no output"
        report.append(f"sk5 ({unit}.hip): " + last)
        m = re.match(r"(\d+) instantiations checked, (\d+) failed", last)
        if m:
            sk5_checked += int(m.group(1)); sk5_failed += int(m.group(2))
        if p.returncode != 0 and unit == "gemm" or (m and int(m.group(2))):
            bad += 1
            report += [l for l in p.stdout.strip().splitlines() if l.startswith("FAIL")][:5]
    print("\n".join(report))
    # one machine-readable line for tests/test_isa_check.py (wording of the lines above is free to change)
    print("SUMMARY " + json.dumps({"failed": bad, "verified": dict(okc), "notes": sum(1 for l in report if l.startswith("note")),
                                   "sk4": {"checked": sk4_checked, "failed": sk4_failed, "rc": sk4_rc}, "sk5": {"checked": sk5_checked, "failed": sk5_failed},
                                   "compiler": _compiler_id()[:80]}))
    print(f"{bad} failed")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
