#!/usr/bin/env python3
"""Hazard screen of the decode GEMMs UNDER CONCURRENT MEMORY LOAD (round 4).  The production dispatch on the tiled weights
(variant 2) -- or any variant -- runs `reps` launches per batch against the v1 kernel while a background kernel on another stream
streams a private buffer (mode 0: through LDS-DMA, 1: LDS-DMA nt, 2: plain register loads, -1: no background).
usage: sk4_load_stress.py M reps [variant ...]      env: BG_MODES="-1 0 2"  BG_BLOCKS=256 BG_DEPTH=16"""
import ctypes as C, os, sys
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "plangen_amd", "lib", "libplangen_diag.so"))      # diagnostics library (pg_bench_* live there, not in the product)
lib.pg_bench_skinny_verify.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float)] * 2
M, reps = int(sys.argv[1]), int(sys.argv[2])
variants = [int(a) for a in sys.argv[3:]] or [2]
modes = [int(v) for v in os.environ.get("BG_MODES", "-1 0 2").split()]
blocks, depth = int(os.environ.get("BG_BLOCKS", "256")), int(os.environ.get("BG_DEPTH", "16"))
shapes = {"qkv": (6144, 2048, (2,)), "o": (2048, 2048, (4,)), "gu": (11264, 2048, (1,)), "down": (2048, 5632, (4,))}
if os.environ.get("SHAPES"):
    shapes = {k: v for k, v in shapes.items() if k in os.environ["SHAPES"].split()}
for mode in modes:
    for v in variants:
        for name, (N, K, Ss) in shapes.items():
            for S in Ss:
                bad = 0; vals = []; overlapped = 0
                for trial in range(6):
                    if mode >= 0:
                        assert lib.pg_bench_background(768, 40, blocks, depth, mode) == 0       # ~25-40 ms of streaming
                    md, mr = C.c_float(0), C.c_float(0)
                    rc = lib.pg_bench_skinny_verify(M, N, K, v, S, 1, reps, C.byref(md), C.byref(mr))
                    if mode >= 0:
                        overlapped += 1 - lib.pg_bench_background_done()              # still running after the batch = the batch ran under load
                        lib.pg_bench_background_join()
                    if rc != 0:
                        bad = -1; break
                    if md.value > 2e-3 * mr.value:
                        bad += 1; vals.append((round(md.value, 4), round(mr.value, 3)))
                print(f"bg {mode:2d} M={M} v{v} {name} S={S}: " + ("unsupported" if bad < 0 else f"{bad}/6 batches of {reps} launches had a wrong element (bg still running after {overlapped}/6)") + (f" {vals}" if bad > 0 else ""), flush=True)
