#!/usr/bin/env python3
"""Decode-GEMM microbenchmark: production v3 (tiled) vs the v4 LDS-DMA variants, correctness screen first.
usage: sk4_sweep.py [M]   (env VARIANTS=51,70,...)"""
import ctypes as C
import os
import sys

import torch  # noqa: F401  (HIP runtime load order)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "plangen_amd", "lib", "libplangen_diag.so"))      # diagnostics library (pg_bench_* live there, not in the product)
lib.pg_bench_skinny.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float)]
lib.pg_bench_skinny_verify.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float)] * 2
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
shapes = {"qkv": (6144, 2048, (1, 2, 4)), "o": (2048, 2048, (2, 4, 8)), "gu": (11264, 2048, (1, 2)), "down": (2048, 5632, (4, 11)), "gh2": (16384, 2048, (1, 2))}
variants = [int(v) for v in os.environ.get("VARIANTS", "51,70,71,72,73,74,75,76,77").split(",")]
tiled = lambda v: v >= 50
for name, (N, K, Ss) in shapes.items():
    wmb = N * K * 2 / 1e6
    print(f"--- {name}: M={M} N={N} K={K} W={wmb:.1f} MB  ideal@6.3TB/s={wmb / 6.3:.2f} us", flush=True)
    for v in variants:
        row = []
        for S in Ss:
            md, mr = C.c_float(0), C.c_float(0)
            rc = lib.pg_bench_skinny_verify(M, N, K, v, S, int(tiled(v)), 3, C.byref(md), C.byref(mr))
            ok = "ok" if rc == 0 and md.value <= 2e-3 * max(mr.value, 1e-6) else ("n/a" if rc == -1 else f"BAD({md.value:.3g}/{mr.value:.3g})")
            if rc == -1:
                continue
            for cons in (0, 1):
                us = C.c_float(0)
                rc = lib.pg_bench_skinny(M, N, K, v, S, 300, cons, C.byref(us))
                row.append(f"S{S}{'+n' if cons else ''}:{us.value:6.2f}" if rc == 0 else f"S{S}:ERR{rc}")
            row.append(ok)
        print(f"  v{v:2d}  " + "  ".join(row), flush=True)
