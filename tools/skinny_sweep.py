#!/usr/bin/env python3
"""Sweep decode-GEMM variants x split-K on the GPU box (tuning aid).  usage: skinny_sweep.py [M]"""
import ctypes as C
import os
import sys

import torch  # noqa: F401  (HIP runtime load order)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "plangen_amd", "lib", "libplangen_diag.so"))      # diagnostics library (pg_bench_* live there, not in the product)
lib.pg_bench_skinny.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float)]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
BK = {0: 128, 1: 128, 2: 256, 3: 256, 4: 256, 5: 128, 6: 256, 7: 128, 8: 128, 9: 128, 10: 256, 11: 64, 20: 128, 21: 128, 22: 128, 23: 128, 24: 128, 25: 128, 30: 128, 31: 128, 32: 128, 33: 128, 34: 128, 35: 128, 36: 128, 37: 128, 38: 128, 26: 128, 27: 128, 40: 64, 50: 128, 51: 128, 52: 128, 53: 128, 54: 128, 60: 128, 61: 128, 62: 128}
shapes = {"qkv": (6144, 2048), "o": (2048, 2048), "gu": (11264, 2048), "down": (2048, 5632), "gh2": (16384, 2048)}
variants = [int(v) for v in os.environ.get("VARIANTS", "0,1,2,3,4,5,6,7,8,9,10,11").split(",")]
for name, (N, K) in shapes.items():
    wmb = N * K * 2 / 1e6
    print(f"--- {name}: N={N} K={K} W={wmb:.1f} MB  ideal@6TB/s={wmb / 6:.2f} us", flush=True)
    for v in variants:
        row = []
        for S in (1, 2, 4, 8, 11, 16):
            if (K // BK[v]) % S or K // BK[v] < S:
                continue
            for cons in (0, 1):
                us = C.c_float(0)
                rc = lib.pg_bench_skinny(M, N, K, v, S, 200, cons, C.byref(us))
                row.append(f"S{S}{'+n' if cons else ''}:{us.value:6.1f}" if rc == 0 else f"S{S}:ERR{rc}")
        print(f"  v{v:2d}  " + "  ".join(row), flush=True)
