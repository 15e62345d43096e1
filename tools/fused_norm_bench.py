#!/usr/bin/env python3
"""Round 5 experiment: split-K producer GEMM with the consumer norm's reduction folded into its tail (gemm_skinny.h EPI 5,
last-arrival ticket per tile, fixed-point atomics for the row sums of squares) against the production pair GEMM + rmsnorm512.
usage: fused_norm_bench.py [iters]"""
import ctypes as C, os, sys
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "plangen_amd", "lib", "libplangen_diag.so"))
lib.pg_bench_fused_norm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)] * 3 + [C.POINTER(C.c_uint)]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
print("| rows | shape | S | GEMM + rmsnorm512 us | GEMM alone us | fused producer us | saved us | x / xw / ssq mismatches |\n|---|---|---|---|---|---|---|---|")
for M, So, Sd in ((16, 4, 4), (32, 8, 11), (64, 8, 11), (128, 4, 4)):
    for name, N, K, S in (("o 2048x2048", 2048, 2048, So), ("down 2048x5632", 2048, 5632, Sd)):
        for rep in range(2):
            a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
            bad = (C.c_uint * 4)()
            rc = lib.pg_bench_fused_norm(M, N, K, S, iters, C.byref(a), C.byref(b), C.byref(c), bad)
            print(f"| {M} | {name} | {S} | {a.value:.2f} | {b.value:.2f} | {c.value:.2f} | {a.value - c.value:+.2f} | {bad[0]} / {bad[1]} / {bad[2]} (rc {rc}) |", flush=True)
