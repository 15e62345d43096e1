#!/usr/bin/env python3
"""x prefetched two chunks ahead in the v3 wide-N decode block (XA = 2): microbenchmark against the production form, chains of dependent launches on rotating weights."""
import ctypes as C, os
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "plangen_amd", "lib", "libplangen_diag.so"))
lib.pg_bench_skinny.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float)]
M = 128
for name, N, K, S, variants in [("gate|up + SwiGLU", 11264, 2048, 1, [(509, "production (XA 1, ring 2)"), (504, "XA 2, ring 2"), (505, "XA 2, ring 3"), (520, "96-col blocks (236)"), (522, "96-col, XA 2 ring 3"), (524, "64-col blocks (352)"), (530, "production + nt W loads")]),
                                ("qkv slabs S=2", 6144, 2048, 2, [(400, "production (XA 1, ring 2)"), (506, "XA 2, ring 2"), (507, "XA 2, ring 3"), (521, "96-col blocks (256)"), (523, "96-col, XA 2 ring 3"), (531, "production + nt W loads")])]:
    for rep in range(2):
        for v, what in variants:
            us = C.c_float(0)
            rc = lib.pg_bench_skinny(M, N, K, v, S, 400, 0, C.byref(us))
            print(f"{name:18s} pass {rep} variant {v} {what:28s} rc {rc} {us.value:6.2f} us  {N * K * 2 / us.value * 1e-6:5.2f} TB/s")
