#!/usr/bin/env python3
"""Microbenchmark of the wide-N decode GEMMs at 128 rows: the v5 kernel (gemm_sk5_kernel: n-tile pairs x two K halves, x by LDS-DMA) against the
round-5 v3 blocks, chains of dependent launches on rotating weights (> 600 MB), us per launch and weight-stream GB/s.   usage: sk5_bench.py"""
import ctypes as C, os
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "plangen_amd", "lib", "libplangen_diag.so"))
lib.pg_bench_skinny.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float)]
lib.pg_bench_skinny_verify.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float)] * 2
M = 128
cases = [("qkv", 6144, 2048, 2, [(400, "v3 64x128 blocks, ring 2 (round 5)"), (511, "v5")]),
         ("gate|up (slab form, S=1)", 11264, 2048, 1, [(400, "v3 64x128 blocks, ring 2"), (511, "v5")]),
         ("gate|up + SwiGLU", 11264, 2048, 1, [(500, "v3 production block, STAMPED build"), (510, "v5 SwiGLU")]),
         ("gen_head 2", 16384, 2048, 1, [(400, "v3"), (511, "v5")]),
         ("lm_head", 102400, 2048, 1, [(400, "v3"), (511, "v5")])]
for name, N, K, S, variants in cases:
    mb = N * K * 2 / 1e6
    for v, what in variants:
        if v in (511,):
            md, mr = C.c_float(0), C.c_float(0)
            rc = lib.pg_bench_skinny_verify(M, N, K, v, S, 1, 3, C.byref(md), C.byref(mr))
            ver = f"verify vs v1: max diff {md.value:.2e} of max |ref| {mr.value:.2f} (rc {rc})"
        else:
            ver = ""
        for cons in (0, 1):
            us = C.c_float(0)
            rc = lib.pg_bench_skinny(M, N, K, v, S, 300, cons if v not in (500, 510) else 0, C.byref(us))
            if v in (500, 510) and cons:
                continue
            print(f"{name:26s} {mb:6.1f} MB  variant {v} {what:40s} {'+ rmsnorm consumer' if cons else 'alone':18s} rc {rc} {us.value:7.2f} us  {mb / us.value * 1e-3 if not cons else 0:5.2f} TB/s  {ver}")
