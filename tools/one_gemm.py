#!/usr/bin/env python3
"""Run one decode-GEMM shape/variant in a loop (for rocprofv3 --pmc). usage: one_gemm.py M N K variant S [iters]"""
import ctypes as C, os, sys
import torch  # noqa: F401
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plangen_amd", "lib", "libplangen_diag.so"))      # diagnostics library (pg_bench_* live there, not in the product)
lib.pg_bench_skinny.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float)]
M, N, K, v, S = (int(a) for a in sys.argv[1:6])
it = int(sys.argv[6]) if len(sys.argv) > 6 else 30
us = C.c_float(0)
rc = lib.pg_bench_skinny(M, N, K, v, S, it, 0, C.byref(us))
print("rc", rc, "us", us.value)
