#!/usr/bin/env python3
"""Floor for a short HBM-bound kernel: pure 16-byte streaming reads of the decode-GEMM weight sizes."""
import ctypes as C, os
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "plangen_amd", "lib", "libplangen_diag.so"))      # diagnostics library (pg_bench_* live there, not in the product)
lib.pg_bench_stream.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
for mb in (8.4, 23.1, 25.2, 46.1, 67.1, 256.0):
    for nt in (0, 1):
        row = []
        for blocks in (256, 512, 1024, 2048, 4096):
            us = C.c_float(0)
            b = int(mb * 1e6) // (16 * blocks * 256) * (16 * blocks * 256)
            if b == 0:
                continue
            lib.pg_bench_stream(b, blocks, 200, nt, C.byref(us))
            row.append(f"{blocks}:{us.value:6.2f}us({b / us.value / 1e6:5.2f}TB/s)")
        print(f"{mb:6.1f} MB nt={nt}  " + "  ".join(row), flush=True)
