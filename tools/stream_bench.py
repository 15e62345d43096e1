#!/usr/bin/env python3
"""Pure streaming-read calibration on the GPU box (ceiling for the HBM-bound kernels)."""
import ctypes as C, os
import torch  # noqa: F401
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plangen_amd", "lib", "libplangen_diag.so"))      # diagnostics library (pg_bench_* live there, not in the product)
lib.pg_bench_stream.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
for mb in (64, 411, 1024):
    for blocks in (1024, 2048, 4096, 8192):
        for nt in (0, 1):
            us = C.c_float(0)
            lib.pg_bench_stream(mb * 1000 * 1000 // (16 * blocks) * (16 * blocks), blocks, 50, nt, C.byref(us))
            print(f"{mb:5d} MB blocks {blocks:5d} nt {nt}: {us.value:8.1f} us  {mb * 1e6 / us.value / 1e6:7.2f} TB/s", flush=True)
