#!/usr/bin/env python3
"""Does a younger register load retire before an older LDS-DMA load of the same wave?  (pg_bench_dma_order)"""
import ctypes as C, os
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "plangen_amd", "lib", "libplangen_diag.so"))      # diagnostics library (pg_bench_* live there, not in the product)
lib.pg_bench_dma_order.argtypes = [C.c_int] * 3 + [C.POINTER(C.c_uint)] * 2
for blocks in (256, 1024):
    for mode in (0, 1, 4, 8, 16, 17, 20):
        for rep in range(2):
            f, n = C.c_uint(0), C.c_uint(0)
            rc = lib.pg_bench_dma_order(blocks, 200, mode, C.byref(f), C.byref(n))
            print(f"blocks {blocks} younger op {'4-page DMA, ' if mode & 16 else ''}{'default, NO wait (positive control)' if mode & 8 else 'second LDS-DMA' if mode & 4 else 'nt' if mode & 1 else 'default'}: rc {rc}  stale reads {f.value} of {n.value} lane checks", flush=True)
