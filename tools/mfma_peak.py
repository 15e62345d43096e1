#!/usr/bin/env python3
"""MFMA-only probe: sustained dense bf16 rate of the matrix cores with no memory traffic, by MFMA shape, waves per SIMD and operand entropy.
usage: mfma_peak.py [iters=20000]"""
import ctypes as C, os, sys
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "plangen_amd", "lib", "libplangen_diag.so"))
lib.pg_bench_mfma_peak.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_float)] * 2
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
print("| MFMA | waves / SIMD | operands | TFLOP/s | of 2.5 PF | ms per launch |\n|---|---|---|---|---|---|")
for mode, name in ((0, "16x16x32 bf16"), (1, "32x32x16 bf16"), (2, "32x32x16, LDS-fed 128x128 per wave")):
    for wps in ((1,) if mode == 2 else (1, 2)):
        for const in (0, 1):
            tf, ms = C.c_float(0), C.c_float(0)
            rc = lib.pg_bench_mfma_peak(mode, wps, const, iters // 16 if mode == 2 else iters, C.byref(tf), C.byref(ms))
            print(f"| {name} | {wps} | {'constant' if const else 'random'} | {tf.value:.0f} | {tf.value / 2500:.3f} | {ms.value:.2f} | (rc {rc})", flush=True)
