#!/usr/bin/env python3
"""Time / verify the MFMA GEMM kernels on one shape.
usage: big_gemm.py M N K [mode=1] [iters=20] [verify=1]            plain A
       big_gemm.py conv B Hi Wi Cin Cout up [mode=1] [iters] [verify]   implicit-im2col 3x3 conv
mode = pg_set_option("gemm256"): 0 = 128x128 kernel, 1 = 256x256 kernel (tile height per launch, two phases per K tile), 4 / 5 / 6 = 256 / 224 / 192 rows pinned, +8 = four phases"""
import ctypes as C, os, sys
import torch  # noqa: F401  (one HIP runtime per process)
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plangen_amd", "lib", "libplangen_diag.so"))      # diagnostics library (pg_bench_* live there, not in the product)
lib.pg_bench_gemm.argtypes = [C.c_int] * 10 + [C.POINTER(C.c_float)] * 2
a = sys.argv[1:]
if a[0] == "conv":
    B, Hi, Wi, Cin, Cout, up = (int(x) for x in a[1:7]); rest = a[7:]
    M, N, K = B * (Hi << up) * (Wi << up), Cout, 9 * Cin
else:
    M, N, K = (int(x) for x in a[0:3]); rest = a[3:]; Hi = Wi = Cin = up = 0
mode = int(rest[0]) if len(rest) > 0 else 1
iters = int(rest[1]) if len(rest) > 1 else 20
verify = int(rest[2]) if len(rest) > 2 else 1
us, md = C.c_float(0), C.c_float(-1)
rc = lib.pg_bench_gemm(M, N, K, Hi, Wi, Cin, up, mode, iters, verify, C.byref(us), C.byref(md))
tf = 2.0 * M * N * K / (us.value * 1e-6) / 1e12 if us.value > 0 else 0
print(f"rc {rc} M {M} N {N} K {K} mode {mode}: {us.value:9.1f} us  {tf:7.1f} TF/s  maxdiff {md.value:.3g}")
