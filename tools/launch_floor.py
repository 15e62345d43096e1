#!/usr/bin/env python3
"""What a short decode-GEMM launch is made of: chains of identical dependent launches on one stream (weights rotated through
> 600 MB so the Infinity Cache cannot hold them), us per launch.   usage: launch_floor.py [o|down|qkv]
  300 production 64-row LDS-DMA block | 320 EMPTY kernel, same geometry | 321 weight stream only | 322 no stores | 323 no x tile | 324 no MFMA"""
import ctypes as C, os, sys
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "plangen_amd", "lib", "libplangen_diag.so"))
lib.pg_bench_skinny.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float)]
shape = sys.argv[1] if len(sys.argv) > 1 else "o"
N, K, S = {"o": (2048, 2048, 4), "down": (2048, 5632, 4), "qkv": (6144, 2048, 2)}[shape]
M = 128
names = {300: "production block", 320: "empty kernel, same geometry", 321: "weight stream only", 322: "no stores", 323: "no x tile", 324: "no MFMA"}
print(f"{shape}: M {M} N {N} K {K} S {S}: {N * K * 2 / 1e6:.1f} MB of weights, grid ({N // 64}, {S}, {M // 64}) x 256 threads")
for rep in range(2):
    for v, nm in names.items():
        us = C.c_float(0)
        rc = lib.pg_bench_skinny(M, N, K, v, S, 400, 0, C.byref(us))
        print(f"  pass {rep}  variant {v} {nm:30s} rc {rc}  {us.value:6.2f} us per launch")
for v, nm in ((300, "production block"), (320, "empty kernel")):
    us = C.c_float(0)
    lib.pg_bench_skinny(M, N, K, v, S, 400, 1, C.byref(us))
    print(f"  with the consumer rmsnorm512 behind every launch: variant {v} {nm:20s} {us.value:6.2f} us per pair")
