import ctypes as C, os, sys
import torch
lib = C.CDLL("/root/repo/plangen_amd/lib/libplangen_diag.so")
lib.pg_bench_skinny.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float)]
for name, (N, K, S) in {"qkv": (6144, 2048, 2), "gu": (11264, 2048, 1), "gh2": (16384, 2048, 1)}.items():
    for v in (400, 401, 402, 403, 400):
        us = C.c_float(0)
        rc = lib.pg_bench_skinny(128, N, K, v, S, 300, 0, C.byref(us))
        print(f"{name} S={S} variant {v}: rc {rc} {us.value:6.2f} us", flush=True)
