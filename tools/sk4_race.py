#!/usr/bin/env python3
"""Race screen: many repetitions of a decode-GEMM variant against the v1 kernel.  usage: sk4_race.py M reps variants..."""
import ctypes as C, os, sys
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "plangen_amd", "lib", "libplangen_diag.so"))      # diagnostics library (pg_bench_* live there, not in the product)
lib.pg_bench_skinny_verify.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float)] * 2
M, reps = int(sys.argv[1]), int(sys.argv[2])
for v in [int(a) for a in sys.argv[3:]]:
    for name, (N, K, Ss) in {"qkv": (6144, 2048, (2, 4)), "o": (2048, 2048, (4, 8)), "gu": (11264, 2048, (1, 2)), "down": (2048, 5632, (4, 11))}.items():
        for S in Ss:
            bad = 0; vals = []
            for trial in range(8):
                md, mr = C.c_float(0), C.c_float(0)
                rc = lib.pg_bench_skinny_verify(M, N, K, v, S, 1, reps, C.byref(md), C.byref(mr))
                if rc != 0:
                    bad = -1; break
                bad += md.value > 2e-3 * mr.value
                if md.value > 2e-3 * mr.value: vals.append((md.value, mr.value))
            print(f"M={M} v{v} {name} S={S}: {'unsupported' if bad < 0 else f'{bad}/8 batches of {reps} launches had a wrong element'} {vals if bad > 0 else ''}", flush=True)
