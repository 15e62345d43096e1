#!/usr/bin/env python3
"""Floor of the persistent decode-chain structure at 16 rows (plangen_amd/csrc/chain.hip): us per launch of the skeleton by component.
mode bits: 1 three grid barriers, 2 run-ahead weight stream (352 / 464 KiB per CU), 4 activation gathers + result publishes."""
import ctypes as C, os, sys
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "plangen_amd", "lib", "libplangen_diag.so"))      # diagnostics library (pg_bench_* live there, not in the product)
lib.pg_bench_chain_skeleton.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint)]
names = {0: "launch only", 1: "3 grid barriers", 2: "weight stream", 4: "gathers + publishes", 3: "barriers + stream", 5: "barriers + gathers/publishes",
         6: "stream + gathers/publishes", 7: "everything"}
for rep in range(2):
    for mode in (0, 1, 2, 4, 3, 5):        # 6 / 7 (stream + gathers in one build) fault: hipcc recycles a ring register whose asm load is still in flight (guide 5.7 item 1) -- the skeleton's numbers do not need them
        us, err = C.c_float(0), C.c_uint(0)
        rc = lib.pg_bench_chain_skeleton(6, 300, mode, C.byref(us), C.byref(err))
        print(f"mode {mode} ({names[mode]:32s}): rc {rc}  {us.value:7.2f} us per launch  barrier give-ups {err.value}", flush=True)
