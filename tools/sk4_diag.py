#!/usr/bin/env python3
"""Race diagnosis for a decode-GEMM variant: per-launch compare, positions of the wrong elements of the first bad launch.
usage: sk4_diag.py M reps batches variant [shape ...]   shapes: qkv2 qkv4 o4 gu1 down4 ..."""
import ctypes as C, os, sys
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "plangen_amd", "lib", "libplangen_diag.so"))      # diagnostics library (pg_bench_* live there, not in the product)
lib.pg_bench_skinny_diag.argtypes = [C.c_int] * 6 + [C.c_uint, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
M, reps, batches, v = (int(a) for a in sys.argv[1:5])
dims = {"qkv": (6144, 2048), "o": (2048, 2048), "gu": (11264, 2048), "down": (2048, 5632), "gh": (16384, 2048)}
for sh in sys.argv[5:] or ["qkv2"]:
    name, S = sh.rstrip("0123456789"), int(sh[len(sh.rstrip("0123456789")):])
    N, K = dims[name]
    tot = 0
    for b in range(batches):
        bad, npos = C.c_int(0), C.c_int(0)
        pos = (C.c_int * 40000)()
        rc = lib.pg_bench_skinny_diag(M, N, K, v, S, reps, (0 if os.environ.get('PG_DIAG_SEED0') else b * 101), C.byref(bad), pos, 20000, C.byref(npos))
        if rc != 0:
            print(f"{sh}: rc {rc}"); break
        tot += bad.value
        if bad.value:
            pts = [(pos[2 * i], pos[2 * i + 1]) for i in range(npos.value)]
            reps_bad = [c for r, c in pts if r == -2]
            tiles = [(r, c) for r, c in pts if r >= 0]
            rows = sorted({r for r, c in tiles}); cts = sorted({c for r, c in tiles})
            first = [c for r, c in pts if r == -1]
            print(f"{sh} v{v} batch {b}: {bad.value}/{reps} bad launches {reps_bad[:24]}; first saved {first}: {len(tiles)} wrong (row, n-tile) cells, rows {rows} n-tiles {cts[:40]}{'...' if len(cts) > 40 else ''} ({len(cts)} distinct)", flush=True)
    print(f"{sh} v{v}: {tot} bad launches in {batches} x {reps}", flush=True)
