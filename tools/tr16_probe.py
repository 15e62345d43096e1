#!/usr/bin/env python3
"""ds_read_b64_tr_b16 semantics: which LDS elements does each lane receive for a given per-lane address pattern?"""
import ctypes as C, os
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "plangen_amd", "lib", "libplangen_diag.so"))      # diagnostics library (pg_bench_* live there, not in the product)
def run(addrs, title):
    a = (C.c_int * 64)(*addrs); o = (C.c_ushort * 256)()
    rc = lib.pg_bench_tr16_probe(a, o)
    print(f"--- {title} (rc {rc})")
    for l in range(64):
        if l < 20 or l % 16 == 0:
            print(f"lane {l:2d} addr {addrs[l]:5d} (elem {addrs[l] // 2:4d}) -> {[o[l * 4 + j] for j in range(4)]}")
# pattern A: rows of 256 B (128 elements): lane i of a 16-lane group -> row i // 4, 4 consecutive elements at column (i % 4) * 4; groups 16 columns apart
run([((l & 15) // 4) * 256 + ((l & 15) % 4) * 8 + (l >> 4) * 32 for l in range(64)], "4 rows x 16 cols per group, row stride 256 B")
# pattern B: all lanes the same address
run([64] * 64, "uniform address 64")
# pattern C: lane-linear 8 B apart
run([l * 8 for l in range(64)], "lane-linear 8 B")
