#!/usr/bin/env python3
"""Per-wave timeline of ONE launch of the bs=64 production wide-N decode GEMM (gemm_skinny3_kernel<4, NCK, 2, true, EPI, 8, true>: 64 rows x 128
columns, 8 waves, tiled W, register ring of 2) from cycle stamps compiled into a diagnostics-library instantiation (where do
gate|up's 5 us above the launch floor go).   usage: sk3_profile.py [500|501|502]      500 gate|up + SwiGLU, 501 qkv (S = 2), 502 gate|up ring 4"""
import ctypes as C, os, sys
import numpy as np
import torch  # noqa: F401  (loads the HIP runtime the library needs)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "plangen_amd", "lib", "libplangen_diag.so"))
v = int(sys.argv[1]) if len(sys.argv) > 1 else 500
N, K, S, NCK = {500: (11264, 2048, 1, 16), 501: (6144, 2048, 2, 8), 502: (11264, 2048, 1, 16), 503: (11264, 2048, 1, 16)}[v]
M, maxw = 128, 4096
buf = np.zeros((maxw, 64), dtype=np.uint64)
lib.pg_bench_sk4_profile.argtypes = [C.c_int] * 5 + [C.c_void_p, C.c_int]
rc = lib.pg_bench_sk4_profile(M, N, K, v, S, buf.ctypes.data, maxw)
b = buf[buf[:, 0] != 0].astype(np.int64)
nst = int((b[0] != 0).sum())
b = b[:, :nst]
# ticks -> us: earlier rounds (tools/sk4_profile.py) read this counter as the ~2.1 GHz shader clock; SK3_TICKS_PER_US overrides.  The script prints the raw
# tick count of a whole wave first, so the scale can be checked against the kernel's rocprofv3 duration (gate|up 17.4 us, qkv 10.7 us at bs=64).
CLK = float(os.environ.get("SK3_TICKS_PER_US", "2100"))
print(f"raw ticks, whole wave: median {np.median((b[:, -1] - b[:, 0])):.0f} (p90 {np.percentile((b[:, -1] - b[:, 0]), 90):.0f}); scale used: {CLK:.0f} ticks per us")
rel = (b - b[:, :1]) / CLK
d = np.diff(rel, axis=1)
print(f"variant {v}: rc={rc}, {len(b)} waves ({len(b) // 8} blocks x 8), {nst} stamps per wave, NCK {NCK}; times in us (constant 100 MHz counter)")
# launch skew: when did each wave start relative to the first wave of its XCD-agnostic launch (counter bases differ per XCD: only spread within a block is exact)
names = ["W ring + x(0) issued", "x(0) landed + staged", "first barrier"]
for k, nm in enumerate(names):
    print(f"  prologue {nm:24s}: median {np.median(d[:, k]):6.2f}  p90 {np.percentile(d[:, k], 90):6.2f}")
per = {"ds_read + MFMA issue": [3 + 3 * c for c in range(NCK)], "wait x(c+1) / W(c+1) + LDS store": [4 + 3 * c for c in range(NCK)], "barrier": [5 + 3 * c for c in range(NCK)]}
for nm, cols in per.items():
    cols = [c for c in cols if c < d.shape[1]]
    body = cols[1:-1] if len(cols) > 2 else cols
    print(f"  per chunk {nm:34s}: median {np.median(d[:, body]):5.2f}  mean {d[:, body].mean():5.2f}   per-wave total {d[:, cols].sum(axis=1).mean():6.2f}")
print(f"  epilogue (SwiGLU / transposed stores issued): median {np.median(d[:, -2]):5.2f}; stores acknowledged: {np.median(d[:, -1]):5.2f}")
print(f"  whole wave: median {np.median(rel[:, -1]):6.2f}  p10 {np.percentile(rel[:, -1], 10):6.2f}  p90 {np.percentile(rel[:, -1], 90):6.2f}  max {rel[:, -1].max():6.2f}")
print("  chunk-done times (median, us since wave start): " + " ".join(f"{np.median(rel[:, 6 + 3 * c]):.2f}" for c in range(NCK) if 6 + 3 * c < rel.shape[1]))
