#!/usr/bin/env python3
"""Per-wave timeline of one decode-GEMM launch (s_memtime stamps).  usage: sk4_profile.py variant S [shape]
variants with stamps compiled in: 271 (64-row blocks), 274 (128-row blocks)."""
import ctypes as C, os, sys
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "plangen_amd", "lib", "libplangen_diag.so"))      # diagnostics library (pg_bench_* live there, not in the product)
v, S = int(sys.argv[1]), int(sys.argv[2])
shape = sys.argv[3] if len(sys.argv) > 3 else "qkv"
N, K = {"qkv": (6144, 2048), "o": (2048, 2048), "gu": (11264, 2048), "down": (2048, 5632)}[shape]
M = 128
maxw = 8192
buf = np.zeros((maxw, 64), dtype=np.uint64)
lib.pg_bench_sk4_profile.argtypes = [C.c_int] * 5 + [C.c_void_p, C.c_int]
rc = lib.pg_bench_sk4_profile(M, N, K, v, S, buf.ctypes.data, maxw)
used = buf[:, 0] != 0
b = buf[used].astype(np.int64)
nst = int((b[0] != 0).sum())
b = b[:, :nst]
rel = (b - b[:, :1]) / 2100.0    # per-wave deltas (s_memtime bases differ between XCDs); ~2.1 GHz shader clock -> us (approximate)
print(f"variant {v} S={S} {shape}: rc={rc} waves={len(b)} stamps/wave={nst}; times in us at an assumed 2.1 GHz")
names = ["start", "issued"] + sum([[f"c{c}.Xlanded", f"c{c}.barrier", f"c{c}.Wlanded", f"c{c}.done"] for c in range((nst - 3) // 4)], []) + ["end"]
d = np.diff(rel, axis=1)
nc = (nst - 3) // 4
for k, nm in enumerate(["wait X", "barrier", "wait W (+ issue X)", "ds_read + MFMA (+ issue W)"]):
    cols = [1 + 4 * c + k for c in range(1, nc)]          # skip chunk 0 (cold)
    print(f"  per-chunk {nm:28s}: median {np.median(d[:, cols]):5.2f} us  mean {d[:, cols].mean():5.2f}  (chunk 0: {np.median(d[:, 1 + k]):5.2f})")
tot = [np.mean(d[:, [1 + 4 * c + k for c in range(nc)]].sum(axis=1)) for k in range(4)]
print("  per-wave totals over all chunks (mean): wait X %.2f  barrier %.2f  wait W %.2f  ds_read+MFMA %.2f us" % tuple(tot))
print("  chunk-done times (median, us since wave start): " + " ".join(f"{np.median(rel[:, 1 + 4 * c + 4]):.2f}" for c in range(nc)))
print(f"  prologue issue {np.median(d[:, 0]):5.2f} us; epilogue {np.median(d[:, -1]):5.2f} us; whole wave {np.median(rel[:, -1]):5.2f} us (p90 {np.percentile(rel[:, -1], 90):5.2f})")
sys.exit(0)
for i in range(nst):
    col = rel[:, i]
    print(f"  {names[i] if i < len(names) else i:12s} median {np.median(col):6.2f}  p10 {np.percentile(col, 10):6.2f}  p90 {np.percentile(col, 90):6.2f}  max {col.max():6.2f}")
