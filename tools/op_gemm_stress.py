#!/usr/bin/env python3
"""Cold-launch stress of the tiled decode GEMM through the C ABI: fresh operands every iteration, one launch, compare with fp64.
usage: op_gemm_stress.py iters [M N K] [stream_gemm]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plangen_amd.config import PlanGenConfig
from plangen_amd.engine import Engine
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
M, N, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (64, 6144, 2048)
cfg = PlanGenConfig.tiny()
e = Engine(cfg, dtype="bf16", max_rows=8, max_prompt=32, max_new=8, max_images=2)
e.init_synthetic(seed=0)
if len(sys.argv) > 5: e.set_option("stream_gemm", int(sys.argv[5]))
g = torch.Generator().manual_seed(1)
bad = 0
junk = []
for it in range(iters):
    a = (torch.randn(M, K, generator=g)).bfloat16().float()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().float()
    if it % 3 == 0: junk.append(torch.empty((1 + it % 7) * 1_000_003, device="cuda"))      # perturb the allocator
    if len(junk) > 4: junk.pop(0)
    out = e.op_gemm(a, w, 4).cpu()
    ref = a.double() @ w.double().t()
    d = (out.double() - ref).abs()
    tol = 2e-4 * ref.abs().max().item() + 1e-4
    if d.max().item() >= tol:
        bad += 1
        idx = (d >= tol).nonzero()
        print(f"iter {it}: max err {d.max().item():.4g} (tol {tol:.3g}), {idx.shape[0]} elements, rows {sorted(set(idx[:,0].tolist()))[:12]}, n-tiles {sorted(set((idx[:,1]//16).tolist()))[:12]}", flush=True)
print(f"M={M} N={N} K={K}: {bad} bad of {iters}")
