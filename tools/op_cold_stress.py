#!/usr/bin/env python3
"""Cold-launch stress of the other LDS-DMA kernels through the C ABI (fresh operands every iteration, one launch, compare with a
torch reference): the 128x128 GEMM, the 256x256 GEMM, the halo-tile 3x3 convolution.
usage: op_cold_stress.py iters"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from plangen_amd.config import PlanGenConfig
from plangen_amd.engine import Engine
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
e = Engine(PlanGenConfig.tiny(), dtype="bf16", max_rows=8, max_prompt=32, max_new=8, max_images=2)
e.init_synthetic(seed=0)
g = torch.Generator().manual_seed(3)
junk = []
def perturb(it):
    if it % 3 == 0: junk.append(torch.empty((1 + it % 7) * 1_000_003, device="cuda"))
    if len(junk) > 4: junk.pop(0)
for name, (M, N, K) in {"gemm 128x128 kernel": (640, 384, 1024), "gemm 256x256 kernel": (4096, 2048, 1024)}.items():
    bad = 0
    for it in range(iters):
        a = torch.randn(M, K, generator=g).bfloat16().float(); w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().float()
        perturb(it)
        out = e.op_gemm(a, w, 2).cpu()                      # kind 2: never the decode kernels
        ref = (a.cuda().double() @ w.cuda().double().t()).cpu()
        d = (out.double() - ref).abs(); tol = 2e-4 * ref.abs().max().item() + 1e-4
        if d.max().item() >= tol:
            bad += 1; idx = (d >= tol).nonzero()
            print(f"{name} iter {it}: max err {d.max().item():.4g}, {idx.shape[0]} elements, rows {sorted(set(idx[:,0].tolist()))[:8]} cols {sorted(set(idx[:,1].tolist()))[:8]}", flush=True)
    print(f"{name} M={M} N={N} K={K}: {bad} bad of {iters}", flush=True)
bad = 0
for it in range(iters):
    x = torch.randn(1, 128, 256, 128, generator=g).bfloat16().float()
    w = (torch.randn(128, 128, 3, 3, generator=g) * 0.03).bfloat16().float(); b = torch.randn(128, generator=g) * 0.1
    perturb(it)
    out = e.op_conv3x3(x, w, b).float().cpu()               # NHWC bf16 out
    ref = F.conv2d(x.permute(0, 3, 1, 2).cuda(), w.cuda(), b.cuda(), padding=1).permute(0, 2, 3, 1).cpu()
    d = (out - ref).abs(); tol = 0.02 * ref.abs().max().item()
    if d.max().item() >= tol:
        bad += 1; idx = (d >= tol).nonzero()
        print(f"conv iter {it}: max err {d.max().item():.4g} (tol {tol:.3g}), {idx.shape[0]} elements, y {sorted(set(idx[:,1].tolist()))[:8]} x {sorted(set(idx[:,2].tolist()))[:8]}", flush=True)
print(f"halo conv 128x256x128: {bad} bad of {iters}")
