#!/usr/bin/env python3
"""conv_halo = 3 (channel-split ping-pong halo, round 6) against conv_halo = 1 (one 85 KiB patch): results and whole-decode time.
usage: halo2_check.py [B=64]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from plangen_amd.config import PlanGenConfig
from plangen_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = PlanGenConfig.janus_pro_1b()
e = Engine(cfg, dtype="bf16", max_rows=2, max_prompt=16, max_new=8, max_images=B, with_vq_encoder=False)
e.init_synthetic(seed=0)
bad = 0
for res, b, hs, ws, up in [(False, 4, 64, 128, 0), (True, 3, 96, 160, 0), (True, 1, 192, 192, 0), (False, 4, 32, 64, 1), (False, 2, 96, 96, 1), (True, 2, 384, 384, 0)]:
    g = torch.Generator().manual_seed(7 + hs)
    x = torch.randn(b, 128, hs, ws, generator=g).bfloat16().float()
    w = (torch.randn(128, 128, 3, 3, generator=g) / 34).bfloat16().float()
    bias = torch.randn(128, generator=g)
    xn = x.permute(0, 2, 3, 1).contiguous()
    ho, wo = hs << up, ws << up
    rn = torch.randn(b, ho, wo, 128, generator=g).bfloat16().float() if res else None
    outs = {}
    for half in ("full", "lo", "hi"):
        xi = xn.clone()
        if half == "lo": xi[..., 64:] = 0
        if half == "hi": xi[..., :64] = 0
        for opt in (1, 3, 4, 5):
            e.set_option("conv_halo", opt)
            outs[half, opt] = e.op_conv3x3(xi, w, bias, rn, up, 0).float().cpu()
            again = e.op_conv3x3(xi, w, bias, rn, up, 0).float().cpu()
            if not torch.equal(again, outs[half, opt]): bad += 1; print("NOT REPRODUCIBLE", half, opt)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x, w, bias, padding=1).permute(0, 2, 3, 1)
    if res: ref = ref + rn
    d_full = (outs["full", 3] - outs["full", 1]).abs().max().item()
    eq_lo = torch.equal(outs["lo", 3], outs["lo", 1]); eq_hi = torch.equal(outs["hi", 3], outs["hi", 1])
    same = all(torch.equal(outs[h, 1], outs[h, o]) for h in ("full", "lo", "hi") for o in (4, 5))
    if not same: bad += 1
    print(f"   fast epilogue (1) == prefetch-3 (4) == generic epilogue (5): {same}")
    e3 = (outs["full", 3] - ref).abs().max().item(); e1 = (outs["full", 1] - ref).abs().max().item()
    nd = (outs["full", 3] != outs["full", 1]).float().mean().item()
    print(f"res={res} B={b} {hs}x{ws} up={up}: halves bit-identical lo={eq_lo} hi={eq_hi}; full: max|new-old|={d_full:.4g} ({100*nd:.3f}% of elements differ), "
          f"max err vs fp32 conv new={e3:.4g} old={e1:.4g} (|ref| max {ref.abs().max().item():.3g})")
    if not (eq_lo and eq_hi) or e3 > 2e-2 * ref.abs().max().item(): bad += 1
print("RESULT", "ok" if bad == 0 else f"{bad} FAILED")

codes = torch.randint(0, cfg.img_vocab, (B, cfg.img_tokens)).int()
imgs = {}
for rep in range(2):
    for opt in (5, 1, 4, 3):
        e.set_option("conv_halo", opt)
        ts = []
        for i in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            img = e.vq_decode(codes)
            torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        imgs[opt] = img.float().cpu()
        print(f"conv_halo={opt}: vq_decode of {B} images {min(ts[1:]):.2f} ms (runs {', '.join(f'{t:.1f}' for t in ts)})")
print('decode pixels 5 == 1 == 4:', torch.equal(imgs[5], imgs[1]) and torch.equal(imgs[4], imgs[1]))
d = (imgs[3] - imgs[1])
print(f"decode pixels new vs old: max |d| {d.abs().max().item():.4g}, mse {d.pow(2).mean().item():.3g}")
