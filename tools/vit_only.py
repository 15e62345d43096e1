#!/usr/bin/env python3
"""SigLIP-L/16-384 + aligner on B images (rocprofv3 kernel traces). usage: vit_only.py [B=64] [iters=2] [key=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plangen_amd.config import PlanGenConfig
from plangen_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
it = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = PlanGenConfig.janus_pro_1b()
e = Engine(cfg, dtype="bf16", max_rows=2, max_prompt=16, max_new=8, max_images=1, with_vision=True, max_vision_images=B)
e.init_synthetic(seed=0)
for kv in sys.argv[3:]:
    k, v = kv.split("="); e.set_option(k, int(v))
pix = (torch.rand(B, 3, cfg.vit_img, cfg.vit_img) * 2 - 1).to(e.device)
for i in range(it):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f = e.vision_encode(pix, dtype=torch.bfloat16)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"vision_encode {1e3 * (t1 - t0):.1f} ms  ({2 * 303e6 * cfg.vit_tokens * B / (t1 - t0) / 1e12:.0f} TFLOP/s)")
