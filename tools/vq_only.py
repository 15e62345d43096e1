#!/usr/bin/env python3
"""VQ-16 decode (and encode) of B images on one GPU, for rocprofv3 kernel traces. usage: vq_only.py [B=64] [iters=2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plangen_amd.config import PlanGenConfig
from plangen_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
it = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = PlanGenConfig.janus_pro_1b()
e = Engine(cfg, dtype="bf16", max_rows=2, max_prompt=16, max_new=8, max_images=B, with_vq_encoder=True)
e.init_synthetic(seed=0)
for kv in os.environ.get("PG_OPTS", "").split(","):          # e.g. PG_OPTS=conv_halo=3
    if kv: e.set_option(kv.split("=")[0], int(kv.split("=")[1]))
codes = torch.randint(0, cfg.img_vocab, (B, cfg.img_tokens)).int()
for i in range(it):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img = e.vq_decode(codes)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    idx = e.vq_encode(img.clamp(-1, 1))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"decode {1e3 * (t1 - t0):.1f} ms  encode {1e3 * (t2 - t1):.1f} ms")
