#!/usr/bin/env python3
"""Host enqueue time of the eager decode loop vs its GPU time (is the launch loop host-bound?).  usage: host_margin.py [batch ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_prompts  # noqa: E402
from plangen_amd.config import PlanGenConfig  # noqa: E402
from plangen_amd.engine import Engine  # noqa: E402

cfg = PlanGenConfig.janus_pro_1b()
for B in [int(a) for a in sys.argv[1:]] or [8, 64]:
    eng = Engine(cfg, dtype="bf16", max_rows=2 * B, max_prompt=256, max_new=cfg.img_tokens, max_images=B, device=0)
    eng.init_synthetic(seed=0)
    ids, mask = synth_prompts(B, 256, cfg.vocab, cfg.pad_id, seed=0)
    ids = ids.cuda()
    pad = Engine.pad_len_from_mask(torch.cat([mask, torch.ones((2 * B, cfg.img_tokens), dtype=torch.int32)], dim=1), 256)
    for it in range(3):
        eng.prefill(ids, pad, position_mode=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.decode_image_tokens(T=cfg.img_tokens, cfg_weight=cfg.cfg_weight, temperature=1.0, seed=it)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"bs={B}: graph={os.environ.get('PG_USE_GRAPH', '0') == '1'}  host enqueue {1e3 * (t1 - t0):.1f} ms, until done {1e3 * (t2 - t0):.1f} ms", flush=True)
    eng.close() if hasattr(eng, "close") else None
    del eng
