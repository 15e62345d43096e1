#!/usr/bin/env python3
"""Idle-GPU stress of the whole generation path at full size: prefill + a few greedy decode steps per iteration with a pause in between;
every iteration's tokens must equal the first iteration's.  usage: decode_cold_stress.py iters [batch] [tokens] [pause_s]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import synth_prompts
from plangen_amd.config import PlanGenConfig
from plangen_amd.engine import Engine
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T = int(sys.argv[3]) if len(sys.argv) > 3 else 16
pause = float(sys.argv[4]) if len(sys.argv) > 4 else 0.2
cfg = PlanGenConfig.janus_pro_1b()
L = 256
e = Engine(cfg, dtype="bf16", max_rows=2 * B, max_prompt=L, max_new=cfg.img_tokens, max_images=B)
e.init_synthetic(seed=0)
ids, mask = synth_prompts(B, L, cfg.vocab, cfg.pad_id, seed=0)
mask = torch.cat([mask, torch.ones((2 * B, cfg.img_tokens), dtype=torch.int32)], 1)
pad = Engine.pad_len_from_mask(mask, L)
first = None; bad = 0
for it in range(iters):
    time.sleep(pause)
    e.prefill(ids, pad, position_mode=0)
    toks = e.decode_image_tokens(T=T, cfg_weight=cfg.cfg_weight, temperature=0.0).cpu()
    if first is None: first = toks.clone()
    elif not torch.equal(toks, first):
        bad += 1
        print(f"iter {it}: {(toks != first).sum().item()} tokens differ, first at step {int((toks != first).any(0).nonzero()[0])}", flush=True)
print(f"bs={B}, {T} greedy steps: {bad} of {iters - 1} iterations differ from the first")
