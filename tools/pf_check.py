#!/usr/bin/env python3
"""Run-ahead weight stream (mall_prefetch): same tokens with and without it, prefetcher statistics, loop time.
usage: python tools/pf_check.py [batch] [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_prompts
from plangen_amd.config import PlanGenConfig
from plangen_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 96
cfg = PlanGenConfig.janus_pro_1b()
L = 256
e = Engine(cfg, dtype="bf16", max_rows=2 * B, max_prompt=L, max_new=cfg.img_tokens, max_images=B)
e.init_synthetic(seed=0)
ids, mask = synth_prompts(B, L, cfg.vocab, cfg.pad_id, seed=0)
pad = Engine.pad_len_from_mask(mask, L)
out = {}
for name, opts in (("off", {"mall_prefetch": 0}), ("on", {"mall_prefetch": 1}), ("off2", {"mall_prefetch": 0}), ("on2", {"mall_prefetch": 1}), ("off3", {"mall_prefetch": 0})):
    for k, v in opts.items():
        e.set_option(k, v)
    e.prefill(ids, pad)
    torch.cuda.synchronize(); t0 = time.time()
    toks = e.decode_image_tokens(T=T, cfg_weight=5.0, temperature=1.0, seed=3)
    torch.cuda.synchronize(); dt = time.time() - t0
    st = e.debug_read("pf_stats", 0, 4, torch.int32).cpu().tolist()
    out[name] = toks.cpu()
    print(f"{name:5s} loop {dt * 1e3:8.1f} ms  ({dt * 1e6 / (T - 1):7.1f} us/step)  pf_stats done/skipped/timeout = {st[:3]}", flush=True)
names = list(out)
for i in range(len(names)):
    for j in range(i + 1, len(names)):
        a, b = out[names[i]], out[names[j]]
        d = (a != b)
        if d.any():
            first = int(d.any(0).float().argmax())
            print(f"{names[i]} vs {names[j]}: {int(d.sum())} tokens differ, images {d.any(1).nonzero().flatten().tolist()[:8]}, first step {first}")
        else:
            print(f"{names[i]} vs {names[j]}: identical")
