#!/usr/bin/env python3
"""Same sampled tokens with and without a set of engine options (bench workload, shortened), plus loop time.
usage: opt_equal.py B T key=value [diag:key=value ...]     (diag: = pg_diag_set_option; the engine then lives in libplangen_diag.so)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_prompts
from plangen_amd.config import PlanGenConfig
from plangen_amd.engine import Engine

B, T = int(sys.argv[1]), int(sys.argv[2])
opts = dict((k, int(v)) for k, v in (kv.split("=") for kv in sys.argv[3:]))
cfg = PlanGenConfig.janus_pro_1b()
L = 256
e = Engine(cfg, dtype="bf16", max_rows=2 * B, max_prompt=L, max_new=cfg.img_tokens, max_images=B, diag=any(k.startswith("diag:") for k in opts))
e.init_synthetic(seed=0)
ids, mask = synth_prompts(B, L, cfg.vocab, cfg.pad_id, seed=0)
pad = Engine.pad_len_from_mask(mask, L)
out = {}
for name in ("base", "opts", "base2", "opts2"):
    for k, v in opts.items():
        val = v if name.startswith("opts") else 0
        e.set_diag_option(k[5:], val) if k.startswith("diag:") else e.set_option(k, val)
    e.prefill(ids, pad)
    torch.cuda.synchronize(); t0 = time.time()
    toks = e.decode_image_tokens(T=T, cfg_weight=5.0, temperature=1.0, seed=3)
    torch.cuda.synchronize(); dt = time.time() - t0
    out[name] = toks.cpu()
    print(f"{name:6s} loop {dt * 1e3:8.1f} ms ({dt * 1e6 / (T - 1):7.1f} us/step)", flush=True)
same = all(torch.equal(out["base"], out[n]) for n in ("opts", "base2", "opts2"))
print("tokens identical" if same else "TOKENS DIFFER")
sys.exit(0 if same else 1)
