#!/usr/bin/env python3
"""Idle-GPU stress of the prefill path (MFMA flash attention v2 with its LDS-DMA K/V ring, 256x256 GEMMs): the full-width 128-row
fixture batch is prefilled once per iteration with a pause in between; every result must equal the first bit for bit.
usage: prefill_cold_stress.py iters [pause_s]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_fullwidth as T
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
pause = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
e = T._engine("bf16")
ids, pad = T._inputs()
first = e.prefill(ids, pad, position_mode=0, return_hidden=True).clone()
bad = 0
for it in range(iters):
    time.sleep(pause)
    h = e.prefill(ids, pad, position_mode=0, return_hidden=True)
    if not torch.equal(h, first):
        bad += 1
        d = (h.float() - first.float()).abs()
        idx = (d > 0).nonzero()
        print(f"iter {it}: {idx.shape[0]} elements differ, rows {sorted(set(idx[:,0].tolist()))[:10]}, positions {sorted(set(idx[:,1].tolist()))[:10]}, max {d.max().item():.4g}", flush=True)
print(f"prefill: {bad} of {iters} differ from the first result")
