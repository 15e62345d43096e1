#!/usr/bin/env python3
"""TEMP: per-tile stamps of the halo kernels (block 7, first 64 tiles; waves 0 (leading) and 4 (trailing))."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from plangen_amd.config import PlanGenConfig
from plangen_amd.engine import Engine
cfg = PlanGenConfig.janus_pro_1b()
e = Engine(cfg, dtype="bf16", max_rows=2, max_prompt=16, max_new=8, max_images=2, with_vq_encoder=False, diag=True)
e.init_synthetic(seed=0)
B = 64
g = torch.Generator().manual_seed(1)
xn = torch.randn(B, 384, 384, 128, generator=g).bfloat16()
w = (torch.randn(128, 128, 3, 3, generator=g) / 34).bfloat16().float()
bias = torch.randn(128, generator=g)
rn = torch.randn(B, 384, 384, 128, generator=g).bfloat16()
e.lib.pg_bench_halo_prof.argtypes = [C.c_void_p]
e.lib.pg_bench_halo_prof2.argtypes = [C.c_void_p]
for res in (False,):
    for opt in (11, 12):
        e.set_option("conv_halo", opt)
        for _ in range(2): e.op_conv3x3(xn, w, bias, rn if res else None, 0, 0)
        buf = np.zeros(2 * 64 * 8, dtype=np.uint64)
        assert e.lib.pg_bench_halo_prof(buf.ctypes.data) == 0
        st = buf.reshape(2, 64, 8).astype(np.int64)
        for grp in (0, 1):
            s = st[grp, 8:56, :5] * 0.01          # us (100 MHz)
            d = np.diff(s, axis=1)                 # top->loopstart, loop, loopend->realign, epilogue
            per_tile = np.diff(st[grp, 8:56, 0] * 0.01)
            print(f"res={res} conv_halo={opt} group {grp}: tile period {per_tile.mean():.2f} us | top wait {d[:,0].mean():.2f}  loop {d[:,1].mean():.2f}  realign {d[:,2].mean():.2f}  epilogue(+fill issue) {d[:,3].mean():.2f}")
        b2 = np.zeros(2 * 18 * 8, dtype=np.uint64)
        assert e.lib.pg_bench_halo_prof2(b2.ctypes.data) == 0
        p2 = b2.reshape(2, 18, 8).astype(np.int64) * 0.01
        for grp in (0, 1):
            d = np.diff(p2[grp, :, :7], axis=1)[1:17]            # issue reads | stage W | vmcnt wait | barrier X | lgkm wait | MFMA issue | barrier Y
            nxt = (p2[grp, 1:, 0] - p2[grp, :-1, 6])[1:16]
            print(f"   phases, group {grp}: reads-issue {d[:,0].mean():.3f}  stage+vmcnt {d[:,1].mean():.3f}  barX {d[:,2].mean():.3f}  lgkm {d[:,3].mean():.3f}  mfma-issue {d[:,4].mean():.3f}  barY {d[:,5].mean():.3f}  loop-back {nxt.mean():.3f} | phase {(p2[grp,17,0]-p2[grp,1,0])/16:.3f} us")
