#!/usr/bin/env python3
"""Secondary BASELINE.json workloads on one MI355X (not the driver's bench line; numbers go to DESIGN.md):
  uni_2stage  configs[2]: stage-1 prompt L1=128, 256 forced new text tokens (EOS suppressed), then the
              layout->image path (L=256, 576 image tokens, VQ decode), bs=32            
  mmu         configs[4]: B=64 images -> SigLIP-L + aligner, prefill 576+64 embeddings per row, 256 forced new
              text tokens; VQ encode of the same images timed beside it                
usage: bench_configs.py {uni_2stage|mmu} [--batch N] [--steps K]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import synth_prompts
from plangen_amd.config import PlanGenConfig
from plangen_amd.engine import Engine


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=["uni_2stage", "mmu"])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--new-tokens", type=int, default=256)
    a = ap.parse_args()
    cfg = PlanGenConfig.janus_pro_1b()
    g = torch.Generator().manual_seed(0)
    NT = a.new_tokens
    if a.workload == "uni_2stage":
        B = a.batch or 32
        L1, L2 = 128, 256
        e = Engine(cfg, dtype="bf16", max_rows=2 * B, max_prompt=L2, max_new=cfg.img_tokens, max_images=B, with_lm_head=True)
        e.init_synthetic(seed=0)
        ids1 = torch.randint(10, cfg.vocab - 2048, (B, L1), generator=g).int()
        ids2, mask2 = synth_prompts(B, L2, cfg.vocab, cfg.pad_id, seed=0)
        pad2 = Engine.pad_len_from_mask(torch.cat([mask2, torch.ones((2 * B, cfg.img_tokens), dtype=torch.int32)], 1), L2)
        res = []
        for s in range(a.steps + 1):
            _, t_p1 = timed(lambda: e.prefill(ids1, [0] * B, position_mode=1))
            txt, t_txt = timed(lambda: e.generate_text_greedy(NT, cfg.eos_id, min_new_tokens=NT))
            _, t_p2 = timed(lambda: e.prefill(ids2, pad2, position_mode=0))
            toks, t_img = timed(lambda: e.decode_image_tokens(T=cfg.img_tokens, cfg_weight=5.0, temperature=1.0, seed=s))
            _, t_vq = timed(lambda: e.vq_decode(toks))
            res.append(dict(prefill1=t_p1, text_decode=t_txt, prefill2=t_p2, image_decode=t_img, vq_decode=t_vq))
        r = res[-1]; tot = sum(r.values())
        print(json.dumps({"workload": "uni_2stage bs=%d: L1=128 + %d forced text tokens, then L=256 + 576 image tokens + VQ decode" % (B, NT),
                          "ms": r, "total_ms": tot, "images_per_s": B / tot * 1e3, "text_tokens_per_s": B * NT / r["text_decode"] * 1e3,
                          "text_ms_per_step": r["text_decode"] / NT}))
    else:
        B = a.batch or 64
        P, Lt = cfg.vit_tokens, 64
        L = P + Lt
        e = Engine(cfg, dtype="bf16", max_rows=B, max_prompt=L, max_new=NT, max_images=B, with_lm_head=True, with_vq_encoder=True,
                   with_vision=True, max_vision_images=B)
        e.init_synthetic(seed=0)
        pix = (torch.rand(B, 3, cfg.vit_img, cfg.vit_img, generator=g) * 2 - 1).to(e.device)
        txt = e.embed_tokens(torch.randint(10, cfg.vocab - 2048, (B, Lt), generator=g).int())
        res = []
        for s in range(a.steps + 1):
            feats, t_vit = timed(lambda: e.vision_encode(pix, dtype=torch.bfloat16))
            emb = torch.cat([txt[:, :1].to(feats.dtype), feats, txt[:, 1:].to(feats.dtype)], 1).contiguous()
            _, t_pre = timed(lambda: e.prefill_embeds(emb, [0] * B, position_mode=1))
            out, t_txt = timed(lambda: e.generate_text_greedy(NT, cfg.eos_id, min_new_tokens=NT))
            _, t_enc = timed(lambda: e.vq_encode(pix))
            res.append(dict(vision_encode=t_vit, prefill=t_pre, text_decode=t_txt, vq_encode_beside=t_enc))
        r = res[-1]; tot = r["vision_encode"] + r["prefill"] + r["text_decode"]
        print(json.dumps({"workload": "mmu bs=%d: SigLIP-L/16-384 + aligner, prefill %d embeddings/row, %d forced text tokens" % (B, L, NT),
                          "ms": r, "total_ms": tot, "samples_per_s": B / tot * 1e3, "text_tokens_per_s": B * NT / r["text_decode"] * 1e3,
                          "text_ms_per_step": r["text_decode"] / NT,
                          "vision_tflops": 2 * 303e6 * P * B / (r["vision_encode"] * 1e-3) / 1e12,
                          "vq_encode_tflops": 310e9 * B / (r["vq_encode_beside"] * 1e-3) / 1e12}))


if __name__ == "__main__":
    main()
