#!/usr/bin/env python3
"""Benchmark of the layout->image hot path on MI355X (BASELINE.json metric).

One "step" = one full pass of the path over one batch of synthetic prompts per GPU:
prefill of 2B CFG rows (left-padded to L) -> 576-step CFG decode loop (multinomial sampling at
temperature 1 via Gumbel-max, same cost as greedy) -> VQ-16 decode to B 384x384 images.
Inputs (token ids) are resident in HBM before the timed region.  Multi-GPU: prompts are
sharded over ranks (weak scaling: B images per GPU), rank 0 broadcasts the collated ids over
RCCL, no collective inside the loop, tokens all-gathered at the end.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def synth_prompts(B, L, vocab, pad_id, seed):
    """cond prompts with true lengths U{160..256} (scaled to L) left-padded to L;
    one fixed 96-token (scaled) uncond prompt shared by all rows; ids uniform, first real
    token BOS(=1)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    lo, unc = max(1, (L * 160) // 256), max(1, (L * 96) // 256)
    ids = torch.full((2 * B, L), pad_id, dtype=torch.int32)
    mask = torch.zeros((2 * B, L), dtype=torch.int32)
    unc_ids = torch.randint(10, vocab - 2048, (unc,), generator=g).int()
    unc_ids[0] = 1
    for b in range(B):
        n = int(torch.randint(lo, L + 1, (1,), generator=g))
        row = torch.randint(10, vocab - 2048, (n,), generator=g).int()
        row[0] = 1
        ids[2 * b, L - n:] = row
        mask[2 * b, L - n:] = 1
        ids[2 * b + 1, L - unc:] = unc_ids
        mask[2 * b + 1, L - unc:] = 1
    return ids, mask


def cpu_baseline(L, steps_sample, threads):
    """The CPU oracle (torch-CPU fp32 restatement of the reference, kind 'port') on a bounded
    sample of the same workload: 1 image (2 CFG rows), prefill at L, ``steps_sample`` decode
    steps, 1 VQ decode; extrapolated to 576 steps."""
    import torch
    from oracle import ref_cpu as R
    torch.set_num_threads(threads)
    cfg = R.OracleCfg(vocab=4096)              # embedding rows do not affect the timed arithmetic
    W = R.make_weights(cfg, seed=0, with_lm_head=False)
    ids, mask = synth_prompts(1, L, cfg.vocab, 3, 0)
    mask = torch.cat([mask, torch.ones((2, cfg.img_tokens), dtype=torch.int32)], dim=1)
    with torch.no_grad():
        emb = R.embed_tokens(W, ids)
        t0 = time.time()
        pos = torch.arange(L)[None].expand(2, -1)
        hid, cache = R.llama_forward(W, cfg, emb, mask, pos)
        t_prefill = time.time() - t0
        x = R.prepare_gen_img_embeds(W, torch.zeros(2, dtype=torch.long))[:, None]
        t0 = time.time()
        for i in range(steps_sample):
            p = torch.full((2, 1), L + i)
            hid, cache = R.llama_forward(W, cfg, x, mask, p, cache)
            logits = R.gen_head(W, hid[:, -1])
            mixed = logits[1::2] + 5.0 * (logits[0::2] - logits[1::2])
            tok = torch.argmax(mixed, -1)
            x = R.prepare_gen_img_embeds(W, torch.stack([tok, tok], 1).view(-1))[:, None]
        t_step = (time.time() - t0) / steps_sample
        codes = torch.randint(0, cfg.img_vocab, (1, cfg.img_tokens))
        t0 = time.time()
        R.vq_decode_code(W, cfg, codes)
        t_vq = time.time() - t0
    per_image = t_prefill + 576 * t_step + t_vq
    return {"value": 1.0 / per_image, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"1 image (2 CFG rows): prefill L={L} {t_prefill:.2f}s, {steps_sample} decode steps "
                      f"{t_step * 1e3:.0f} ms/step extrapolated to 576, VQ decode {t_vq:.2f}s",
            "image_tokens_per_s": 576.0 / (t_prefill + 576 * t_step)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU (BASELINE: bs=64)")
    ap.add_argument("--prompt-len", type=int, default=256)
    ap.add_argument("--tokens", type=int, default=None, help="image tokens per image (default 576)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--tiny", action="store_true", help="tiny config (plumbing check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=64)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--opt", action="append", default=[], help="engine tuning option key=value (pg_set_option)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from plangen_amd.config import PlanGenConfig
    from plangen_amd.dist import broadcast_prompts, gather_rows
    from plangen_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("PG_FORCE_DEVICE") is not None:      # debugging aid: several ranks on one GPU (gloo only)
        local = int(os.environ["PG_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("PG_DIST_BACKEND", "nccl")    # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cfg = PlanGenConfig.tiny() if args.tiny else PlanGenConfig.janus_pro_1b()
    B, L = args.batch, args.prompt_len
    T = args.tokens or cfg.img_tokens
    eng = Engine(cfg, dtype=args.dtype, max_rows=2 * B, max_prompt=L, max_new=cfg.img_tokens, max_images=B, device=local)
    eng.init_synthetic(seed=0)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))

    # rank 0 collates the global batch (B images per rank) and broadcasts it over RCCL
    if rank == 0:
        g_ids, g_mask = synth_prompts(B * world, L, cfg.vocab, cfg.pad_id, seed=0)
        g_mask = torch.cat([g_mask, torch.ones((2 * B * world, cfg.img_tokens), dtype=torch.int32)], dim=1)
    else:
        g_ids = g_mask = None
    ids, mask, lo, hi, nB = broadcast_prompts(g_ids, g_mask, dev)
    pad = Engine.pad_len_from_mask(mask, L)

    def step(seed):
        eng.prefill(ids, pad, position_mode=0)
        toks = eng.decode_image_tokens(T=T, cfg_weight=cfg.cfg_weight, temperature=args.temperature, seed=seed)
        img = eng.vq_decode(toks) if T == cfg.img_tokens else None
        all_toks = gather_rows(toks, nB)
        return all_toks, img

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for w in range(args.warmup):
        step(1000 + w)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tm = eng.timing()
    images = B * world * args.steps
    out = {
        "metric": "images/sec, 384px layout2image (576 image tokens, CFG, VQ-16 decode), bs=%d per MI355X" % B,
        "value": images / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic prompts (SURVEY 8d), seeded random-init Janus-Pro-1B-shaped weights",
        "config": {"workload": "task_type='uni' layout2image, %dx%d / %d image tokens, bs=%d per GPU, L=%d left-padded, "
                               "cfg_weight=%g, temperature=%g" % (cfg.img_size, cfg.img_size, T, B, L, cfg.cfg_weight, args.temperature),
                   "images_per_gpu": B, "prompt_len": L, "rows_per_gpu": 2 * B, "parallelism": "prompt-sharded x%d" % world},
        "image_tokens_per_sec_per_gpu": B * T * args.steps / dt,
        "last_step_ms": {"prefill": tm["prefill_ms"], "decode_loop": tm["decode_ms"], "vq_decode": tm["vq_ms"]},
        "device_gb": eng.device_bytes() / 2 ** 30,
    }

    if not args.no_roofline:
        # Dominant kernel = decode attention (HBM-bound KV streaming).  Instrumented pass right
        # after the timed region: same batch, eager launches, hipEvents on the launch stream
        # around every decode-attention launch (24 layers x 575 steps).
        eng.set_option("time_attn", 1)
        eng.prefill(ids, pad, position_mode=0)
        eng.decode_image_tokens(T=T, cfg_weight=cfg.cfg_weight, temperature=args.temperature, seed=99)
        torch.cuda.synchronize()
        ta = eng.timing()
        eng.set_option("time_attn", 0)
        if ta["attn_launches"]:
            ach = ta["attn_bytes_sum"] / (ta["attn_ms_sum"] * 1e-3) / 1e9
            # PMC traffic cannot be collected from inside this process: it comes from separate
            # rocprofv3 --pmc passes of this same command (profiles/r01_b_pmc_attn_traffic.md);
            # the measured traffic / algorithmic ratio of the kernel is applied to this run's bytes.
            traffic = None
            pj = os.path.join(ROOT, "profiles", "pmc_attn.json")
            if os.path.exists(pj):
                ratio = json.load(open(pj))["traffic_per_algorithmic_byte"]
                traffic = ratio * ta["attn_bytes_sum"] / ta["attn_launches"]
            out["roofline"] = {"bound": "hbm", "kernel": "attn_decode_fused_kernel (RoPE + KV append + decode attention)",
                               "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic,
                               "launches": ta["attn_launches"], "avg_launch_us": ta["attn_ms_sum"] / ta["attn_launches"] * 1e3,
                               "algorithmic_bytes_per_launch": ta["attn_bytes_sum"] / ta["attn_launches"],
                               "decode_loop_share": ta["attn_ms_sum"] / max(ta["decode_ms"], 1e-9)}
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.tiny:
        # 16 threads: measured fastest for these small torch-CPU GEMMs on the 256-core bench host
        # (ms/step: 16 thr 89, 32 thr 144, 64 thr 298, 256 thr 42 466)
        out["cpu_baseline"] = cpu_baseline(L, args.cpu_steps, min(args.cpu_threads, os.cpu_count() or 1))
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
