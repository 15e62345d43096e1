#!/usr/bin/env python3
"""Benchmark of the layout->image hot path on MI355X (BASELINE.json metric).

One "step" = one full pass of the path over one batch of synthetic prompts per GPU:
prefill of 2B CFG rows (left-padded to L) -> 576-step CFG decode loop (multinomial sampling at
temperature 1 via Gumbel-max, same cost as greedy) -> VQ-16 decode to B 384x384 images.
Inputs (token ids) are resident in HBM before the timed region.

Multi-GPU: prompts are sharded over ranks, one process per GPU; rank 0 broadcasts the
collated ids over RCCL, no collective inside the loop, tokens gathered to rank 0 at the end.
  python bench.py --gpus N                      spawns N ranks itself (fresh child processes; the parent
                                                never touches a GPU) and relays rank 0's JSON line
  python -m torch.distributed.run ... bench.py --gpus N    runs as the rank the launcher made it
  default: weak scaling, --batch images per GPU;  --global-batch G (BASELINE configs[3]: 256 over 8 GPUs):
  strong scaling, G / N images per GPU.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")     # the ROCm 7.2 default, pinned: stream launches need it (plangen_amd/__init__.py)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable float4 copy)
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA
WEIGHT_PARAMS_LAYER = 51_380_224          # SURVEY App. A [probe]
VQ_DECODE_GFLOP_PER_IMAGE = 567.40 + 4 * 0.68   # SURVEY 8d / App. C conv + AttnBlock FLOPs


def synth_prompts(B, L, vocab, pad_id, seed):
    """cond prompts with true lengths U{160..256} (scaled to L) left-padded to L;
    one fixed 96-token (scaled) uncond prompt shared by all rows; ids uniform, first real
    token BOS(=1)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    lo, unc = max(1, (L * 160) // 256), max(1, (L * 96) // 256)
    hi = vocab - 2048 if vocab > 4096 else vocab               # full vocabulary: stay below the special-token tail
    ids = torch.full((2 * B, L), pad_id, dtype=torch.int32)
    mask = torch.zeros((2 * B, L), dtype=torch.int32)
    unc_ids = torch.randint(10, hi, (unc,), generator=g).int()
    unc_ids[0] = 1
    for b in range(B):
        n = int(torch.randint(lo, L + 1, (1,), generator=g))
        row = torch.randint(10, hi, (n,), generator=g).int()
        row[0] = 1
        ids[2 * b, L - n:] = row
        mask[2 * b, L - n:] = 1
        ids[2 * b + 1, L - unc:] = unc_ids
        mask[2 * b + 1, L - unc:] = 1
    return ids, mask


def cpu_baseline(L, steps_sample, threads):
    """The CPU oracle (torch-CPU fp32 restatement of the reference, kind 'port') on a bounded sample of the same workload: 1 image
    (2 CFG rows), prefill at L, decode steps timed at FIVE contexts spread over the loop (L + {0, 144, 288, 432, 560}: the K/V the steps
    read grows from 257 to 831 keys, so the first steps alone would understate a step), 1 VQ decode; the per-step time is the mean over
    the five contexts, extrapolated to 576 steps.  ``threads`` is the measured-fastest thread count; an all-core figure is reported
    beside it from a 2-layer slice of one step (SURVEY 8d asks for "all physical cores", which is far slower for these skinny GEMMs)."""
    import torch
    from oracle import ref_cpu as R
    torch.set_num_threads(threads)
    cfg = R.OracleCfg(vocab=4096)              # embedding rows do not affect the timed arithmetic
    W = R.make_weights(cfg, seed=0, with_lm_head=False)
    ids, mask = synth_prompts(1, L, cfg.vocab, 3, 0)
    mask = torch.cat([mask, torch.ones((2, cfg.img_tokens), dtype=torch.int32)], dim=1)
    offsets = [0, 144, 288, 432, 560]
    per_ctx = max(2, steps_sample // len(offsets))
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        emb = R.embed_tokens(W, ids)
        t0 = time.time()
        pos = torch.arange(L)[None].expand(2, -1)
        hid, cache = R.llama_forward(W, cfg, emb, mask, pos)
        t_prefill = time.time() - t0
        x = R.prepare_gen_img_embeds(W, torch.zeros(2, dtype=torch.long))[:, None]
        step_ms = {}
        for off in offsets:
            # a cache of L + off keys: the prefilled prompt + random K/V of the right shape (step time does not depend on the values)
            grow = L + off - cache.length()
            if grow > 0:
                for i in range(cfg.n_layers):
                    pad_kv = torch.randn((2, cfg.n_heads, grow, cfg.head_dim), generator=g) * 0.5
                    cache.k[i] = torch.cat([cache.k[i], pad_kv], dim=2)
                    cache.v[i] = torch.cat([cache.v[i], pad_kv], dim=2)
            t0 = time.time()
            for i in range(per_ctx):
                p = torch.full((2, 1), L + off + i)
                hid, cache = R.llama_forward(W, cfg, x, mask, p, cache)
                logits = R.gen_head(W, hid[:, -1])
                mixed = logits[1::2] + 5.0 * (logits[0::2] - logits[1::2])
                tok = torch.argmax(mixed, -1)
                x = R.prepare_gen_img_embeds(W, torch.stack([tok, tok], 1).view(-1))[:, None]
            step_ms[off] = (time.time() - t0) / per_ctx * 1e3
        t_step = sum(step_ms.values()) / len(step_ms) * 1e-3
        codes = torch.randint(0, cfg.img_vocab, (1, cfg.img_tokens))
        t0 = time.time()
        R.vq_decode_code(W, cfg, codes)
        t_vq = time.time() - t0
        # all cores, bounded: ONE decode step through a 2-layer slice of the same weights (x 12 = the 24-layer stack; gen_head excluded)
        import dataclasses
        ncpu = os.cpu_count() or 1
        all_core = None
        if ncpu != threads:
            torch.set_num_threads(ncpu)
            cfg2 = dataclasses.replace(cfg, n_layers=2)
            c2 = R.KVCache([cache.k[0][:, :, :L + 288].clone(), cache.k[1][:, :, :L + 288].clone()],
                           [cache.v[0][:, :, :L + 288].clone(), cache.v[1][:, :, :L + 288].clone()])
            p = torch.full((2, 1), L + 288)
            R.llama_forward(W, cfg2, x, mask, p, c2)                    # warm-up (thread pool start)
            t0 = time.time()
            for i in range(2):
                R.llama_forward(W, cfg2, x, mask, torch.full((2, 1), L + 289 + i), c2)
            all_core = {"cores": ncpu, "ms_per_step_est": (time.time() - t0) / 2 * 12 * 1e3,
                        "what": "2 decode steps through layers 0-1 at context L+288, x12 (gen_head excluded)"}
            torch.set_num_threads(threads)
    per_image = t_prefill + 576 * t_step + t_vq
    out = {"value": 1.0 / per_image, "unit": "images/s", "cores": threads, "kind": "port",
           "sample": f"1 image (2 CFG rows): prefill L={L} {t_prefill:.2f}s, {per_ctx} decode steps at each of the contexts L+{offsets} "
                     f"(ms/step {[round(step_ms[o]) for o in offsets]}, mean {t_step * 1e3:.0f}) extrapolated to 576, VQ decode {t_vq:.2f}s",
           "image_tokens_per_s": 576.0 / (t_prefill + 576 * t_step), "ms_per_step_by_context": {str(L + o): step_ms[o] for o in offsets}}
    if all_core:
        out["all_cores"] = all_core
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU (BASELINE: bs=64); weak scaling")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="total images over all GPUs (BASELINE configs[3]: 256 over 8 GPUs); strong scaling")
    ap.add_argument("--prompt-len", type=int, default=256)
    ap.add_argument("--tokens", type=int, default=None, help="image tokens per image (default 576)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--tiny", action="store_true", help="tiny config (plumbing check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-shard-check", action="store_true")
    ap.add_argument("--pipeline", type=int, default=0, help="N > 0: after the timed region, N batches with pg_vq_decode of batch k on a second stream under the "
                    "prefill + decode loop of batch k+1 (VERDICT r4 item 6); reported as pipelined_images_per_s, never as the headline value")
    ap.add_argument("--no-gemm-phase", action="store_true", help="skip the second (libplangen_diag.so) handle that times the decode step without attention")
    ap.add_argument("--diag-opt", action="append", default=[], help="MEASUREMENT ONLY: key=value for pg_diag_set_option; the whole run then uses "
                    "libplangen_diag.so and the line says so (tools/ab_loop.sh)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` block (BASELINE configs[1], [2], [4] at full length after the timed region; ~20 s)")
    ap.add_argument("--cpu-steps", type=int, default=64)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--time-stride", type=int, default=8, help="instrumented pass: time every n-th decode step")
    ap.add_argument("--opt", action="append", default=[], help="engine tuning option key=value (pg_set_option)")
    ap.add_argument("--no-rccl-selftest", action="store_true", help="N=1: skip the one-rank RCCL self-test child run after the timed region")
    ap.add_argument("--launch-check", action="store_true",
                    help="launcher dry run (no GPU): every rank checks its env, joins a gloo group, all-reduces, rank 0 prints JSON")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ launcher
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv, poll_s=0.2, grace_s=5.0):
    """Start ``n`` fresh rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relay rank 0's stdout.
    The parent has not touched the GPU (no torch.cuda call, no HIP call) and never execs.  ALL children are polled: on the first
    non-zero exit the others are terminated (SIGTERM, SIGKILL after ``grace_s``) and the launcher returns non-zero at once --
    a rank that dies before a collective would otherwise leave the survivors in RCCL until the watchdog fires (minutes)."""
    import tempfile
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    out0 = tempfile.TemporaryFile()

    # The ranks live in their own sessions, so a signal to the launcher's process group does not reach them.  Python's default
    # SIGTERM / SIGHUP action would kill the launcher without unwinding and leave the ranks holding the GPUs: turn
    # both into an exception so the kill path below runs.  Second line of defence: every rank asks the kernel to SIGTERM it
    # when its parent dies (PR_SET_PDEATHSIG), which also covers a SIGKILLed launcher.
    class _Terminated(BaseException):
        pass

    def _on_signal(signum, frame):
        raise _Terminated(signum)
    old_handlers = {sg: signal.signal(sg, _on_signal) for sg in (signal.SIGTERM, signal.SIGHUP)}
    old_int = signal.getsignal(signal.SIGINT)

    def _pdeathsig():
        try:
            import ctypes
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)      # PR_SET_PDEATHSIG = 1
        except Exception:
            pass

    rcs = [None] * n
    failed = None
    interrupted = False
    try:
        # the spawn loop is INSIDE the try: a signal during rank start-up must reach the kill path with the ranks started so far
        for r in range(n):
            env = dict(os.environ)
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=port, PG_BENCH_CHILD="1")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=out0 if r == 0 else None, start_new_session=True, preexec_fn=_pdeathsig))
        while any(rc is None for rc in rcs):
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    rcs[r] = p.poll()
                    if rcs[r] not in (None, 0) and failed is None:
                        failed = r
            if failed is not None:
                break
            time.sleep(poll_s)
    except BaseException:                                      # Ctrl-C / SIGTERM of the launcher: the ranks live in their own sessions and would survive it
        interrupted = True
        failed = next((r for r in range(len(procs)) if rcs[r] is None), 0)
    try:
        if failed is not None:
            # a second signal must not abort the kill path (the SIGKILL escalation would never run): ignored until the ranks are gone
            for sg in (signal.SIGTERM, signal.SIGHUP, signal.SIGINT):
                signal.signal(sg, signal.SIG_IGN)
            sys.stderr.write(("bench.py: launcher interrupted" if interrupted else f"bench.py: rank {failed} exited with rc {rcs[failed]}") + "; terminating the other ranks\n")
            live = [p for p in procs if p.poll() is None]
            for p in live:
                try:
                    os.killpg(p.pid, signal.SIGTERM)          # the rank's own process group (start_new_session): exact PIDs, no pattern
                except ProcessLookupError:
                    pass
            t_end = time.time() + grace_s
            for p in live:
                try:
                    p.wait(timeout=max(0.0, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    try:
                        os.killpg(p.pid, signal.SIGKILL)
                    except ProcessLookupError:
                        pass
                    p.wait()
            rcs = [p.returncode for p in procs] + [None] * (n - len(procs))
            if interrupted:
                return 130
    finally:
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
        signal.signal(signal.SIGINT, old_int)
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write(f"bench.py: ranks failed (rank, rc): {bad}\n")
        return 1
    return 0


def launch_check(args, world, rank):
    """No-GPU dry run of the multi-rank plumbing (CPU test of the launcher)."""
    import torch
    import torch.distributed as dist
    from plangen_amd.dist import gather_rows, shard_range
    assert world == args.gpus, (world, args.gpus)
    assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
    if world > 1:
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=float(os.environ.get("PG_DIST_TIMEOUT_S", "900"))))
    die = os.environ.get("PG_TEST_DIE_RANK")                # test hook: this rank exits before the first collective
    if die is not None and int(die) == rank:
        sys.stderr.write(f"bench.py: rank {rank} dying on request (PG_TEST_DIE_RANK)\n")
        os._exit(7)
    hang = os.environ.get("PG_TEST_HANG_S")                 # test hook: every rank reports its pid, then sleeps (launcher-signal test)
    if hang is not None:
        with open(os.path.join(os.environ["PG_TEST_PID_DIR"], f"rank{rank}.pid"), "w") as f:
            f.write(str(os.getpid()))
        time.sleep(float(hang))
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
    G = args.global_batch or args.batch * world
    lo, hi = shard_range(G, world, rank)
    rows = gather_rows(torch.arange(lo, hi, dtype=torch.int32)[:, None].repeat(1, 3), G)
    if rank == 0:
        ok = float(t) == world * (world + 1) / 2 and rows[:, 0].tolist() == list(range(G))
        print(json.dumps({"launch_check": "ok" if ok else "FAILED", "world": world, "global_images": G,
                          "images_rank0": hi - lo, "backend": "gloo"}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def loop_roofline(cfg, dtype, B, L, T, pad, decode_ms, uncond_shared):
    """Whole-loop HBM roofline of THIS configuration, two byte counts over the 8 TB/s peak:
      algorithmic_gb  SURVEY 8d's definition -- per decode step the weights once (24 layers + gen_head + gen_aligner, compute dtype) + the full
                      K/V of every row at the PADDED length (no pad skipping, no shared negative prompt): the survey's conservative count;
      moved_gb        what the kernels have to stream: left-pad slots are never stored or read, and a batch-constant negative prompt is
                      stored once and its K/V read from HBM once per (layer, step) -- the other uncond rows hit it in L2.  The savings are
                      real wins, but the time-over-bytes fraction of the hardware is the ``moved_*`` one."""
    esz_ = 2 if dtype == "bf16" else 4
    w_step = (cfg.n_layers * WEIGHT_PARAMS_LAYER + cfg.gen_head_dim * cfg.hidden + cfg.img_vocab * cfg.gen_head_dim
              + cfg.hidden * cfg.img_dim + cfg.hidden * cfg.hidden) * esz_
    kv_key = cfg.n_layers * 2 * cfg.n_heads * cfg.head_dim * esz_                    # bytes per (row, key)
    floor_bytes = sum(w_step + 2 * B * (L + t) * kv_key for t in range(1, T))
    real = [L - p for p in pad]
    cond_keys = sum(real[0::2])
    if uncond_shared:
        unc_prompt_keys, unc_rows = real[1], B                                         # the prompt once, every row's private suffix
    else:
        unc_prompt_keys, unc_rows = sum(real[1::2]), B
    moved = sum(w_step + (cond_keys + B * t + unc_prompt_keys + unc_rows * t) * kv_key for t in range(1, T))
    floor_ms = floor_bytes / (HBM_PEAK_GBS * 1e9) * 1e3
    moved_ms = moved / (HBM_PEAK_GBS * 1e9) * 1e3
    return {"bound": "hbm", "floor_ms": floor_ms, "measured_ms": decode_ms, "frac": floor_ms / max(decode_ms, 1e-9),
            "algorithmic_gb": floor_bytes / 1e9, "moved_gb": moved / 1e9, "moved_frac": moved_ms / max(decode_ms, 1e-9),
            "moved_gbs": moved / 1e9 / max(decode_ms * 1e-3, 1e-12), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "what": "frac = SURVEY 8d decode-loop byte count of this batch (weights once per step + full K/V of all rows at the padded length) / 8 TB/s over the "
                    "measured loop of the last timed step; moved_* = the bytes the kernels actually stream (pads skipped, shared negative prompt read once)"}


def instrumented_pass(eng, args, cfg, B, L, T, ids, pad, tm, phase_ms=None):
    """Right after the timed region, same batch, eager launches: HIP events ON THE LAUNCH STREAM around every decode kernel class on every
    --time-stride-th step.  Dominant kernel = decode attention (HBM-bound K/V streaming).  Returns the ``roofline`` object (or None)."""
    import torch
    eng.set_option("time_attn", 1)
    eng.set_option("time_stride", args.time_stride)
    eng.prefill(ids, pad, position_mode=0)
    eng.decode_image_tokens(T=T, cfg_weight=cfg.cfg_weight, temperature=args.temperature, seed=99)
    torch.cuda.synchronize()
    ta = eng.timing()
    cls = eng.class_timing()
    eng.set_option("time_attn", 0)
    if not ta["attn_launches"]:
        return None
    ach = ta["attn_bytes_sum"] / (ta["attn_ms_sum"] * 1e-3) / 1e9
    kname = "attn_decode_fused_kernel"
    # PMC traffic cannot be collected from inside this process: it comes from separate rocprofv3 --pmc passes of this same command
    # (tools/pmc_traffic.py writes profiles/pmc_attn.json with the kernel symbol and the source hash it was measured at).  A ratio
    # measured on another kernel symbol / source is not applied.
    traffic = None
    pj = os.path.join(ROOT, "profiles", "pmc_attn.json")
    if os.path.exists(pj):
        pm = json.load(open(pj))
        if kname in pm.get("kernel_symbol", "") and pm.get("kernel_src_sha") == kernel_src_sha():
            traffic = pm["traffic_per_algorithmic_byte"] * ta["attn_bytes_sum"] / ta["attn_launches"]
    rf = {"bound": "hbm", "kernel": kname + " (RoPE + KV append + decode attention)",
          "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
          "launches": ta["attn_launches"], "avg_launch_us": ta["attn_ms_sum"] / ta["attn_launches"] * 1e3,
          "algorithmic_bytes_per_launch": ta["attn_bytes_sum"] / ta["attn_launches"],
          "decode_loop_share": ta["attn_ms_sum"] * args.time_stride / max(ta["decode_ms"], 1e-9),
          "timed_every_nth_step": args.time_stride}
    # Event timing of SHORT kernels is pessimistic: an event pair around nothing already costs a few us in this eager pass
    # ("event_pair_overhead_us"; measured ~4.7 us, about half of it lands inside a timed interval).  "avg_launch_us" is
    # the raw event interval; the rocprofv3 --kernel-trace averages of the replayed loop are in profiles/.
    empty = cls.get("empty_event_pair", {"ms_sum": 0.0, "launches": 0})
    rf["event_pair_overhead_us"] = empty["ms_sum"] / empty["launches"] * 1e3 if empty["launches"] else None
    classes = {}
    for name, c in cls.items():
        if not c["launches"] or name == "empty_event_pair":
            continue
        gbs = c["bytes_sum"] / (c["ms_sum"] * 1e-3) / 1e9
        classes[name] = {"bound": "hbm", "avg_launch_us": c["ms_sum"] / c["launches"] * 1e3,
                         "algorithmic_mb_per_launch": c["bytes_sum"] / c["launches"] / 1e6,
                         "achieved_gbs": gbs, "frac": gbs / HBM_PEAK_GBS, "launches": c["launches"]}
    if not args.tiny and T == cfg.img_tokens:
        # MFMA-bound phases, whole-phase times of the LAST TIMED step (HIP events inside the library):
        ntok = sum(L - p for i, p in enumerate(pad) if i % 2 == 0) + (L - pad[1])        # shared uncond prompt prefilled once
        lens = [L - p for i, p in enumerate(pad) if i % 2 == 0] + [L - pad[1]]
        fl = 2.0 * ntok * cfg.n_layers * WEIGHT_PARAMS_LAYER + sum(4.0 * cfg.n_layers * cfg.hidden * n * (n + 1) / 2 for n in lens)
        p_ms = phase_ms["prefill"]["mean"] if phase_ms and "prefill" in phase_ms else tm["prefill_ms"]
        tf = fl / (p_ms * 1e-3) / 1e12
        classes["prefill (packed GEMMs + flash attention)"] = {"bound": "mfma", "gflop": fl / 1e9, "ms": p_ms, "ms_last_step": tm["prefill_ms"],
                                                               "achieved_tflops": tf, "frac": tf / MFMA_PEAK_TFLOPS,
                                                               "timed": "mean over the timed steps (events around pg_prefill)" if phase_ms else "last timed step"}
        fl = VQ_DECODE_GFLOP_PER_IMAGE * 1e9 * B
        v_ms = phase_ms["vq_decode"]["mean"] if phase_ms and "vq_decode" in phase_ms else tm["vq_ms"]
        tf = fl / (v_ms * 1e-3) / 1e12
        classes["vq_decode (convs + GroupNorm + AttnBlock)"] = {"bound": "mfma", "gflop": fl / 1e9, "ms": v_ms, "ms_last_step": tm["vq_ms"],
                                                                "achieved_tflops": tf, "frac": tf / MFMA_PEAK_TFLOPS}
    # rocprofv3 kernel durations of the same classes (the event-timed intervals above read ~2 us high on short kernels): attached only when
    # profiles/kernel_classes_b<B>.json was measured on exactly the kernel sources this process runs (tools/trace_classes.py)
    kj = os.path.join(ROOT, "profiles", "kernel_classes_b%d.json" % B)
    if os.path.exists(kj):
        kc = json.load(open(kj))
        if kc.get("csrc_sha") == csrc_sha():
            for cn, ce in kc["classes"].items():
                if cn in classes and classes[cn].get("algorithmic_mb_per_launch"):
                    gbs = classes[cn]["algorithmic_mb_per_launch"] * 1e6 / (ce["avg_us"] * 1e-6) / 1e9
                    classes[cn].update({"rocprof_avg_launch_us": ce["avg_us"], "rocprof_achieved_gbs": gbs, "rocprof_frac": gbs / HBM_PEAK_GBS})
            rf["rocprof_classes_source"] = kc.get("source")
            if "decode_attention" in kc["classes"]:
                a_us = kc["classes"]["decode_attention"]["avg_us"]
                rf["rocprof_avg_launch_us"] = a_us
                rf["rocprof_frac"] = rf["algorithmic_bytes_per_launch"] / (a_us * 1e-6) / 1e9 / HBM_PEAK_GBS
    rf["classes"] = classes
    rf["_class_sums"] = cls
    return rf


def gemm_norm_phase(args, cfg, B, L, T, ids, pad, cls, device):
    """The decode GEMM + RMSNorm phase (everything of a layer except attention), two measurements:
      events_*   sum of the per-launch HIP-event intervals of the instrumented pass (one event pair, ~4.6 us, per launch on ~150 short
                 launches per step: pessimistic);
      (plain)    the SAME decode step replayed, as the timed region launches it, WITHOUT its 24 attention launches -- a switch that exists
                 only in libplangen_diag.so (``skip_attn``; results are garbage by construction) on a second handle with the same weights,
                 minus the event-timed gen_head + sampler share (5 launches per step)."""
    import torch
    from plangen_amd.engine import Engine
    dsum = sum(c["ms_sum"] for n, c in cls.items() if n.startswith("decode_gemm") or n == "decode_rmsnorm")
    dby = sum(c["bytes_sum"] for n, c in cls.items() if n.startswith("decode_gemm"))
    if dsum <= 0:
        return None
    nsteps_timed = max(1, cls["decode_gemm_qkv"]["launches"] // cfg.n_layers)
    ph = {"events_weight_gbs": dby / (dsum * 1e-3) / 1e9, "events_frac": dby / (dsum * 1e-3) / 1e9 / HBM_PEAK_GBS,
          "events_ms_per_step": dsum / nsteps_timed}
    try:
        deng = Engine(cfg, dtype=args.dtype, max_rows=2 * B, max_prompt=L, max_new=cfg.img_tokens, max_images=B, device=device, diag=True)
    except Exception as ex:                                   # noqa: BLE001 -- the diagnostics library is optional for a bench line
        ph["skip_attn_pass"] = "unavailable: " + repr(ex)[:200]
        return ph
    try:
        deng.init_synthetic(seed=0)
        for kv in args.opt:
            k, v = kv.split("=")
            deng.set_option(k, int(v))
        Tn = min(T, 96)
        deng.set_diag_option("skip_attn", 1)
        for _ in range(2):
            deng.prefill(ids, pad, position_mode=0)
            deng.decode_image_tokens(T=Tn, cfg_weight=cfg.cfg_weight, temperature=args.temperature, seed=7)
            torch.cuda.synchronize()
        t2 = deng.timing()
    finally:
        deng.close()
    head_ms = sum(c["ms_sum"] / max(1, c["launches"]) for n, c in cls.items()          # one event interval per step each
                  if n in ("decode_gen_head", "decode_cfg_sampler"))
    g_ms = t2["decode_ms"] / Tn - head_ms
    if g_ms <= 0:                      # tiny shapes under heavy external load: the event-timed head can exceed the replayed step; report the step itself
        g_ms = t2["decode_ms"] / Tn
        ph["head_not_subtracted"] = True
    w_step = dby / nsteps_timed
    ph.update({"ms_per_step": g_ms, "weight_gbs": w_step / (g_ms * 1e-3) / 1e9, "frac": w_step / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "method": "decode step replayed without its attention launches on a libplangen_diag.so handle (%d steps, same launch mode as the timed region) minus "
                         "event-timed gen_head + sampler (%.3f ms/step); events_* = sum of per-launch HIP-event intervals" % (Tn, head_ms)})
    return ph


def pipelined_pass(eng, args, cfg, B, T, ids, pad, uncond_shared, n_batches):
    """Cross-batch overlap experiment (a ``validation``-style stream of batches, plangen_base.py:1087-1181): pg_vq_decode of batch k runs on a SECOND
    stream while prefill + the decode loop of batch k+1 run on the main one -- the VQ decoder is MFMA-bound and touches nothing the loop owns except the
    token tensor it reads (its own activations / GroupNorm workspaces), the loop's GEMM + norm phase leaves MFMA and HBM mostly idle.  Serial and
    pipelined runs use the same seeds; tokens must be identical and the images equal.  Reported beside the headline (SURVEY 8d defines images/s on the
    serial sum)."""
    import torch
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream()

    def run(overlap):
        toks_all, imgs = [], []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pend = None                                   # (tokens, event) of the batch whose pixels are still to be decoded
        for k in range(n_batches):
            eng.prefill(ids, pad, position_mode=0, uncond_shared=uncond_shared)
            toks = eng.decode_image_tokens(T=T, cfg_weight=cfg.cfg_weight, temperature=args.temperature, seed=500 + k)
            if overlap:
                ev = torch.cuda.Event(); ev.record(main)
                if pend is not None:
                    imgs.append(pend)
                side.wait_event(ev)                   # tokens of batch k are complete
                with torch.cuda.stream(side):
                    pend = eng.vq_decode(toks)        # enqueued now, runs under batch k+1's prefill + loop on the main stream
            else:
                imgs.append(eng.vq_decode(toks))
            toks_all.append(toks)
        if overlap:
            imgs.append(pend)
            main.wait_stream(side)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, toks_all, imgs

    run(False)                                        # warm both orders once (side-stream first use, allocator)
    run(True)
    dt_s, tok_s, img_s = run(False)
    dt_p, tok_p, img_p = run(True)
    same_tok = all(torch.equal(a, b) for a, b in zip(tok_s, tok_p))
    same_img = all(torch.equal(a, b) for a, b in zip(img_s, img_p))
    return {"batches": n_batches, "serial_images_per_s": B * n_batches / dt_s, "pipelined_images_per_s": B * n_batches / dt_p,
            "gain": dt_s / dt_p - 1.0, "tokens_identical": bool(same_tok), "images_identical": bool(same_img),
            "what": "pg_vq_decode of batch k on a second stream under prefill + decode loop of batch k+1; same seeds as the serial run beside it"}


def secondary_workloads(args, cfg, device):
    """BASELINE.json configs[1], [2] and [4] at FULL length under the driver's clock: after the timed region and never part
    of ``value``; one warm-up + one timed step each on a fresh engine sized for the workload (the headline engine is closed first).
      uni bs=8          prefill L=256 + 576-step CFG loop + VQ decode                                              (plangen_base.py:525-607)
      uni_2stage bs=32  stage-1 prompt L1=128, 256 forced layout tokens (EOS suppressed), then the uni path above    (:1112-1127, :513-523)
      mmu bs=64         SigLIP-L + aligner on 64 images, prefill of 576 + 64 embeddings per row, 256 forced tokens   (:366, :513-523)
    Text decode (K13: 24 layers + lm_head + argmax per step) gets its own HBM roofline class: bytes per step = layer weights + lm_head (419 MB)
    + the K/V of every row at the step's context, over the measured ms per step."""
    import torch
    from plangen_amd.engine import Engine
    esz_ = 2 if args.dtype == "bf16" else 4
    w_layers = cfg.n_layers * WEIGHT_PARAMS_LAYER * esz_
    w_lm = cfg.vocab * cfg.hidden * esz_
    kv_key = cfg.n_layers * 2 * cfg.n_heads * cfg.head_dim * esz_
    NT, T = 256, cfg.img_tokens
    g = torch.Generator().manual_seed(0)

    def timed(fn):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        return r, (time.perf_counter() - t0) * 1e3

    def text_class(rows, ctx0, ms_total):
        by = sum(w_layers + w_lm + rows * (ctx0 + t) * kv_key for t in range(NT))
        gbs = by / (ms_total * 1e-3) / 1e9
        return {"bound": "hbm", "ms_per_step": ms_total / NT, "algorithmic_mb_per_step": by / NT / 1e6, "achieved_gbs": gbs, "frac": gbs / HBM_PEAK_GBS,
                "what": "per step: 24 layers of weights + lm_head once + K/V of %d rows at contexts %d..%d" % (rows, ctx0, ctx0 + NT - 1)}

    out = {}
    t_all = time.perf_counter()
    # ---- configs[1]: uni bs=8
    B = 8
    e = Engine(cfg, dtype=args.dtype, max_rows=2 * B, max_prompt=256, max_new=T, max_images=B, device=device)
    try:
        e.init_synthetic(seed=0)
        ids, mask = synth_prompts(B, 256, cfg.vocab, cfg.pad_id, seed=0)
        pad = Engine.pad_len_from_mask(torch.cat([mask, torch.ones((2 * B, T), dtype=torch.int32)], 1), 256)
        for s_ in range(2):
            _, t_p = timed(lambda: e.prefill(ids, pad, position_mode=0))
            toks, t_l = timed(lambda: e.decode_image_tokens(T=T, cfg_weight=cfg.cfg_weight, temperature=args.temperature, seed=s_))
            _, t_v = timed(lambda: e.vq_decode(toks))
        tot = t_p + t_l + t_v
        lr = loop_roofline(cfg, args.dtype, B, 256, T, pad, t_l, Engine.uncond_rows_shared(ids, pad))
        out["uni_bs8"] = {"workload": "task_type='uni' layout2image, bs=8, L=256, 576 image tokens (BASELINE configs[1])", "images_per_s": B / tot * 1e3,
                          "ms": {"prefill": t_p, "decode_loop": t_l, "vq_decode": t_v}, "total_ms": tot,
                          "loop_roofline": {k: lr[k] for k in ("frac", "moved_frac", "algorithmic_gb", "moved_gb")}}
    finally:
        e.close()
    # ---- configs[2]: uni_2stage bs=32
    B, L1 = 32, 128
    e = Engine(cfg, dtype=args.dtype, max_rows=2 * B, max_prompt=256, max_new=T, max_images=B, with_lm_head=True, device=device)
    try:
        e.init_synthetic(seed=0)
        ids1 = torch.randint(10, cfg.vocab - 2048, (B, L1), generator=g).int()
        ids2, mask2 = synth_prompts(B, 256, cfg.vocab, cfg.pad_id, seed=0)
        pad2 = Engine.pad_len_from_mask(torch.cat([mask2, torch.ones((2 * B, T), dtype=torch.int32)], 1), 256)
        for s_ in range(2):
            _, t_p1 = timed(lambda: e.prefill(ids1, [0] * B, position_mode=1))
            _, t_txt = timed(lambda: e.generate_text_greedy(NT, cfg.eos_id, min_new_tokens=NT))
            _, t_p2 = timed(lambda: e.prefill(ids2, pad2, position_mode=0))
            toks, t_img = timed(lambda: e.decode_image_tokens(T=T, cfg_weight=cfg.cfg_weight, temperature=args.temperature, seed=s_))
            _, t_vq = timed(lambda: e.vq_decode(toks))
        tot = t_p1 + t_txt + t_p2 + t_img + t_vq
        out["uni_2stage_bs32"] = {"workload": "task_type='uni_2stage', bs=32: stage-1 prompt L1=128 + 256 forced layout tokens, then L=256 + 576 image tokens + VQ decode (BASELINE configs[2])",
                                  "images_per_s": B / tot * 1e3, "total_ms": tot,
                                  "ms": {"prefill_stage1": t_p1, "text_decode": t_txt, "prefill_stage2": t_p2, "image_decode": t_img, "vq_decode": t_vq},
                                  "text_tokens_per_s": B * NT / t_txt * 1e3, "text_decode_roofline": text_class(B, L1, t_txt)}
    finally:
        e.close()
    # ---- configs[4]: mmu bs=64
    B, P, Lt = 64, cfg.vit_tokens, 64
    L = P + Lt
    e = Engine(cfg, dtype=args.dtype, max_rows=B, max_prompt=L, max_new=NT, max_images=B, with_lm_head=True, with_vq_encoder=True, with_vision=True,
               max_vision_images=B, device=device)
    try:
        e.init_synthetic(seed=0)
        pix = (torch.rand(B, 3, cfg.vit_img, cfg.vit_img, generator=g) * 2 - 1).to(e.device)
        txt = e.embed_tokens(torch.randint(10, cfg.vocab - 2048, (B, Lt), generator=g).int())
        cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        for s_ in range(2):
            feats, t_vit = timed(lambda: e.vision_encode(pix, dtype=cdt))
            emb = torch.cat([txt[:, :1].to(feats.dtype), feats, txt[:, 1:].to(feats.dtype)], 1).contiguous()
            _, t_pre = timed(lambda: e.prefill_embeds(emb, [0] * B, position_mode=1))
            _, t_txt = timed(lambda: e.generate_text_greedy(NT, cfg.eos_id, min_new_tokens=NT))
            _, t_enc = timed(lambda: e.vq_encode(pix))
        tot = t_vit + t_pre + t_txt
        out["mmu_bs64"] = {"workload": "task_type='mmu', bs=64: SigLIP-L/16-384 + aligner, prefill of 640 embeddings per row, 256 forced text tokens (BASELINE configs[4])",
                           "samples_per_s": B / tot * 1e3, "total_ms": tot, "ms": {"vision_encode": t_vit, "prefill": t_pre, "text_decode": t_txt},
                           "vq_encode_beside_ms": t_enc, "text_tokens_per_s": B * NT / t_txt * 1e3, "text_decode_roofline": text_class(B, L, t_txt),
                           "vision_tflops": 2 * 303e6 * P * B / (t_vit * 1e-3) / 1e12}
    finally:
        e.close()
    out["wall_s"] = time.perf_counter() - t_all
    out["what"] = "one warm-up + one timed step per workload on its own engine, after the headline's timed region; never part of `value`"
    return out


# ------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a mislabelled number\n")
        return 2
    if args.launch_check:
        return launch_check(args, world, rank)

    # Host placement BEFORE anything touches the GPU (and before torch starts its thread pools): pin this rank to the CPUs of its GPU's NUMA
    # node, disjoint from the other local ranks' slices (plangen_amd/affinity.py; sysfs only, never execs).  Reported in the line.
    from plangen_amd import affinity
    placement = affinity.apply(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))

    import torch
    import torch.distributed as dist
    from plangen_amd.config import PlanGenConfig
    from plangen_amd.dist import broadcast_prompts, gather_rows
    from plangen_amd.engine import Engine

    if os.environ.get("PG_FORCE_DEVICE") is not None:      # debugging aid: several ranks on one GPU (gloo only)
        local = int(os.environ["PG_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = "none"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("PG_DIST_BACKEND", "nccl")    # "nccl" is RCCL on ROCm
        import datetime
        tmo = datetime.timedelta(seconds=float(os.environ.get("PG_DIST_TIMEOUT_S", "900")))      # finite: a dead peer fails the collective
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)

    cfg = PlanGenConfig.tiny() if args.tiny else PlanGenConfig.janus_pro_1b()
    L = args.prompt_len
    strong = args.global_batch is not None
    G = args.global_batch if strong else args.batch * world              # images in the global batch
    if G % world:
        sys.stderr.write(f"bench.py: global batch {G} does not divide over {world} ranks\n")
        return 2
    B = G // world
    T = args.tokens or cfg.img_tokens
    eng = Engine(cfg, dtype=args.dtype, max_rows=2 * B, max_prompt=L, max_new=cfg.img_tokens, max_images=B, device=local, diag=bool(args.diag_opt))
    eng.init_synthetic(seed=0)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    for kv in args.diag_opt:
        k, v = kv.split("=")
        eng.set_diag_option(k, int(v))

    # rank 0 collates the global batch and broadcasts it over RCCL; every rank keeps its contiguous slice
    if rank == 0:
        g_ids, g_mask = synth_prompts(G, L, cfg.vocab, cfg.pad_id, seed=0)
        g_mask = torch.cat([g_mask, torch.ones((2 * G, cfg.img_tokens), dtype=torch.int32)], dim=1)
    else:
        g_ids = g_mask = None
    ids, mask, lo, hi, nB = broadcast_prompts(g_ids, g_mask, dev)
    pad = Engine.pad_len_from_mask(mask, L)
    eng.set_option("rng_image_offset", lo)       # sampling noise keyed on the GLOBAL image index: sharding does not change tokens
    # The collate builds the uncond rows on the HOST by replicating one negative prompt (plangen_base.py:672-686), so "do all uncond rows
    # share their ids" is host knowledge: compared once here on the host copy and handed to every pg_prefill as a hint -- the library then
    # skips its device probe (a 4-byte read + stream sync per batch).
    uncond_shared = Engine.uncond_rows_shared(ids.cpu(), pad)

    phase_events = []            # per timed step: 4 events on the launch stream (recorded asynchronously, read after the final fence)
    enqueue_ms = []              # per timed step: host wall time of the call that enqueues the 576-step loop (~100 k launches; no sync inside)

    def step(seed, n_tok=T, decode_pixels=True, record=False):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if record else None
        if ev: ev[0].record()
        eng.prefill(ids, pad, position_mode=0, uncond_shared=uncond_shared)
        if ev: ev[1].record()
        h0 = time.perf_counter()
        toks = eng.decode_image_tokens(T=n_tok, cfg_weight=cfg.cfg_weight, temperature=args.temperature, seed=seed)
        if ev:
            enqueue_ms.append((time.perf_counter() - h0) * 1e3)
            ev[2].record()
        img = eng.vq_decode(toks) if (decode_pixels and n_tok == cfg.img_tokens) else None
        if ev:
            ev[3].record()
            phase_events.append(ev)
        all_toks = gather_rows(toks, nB)          # rank 0: [G, T]; others: None
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
        step(k, record=True)
    fence()
    dt_local = time.perf_counter() - t0
    dt = dt_local
    rank_ms = [dt_local / max(args.steps, 1) * 1e3]
    if world > 1:
        t = torch.tensor([dt_local], dtype=torch.float64, device=dev)
        ts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(ts, t)
        rank_ms = [float(v.item()) / max(args.steps, 1) * 1e3 for v in ts]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tm = eng.timing()
    # Did the enqueue thread stay ahead of the GPU?  The enqueue call of a whole 576-step loop (~100 k launches) runs into the HIP queue's finite depth
    # and then proceeds at the GPU's pace, so its wall time says nothing (measured round 6: 1 494 ms of host time beside 1 622 ms of device time).  The probe:
    # after a fence, ONE 16-step decode (2.8 k launches: fits the queue) -- host time of the enqueue call against the device time of the same 16 steps.
    # margin = 1 - host / device; <= 0 means this rank is launch-bound (the queue runs dry between kernels).  Outside the timed region.
    host = {"affinity": placement}
    if T > 16:
        eng.prefill(ids, pad, position_mode=0, uncond_shared=uncond_shared)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0 = time.perf_counter(); e0.record()
        eng.decode_image_tokens(T=16, cfg_weight=cfg.cfg_weight, temperature=args.temperature, seed=31337)
        h1 = time.perf_counter(); e1.record()
        torch.cuda.synchronize()
        dev16 = e0.elapsed_time(e1)
        host.update({"probe": "16 decode steps enqueued behind a fence", "enqueue_ms_per_step": (h1 - h0) * 1e3 / 16, "device_ms_per_step": dev16 / 16,
                     "margin": 1.0 - (h1 - h0) * 1e3 / max(dev16, 1e-9), "timed_loop_enqueue_call_ms": enqueue_ms[0] if enqueue_ms else None})
        host["enqueue_ahead"] = bool(host["margin"] > 0.05)
    else:
        host.update({"margin": None, "enqueue_ahead": None})
    if world > 1:
        hosts = [None] * world
        dist.all_gather_object(hosts, {"rank": rank, "margin": host["margin"], "enqueue_ms_per_step": host.get("enqueue_ms_per_step"),
                                       "cpulist": placement.get("cpulist"), "numa_node": placement.get("numa_node"), "applied": placement.get("applied")})
        host["ranks"] = hosts
    images = G * args.steps
    out = {
        "metric": "images/sec, 384px layout2image (576 image tokens, CFG, VQ-16 decode), bs=%d per MI355X" % B,
        "value": images / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic prompts (SURVEY 8d), seeded random-init Janus-Pro-1B-shaped weights",
        "config": {"workload": "task_type='uni' layout2image, %dx%d / %d image tokens, bs=%d per GPU (global %d), L=%d left-padded, "
                               "cfg_weight=%g, temperature=%g" % (cfg.img_size, cfg.img_size, T, B, G, L, cfg.cfg_weight, args.temperature),
                   "images_per_gpu": B, "global_batch": G, "prompt_len": L, "rows_per_gpu": 2 * B,
                   "parallelism": "prompt-sharded x%d" % world},
        "image_tokens_per_sec_per_gpu": B * T * args.steps / dt,
        "last_step_ms": {"prefill": tm["prefill_ms"], "decode_loop": tm["decode_ms"], "vq_decode": tm["vq_ms"]},
        # every timed step's phases (events on the launch stream around the three calls of a step): mean and min over the K steps; the
        # MFMA-bound class fractions below use the MEAN (the last step alone is one sample of a quantity that moves +-2 % with the chip's clock)
        "phase_ms": {k: {"mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in (
            ("prefill", [e[0].elapsed_time(e[1]) for e in phase_events]), ("decode_loop", [e[1].elapsed_time(e[2]) for e in phase_events]),
            ("vq_decode", [e[2].elapsed_time(e[3]) for e in phase_events])) if v},
        "device_gb": eng.device_bytes() / 2 ** 30,
        "rccl_ranks": dist.get_world_size() if world > 1 else 1, "dist_backend": backend,
        "rank_ms_per_step": rank_ms,
        "host": host,
        "library": "libplangen_diag.so (MEASUREMENT BUILD: --diag-opt %s)" % " ".join(args.diag_opt) if args.diag_opt else "libplangen_hip.so",
    }
    if not args.tiny and T > 1:
        out["loop_roofline"] = loop_roofline(cfg, args.dtype, B, L, T, pad, tm["decode_ms"], uncond_shared)
        out["loop_roofline_frac"] = out["loop_roofline"]["frac"]

    if world > 1 and not args.no_shard_check:
        # The gathered token matrix must equal what ONE rank produces for the same seed: all ranks run a short
        # sampled pass (32 tokens), rank 0 then regenerates every other rank's shard itself (same prompts, same
        # global image indices in the RNG) and compares bit for bit.
        from plangen_amd.dist import shard_range
        n_chk = min(32, T)
        gathered, _ = step(4242, n_tok=n_chk, decode_pixels=False)
        if rank == 0:
            bad = 0
            for r in range(1, world):
                rlo, rhi = shard_range(G, world, r)
                ids_r = g_ids[2 * rlo:2 * rhi].to(dev)
                pad_r = Engine.pad_len_from_mask(g_mask[2 * rlo:2 * rhi], L)
                eng.set_option("rng_image_offset", rlo)
                eng.prefill(ids_r, pad_r, position_mode=0)
                mine = eng.decode_image_tokens(T=n_chk, cfg_weight=cfg.cfg_weight, temperature=args.temperature, seed=4242)
                bad += int((mine.cpu() != gathered[rlo:rhi].cpu()).sum())
            eng.set_option("rng_image_offset", lo)
            out["shard_check"] = {"tokens_compared": int((G - (hi - lo)) * n_chk), "mismatches": bad,
                                  "what": "gathered tokens of ranks 1..N-1 == rank 0's own run of those shards, same seed"}
        fence()

    if args.pipeline > 0 and rank == 0 and T == cfg.img_tokens:
        out["pipelined"] = pipelined_pass(eng, args, cfg, B, T, ids, pad, uncond_shared, args.pipeline)
        out["pipelined_images_per_s"] = out["pipelined"]["pipelined_images_per_s"]
    if not args.no_roofline and rank == 0:
        rf = instrumented_pass(eng, args, cfg, B, L, T, ids, pad, tm, out.get("phase_ms"))
        if rf:
            out["roofline"] = rf
            if not args.no_gemm_phase:
                # the decode step WITHOUT its attention launches needs the diagnostics library (no switch of libplangen_hip.so can make a handle
                # skip work): the product engine is released first, a second handle in libplangen_diag.so replays the same steps
                eng.close()
                ph = gemm_norm_phase(args, cfg, B, L, T, ids, pad, rf.pop("_class_sums"), local)
                if ph:
                    rf["decode_gemm_norm_phase"] = ph
            rf.pop("_class_sums", None)
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_secondary and not args.tiny and B == 64 and T == cfg.img_tokens and not args.diag_opt:
        eng.close()                                           # (idempotent: the GEMM-phase measurement above may have released it already)
        try:
            out["secondary"] = secondary_workloads(args, cfg, local)
        except Exception as ex:                               # noqa: BLE001 -- a secondary workload must never cost the headline line
            out["secondary"] = {"failed": repr(ex)[:400]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.tiny:
        # 16 threads: measured fastest for these small torch-CPU GEMMs on the 256-core bench host
        # (ms/step: 16 thr 89, 32 thr 144, 64 thr 298, 256 thr 42 466)
        out["cpu_baseline"] = cpu_baseline(L, args.cpu_steps, min(args.cpu_threads, os.cpu_count() or 1))
    if rank == 0 and world == 1 and not args.no_rccl_selftest and not args.tiny:
        # under rocprofv3 the child would inherit the profiler's preload and write a second results database into the same directory
        profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
        out["rccl_selftest"] = {"rccl_one_rank": "skipped under a profiler"} if profiled else rccl_selftest()
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()
    return 0


def rccl_selftest(timeout_s=180):
    """N = 1 only, after the timed region: a ONE-RANK RCCL communicator in a fresh child process (tests/helpers/rccl_one_rank.py: the
    broadcast / gather / all_gather / barrier of plangen_amd/dist.py on device tensors beside a tiny engine).  A single GPU cannot measure
    the scaling curve; this records, in the driver's own bench line, that librccl initialises and the path's collectives execute on this
    box.  Failure or timeout is REPORTED, never fatal: the N = 1 number does not depend on RCCL."""
    import json as _json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_one_rank.py"), "nccl"], env=env,
                           capture_output=True, text=True, timeout=timeout_s)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode == 0 and lines:
            return _json.loads(lines[-1])
        return {"rccl_one_rank": "failed", "rc": p.returncode, "stderr_tail": p.stderr[-400:]}
    except subprocess.TimeoutExpired:
        return {"rccl_one_rank": "timeout", "timeout_s": timeout_s}
    except Exception as ex:                                   # noqa: BLE001 -- reported in the line
        return {"rccl_one_rank": "failed", "error": repr(ex)[:300]}


def visible_gpus():
    """Device count from a THROW-AWAY child (the launcher itself must stay free of any GPU / HIP state); None when it cannot tell."""
    try:
        p = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
        return int(p.stdout.strip().splitlines()[-1])
    except Exception:                                          # noqa: BLE001
        return None


def csrc_sha():
    """Hash of every kernel source (tools/trace_classes.py computes the same): a rocprofv3 class table from other sources is not attached."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "plangen_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def kernel_src_sha():
    """Identity of the decode-attention kernel SOURCE a PMC ratio belongs to (hash of csrc/attn_decode.h, the kernel template, + the
    production launcher in llm_kernels.hip): a ratio measured on another version of the kernel is not applied."""
    import hashlib
    d = os.path.join(ROOT, "plangen_amd", "csrc")
    src = open(os.path.join(d, "llm_kernels.hip")).read()
    a = src.find("void launch_attn_decode_fused(")
    b = src.find("template void launch_attn_decode_fused<float>")
    return hashlib.sha256((open(os.path.join(d, "attn_decode.h")).read() + src[a:b]).encode()).hexdigest()[:16]


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # parent: start N fresh rank processes; no torch.cuda / HIP call has happened in this process
        if not args.launch_check and os.environ.get("PG_FORCE_DEVICE") is None:
            n = visible_gpus()
            if n is not None and n < args.gpus:
                sys.stderr.write(f"bench.py: --gpus {args.gpus} but this box exposes {n} GPU(s) (torch.cuda.device_count() in a child process); "
                                 "not starting ranks that would die with 'invalid device ordinal'\n")
                return 2
        return spawn_ranks(args.gpus, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
