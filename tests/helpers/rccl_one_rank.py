"""Child process of tests/test_gpu_rccl.py (and of bench.py's ``rccl_selftest``): a ONE-RANK ``nccl`` (= RCCL) process group on the
box's single GPU next to a live engine.  Proves, with what one GPU allows, that librccl loads beside the engine's HIP runtime,
that ``dist.broadcast`` / ``dist.gather`` / ``dist.all_gather`` / ``dist.barrier`` of plangen_amd/dist.py run on device tensors on
this ROCm, and that generation on the engine's stream and RCCL's stream coexist (tokens before == tokens after, gathered == local).
The reference's equivalent: accelerate's process group, plangen_base.py:994.  Prints one JSON line; exit 0 = ok."""
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
    import torch
    import torch.distributed as dist
    from plangen_amd.config import PlanGenConfig
    from plangen_amd.dist import all_gather_rows, broadcast_prompts, gather_rows
    from plangen_amd.engine import Engine
    from plangen_amd.system import t2i_infer_collate_batch

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29653")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cfg = PlanGenConfig.tiny()
    dev = torch.device("cuda", 0) if backend == "nccl" else torch.device("cpu")
    t0 = time.time()
    tmo = datetime.timedelta(seconds=120)
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=tmo)
    else:
        dist.init_process_group(backend, rank=0, world_size=1, timeout=tmo)
    t_init = time.time() - t0
    g = torch.Generator().manual_seed(5)
    cond = [torch.randint(8, cfg.vocab, (n,), generator=g).tolist() for n in (10, 6, 12)]
    neg = torch.randint(8, cfg.vocab, (5,), generator=g).tolist()
    ids, mask = t2i_infer_collate_batch(cond, neg, cfg.pad_id, cfg.img_tokens)
    out = {"backend": backend, "init_s": round(t_init, 2)}
    if backend == "nccl":
        eng = Engine(cfg, dtype="bf16", max_rows=6, max_prompt=16, max_images=3)
        eng.init_synthetic(seed=0)

        def gen(i, p):
            eng.prefill(i, p, position_mode=0)
            return eng.decode_image_tokens(T=16, cfg_weight=5.0, temperature=0.0)
        L = ids.shape[1]
        before = gen(ids, Engine.pad_len_from_mask(mask, L)).clone()
        t0 = time.time()
        my_ids, my_mask, lo, hi, nB = broadcast_prompts(ids, mask, dev, force_collectives=True)
        assert (lo, hi, nB) == (0, 3, 3) and my_ids.is_cuda and torch.equal(my_ids.cpu(), ids.int())
        toks = gen(my_ids, Engine.pad_len_from_mask(my_mask, L))          # engine stream work between two RCCL collectives
        allt = gather_rows(toks, nB, force_collectives=True)
        everywhere = all_gather_rows(toks, nB, force_collectives=True)
        dist.barrier()
        torch.cuda.synchronize()
        out["collectives_s"] = round(time.time() - t0, 2)
        assert torch.equal(allt, toks) and torch.equal(everywhere, toks) and torch.equal(toks, before)
        out["nccl_version"] = list(torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None
        maps = open("/proc/self/maps").read()
        out["librccl_loaded"] = "librccl" in maps
        out["libplangen_loaded"] = "libplangen_hip.so" in maps
        assert out["librccl_loaded"] and out["libplangen_loaded"]
    else:
        my_ids, my_mask, lo, hi, nB = broadcast_prompts(ids, mask, dev, force_collectives=True)
        toks = my_ids[0::2, :4].contiguous()
        assert torch.equal(gather_rows(toks, nB, force_collectives=True), toks)
        assert torch.equal(all_gather_rows(toks, nB, force_collectives=True), toks)
        dist.barrier()
    dist.destroy_process_group()
    out["rccl_one_rank"] = "ok"
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
