"""``System(args, accelerator)`` with the methods ``train.py`` calls in test mode
(reference: project/plangen/plangen_base.py:80 ``__init__``, :980 ``setup_data``,
base_system.py:127 ``resume``, plangen_base.py:1087 ``validation``), backed by the MI355X engine.

Out of the path: tokenizer, datasets and PNG drawing.  Prompts therefore
come pre-tokenised -- a JSONL file with {"cond_ids": [...], "neg_ids": [...]} per line
(``test_data.ids_file``) -- or are synthetic (``test_data.data_name='synthetic'``).
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import torch

from plangen_amd.config import PlanGenConfig
from plangen_amd.dist import shard_range, world
from plangen_amd.engine import Engine
from plangen_amd.system import System as _HotPath, denorm_pt
from plangen_amd.weights import latest_checkpoint, load_checkpoint


class System(_HotPath):
    def __init__(self, args, accelerator=None):
        cfg = PlanGenConfig.janus_pro_1b() if not getattr(args, "tiny", False) else PlanGenConfig.tiny()
        cfg.seed, cfg.cfg_weight, cfg.temperature = args.seed, args.cfg_weight, args.temperature
        bs = int(args.test_batch_size)
        rank, ws = world()
        device = int(os.environ.get("LOCAL_RANK", "0"))
        eng = Engine(cfg, dtype=args.dtype, max_rows=2 * bs, max_prompt=int(getattr(args, "max_prompt", 768)),
                     max_new=cfg.img_tokens if args.test_data["task_type"] != "mmu" else 512, max_images=bs,
                     with_lm_head=args.test_data["task_type"] != "uni", with_vq_encoder=bool(args.use_teacher_forcing),
                     with_vision=args.test_data["task_type"] == "mmu", device=device)
        super().__init__(cfg, eng, SimpleNamespace(seed=args.seed, parallel_size=args.parallel_size, cfg_weight=args.cfg_weight,
                                                   temperature=args.temperature, use_teacher_forcing=args.use_teacher_forcing,
                                                   debug_max_seq_len=args.debug_max_seq_len, janus_hw=args.janus_hw))
        self.cli = args
        self.accelerator = accelerator

    # ------------------------------------------------------------------ train.py:90
    def setup_data(self, accelerator=None):
        a = self.cli
        rank, ws = world()
        td = a.test_data
        batches = []
        if td.get("ids_file"):
            rows = [json.loads(l) for l in open(td["ids_file"]) if l.strip()]
        else:
            g = torch.Generator().manual_seed(a.seed)
            hi = self.cfg.vocab - 2048 if self.cfg.vocab > 4096 else self.cfg.vocab
            neg = torch.randint(10, hi, (12,), generator=g).tolist()
            rows = [{"cond_ids": torch.randint(10, hi, (int(torch.randint(8, 48, (1,), generator=g)),),
                                               generator=g).tolist(), "neg_ids": neg}
                    for _ in range(a.test_batch_size * a.max_test_len)]
        lo, hi = shard_range(len(rows), ws, rank)                    # prompt sharding (plangen_base.py:994)
        rows = rows[lo:hi]
        bs = a.test_batch_size
        for i in range(0, len(rows), bs):
            batches.append(rows[i:i + bs])
        self.test_dataloader = batches[: a.max_test_len]
        return self.test_dataloader

    # ------------------------------------------------------------------ train.py:91
    def resume(self, accelerator=None):
        """Base Janus-Pro weights + the PlanGen overlay (plangen_base.py:104-113, base_system.py:127-160);
        seeded synthetic weights when no checkpoint directory exists (offline smoke runs)."""
        a = self.cli
        ck = latest_checkpoint(a.out_path) if a.resume == "latest" else a.resume
        if a.janus_path and os.path.isdir(str(a.janus_path)):
            load_checkpoint(self.engine, a.janus_path, overlay=ck if ck and os.path.exists(ck) else None, strict=True)
        elif ck:
            raise FileNotFoundError(f"resume={ck!r} needs the base weights at janus_path={a.janus_path!r}")
        else:
            self.engine.init_synthetic(seed=a.seed)
        return 0

    # ------------------------------------------------------------------ train.py:134-136
    @torch.no_grad()
    def validation(self, global_step=0):
        a = self.cli
        task = a.test_data["task_type"]
        out_dir = os.path.join(a.out_path, "test", f"{a.test_data['data_name']}_{task}", str(global_step), "pr_image")
        os.makedirs(out_dir, exist_ok=True)
        rank, _ = world()
        n_img = 0
        for idx, rows in enumerate(self.test_dataloader):
            cond = [r["cond_ids"] for r in rows]
            negs = [r["neg_ids"] for r in rows]
            shared = all(n == negs[0] for n in negs)
            ids, mask = self.t2i_infer_collate_batch(cond, negs[0] if shared else negs)
            dec, toks = self.t2i(ids, mask)
            img = (denorm_pt(dec.float()) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).cpu()
            for i in range(img.shape[0]):
                name = os.path.join(out_dir, f"r{rank}_{idx * a.test_batch_size + i:06d}")
                try:
                    from PIL import Image
                    Image.fromarray(img[i].numpy()).save(name + ".png")
                except Exception:
                    torch.save(img[i], name + ".pt")
                n_img += 1
            with open(os.path.join(out_dir, f"r{rank}_{idx:04d}_tokens.json"), "w") as f:
                json.dump(toks.cpu().tolist(), f)
        return {"images": n_img, "out_dir": out_dir}
