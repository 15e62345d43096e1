"""``System(args, accelerator)`` with the methods ``train.py`` calls in test mode
(reference: project/plangen/plangen_base.py:80 ``__init__``, :980 ``setup_data``, :760-830 ``mmu_collate``,
base_system.py:127 ``resume``, plangen_base.py:1087-1181 ``validation``), backed by the MI355X engine.

Data: the reference's datasets (LayoutSAM, HICO, ...) are out of scope; rows come from a JSONL file
(``test_data.data_file``), one object per sample:
    {"base_caption": str, "gt_grounding": str, "image_id": str}          text, needs a tokenizer
  | {"cond_ids": [...], "stage1_ids": [...], "neg_ids": [...]}          pre-tokenised (no tokenizer files needed)
or are synthetic (``test_data.data_name='synthetic'``: seeded captions / layouts through the offline TagWordCodec).
The batches handed to ``uni_generate`` carry the reference's keys (``mmu_collate``, :771-795): base_caption, gt_grounding,
image_id, uni_inputs_ids / uni_attention_mask, uni_stage1_inputs_ids / uni_stage1_attention_mask, prepare_inputs_infer.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import torch

from plangen_amd.config import PlanGenConfig
from plangen_amd.dist import shard_range, world
from plangen_amd.engine import Engine, PlanGenError
from plangen_amd.system import System as _HotPath, denorm_pt
from plangen_amd.textproc import HFCodec, TagWordCodec
from plangen_amd.weights import latest_checkpoint, load_checkpoint

TASKS = ("t2i", "uni_2stage", "uni", "mmu", "plan")         # plangen_base.py:1112-1127

_WORDS = ("a photo of red blue green small large wooden metal cat dog car tree house table chair lamp bird boat "
          "on under next to in front behind the street field room sky river").split()


def save_image(chw: torch.Tensor, path: str) -> str:
    """to_pil(denorm_pt(x)).save(path) (plangen_base.py:1162-1181); torch.save when PIL is unavailable."""
    img = (denorm_pt(chw.float()) * 255).round().to(torch.uint8).permute(1, 2, 0).cpu()
    try:
        from PIL import Image
        Image.fromarray(img.numpy()).save(path)
        return path
    except ImportError:
        alt = os.path.splitext(path)[0] + ".pt"
        torch.save(img, alt)
        return alt


class System(_HotPath):
    def __init__(self, args, accelerator=None):
        td = args.test_data
        task = td["task_type"]
        if task not in TASKS:
            raise PlanGenError(f"test_data.task_type={task!r}: expected one of {TASKS} (plangen_base.py:1112-1127)")
        cfg = PlanGenConfig.janus_pro_1b() if not getattr(args, "tiny", False) else PlanGenConfig.tiny()
        cfg.seed, cfg.cfg_weight, cfg.temperature = args.seed, args.cfg_weight, args.temperature
        bs = int(args.test_batch_size)
        device = int(os.environ.get("LOCAL_RANK", "0"))
        self.synthetic = td.get("data_name") == "synthetic" or bool(getattr(args, "synthetic", False))
        tok_dir = str(args.janus_path) if args.janus_path else ""
        if os.path.exists(os.path.join(tok_dir, "tokenizer.json")) or os.path.exists(os.path.join(tok_dir, "tokenizer.model")):
            codec = HFCodec(tok_dir)
            cfg.eos_id, cfg.pad_id = codec.eos_token_id, codec.pad_id
        elif self.synthetic:
            codec = TagWordCodec(cfg.vocab, eos_id=cfg.eos_id, pad_id=cfg.pad_id)
        else:
            codec = None            # pre-tokenised rows only; text steps raise a clear error
        text_new = int(getattr(args, "max_new_tokens", 512))
        needs_image = task in ("t2i", "uni", "uni_2stage")
        eng = Engine(cfg, dtype=args.dtype, max_rows=2 * bs * int(args.parallel_size), max_prompt=int(getattr(args, "max_prompt", 768)),
                     max_new=max(cfg.img_tokens if needs_image else 1, text_new if task != "uni" and task != "t2i" else 1),
                     max_images=bs * int(args.parallel_size), with_lm_head=task in ("uni_2stage", "mmu", "plan"),
                     with_vq_encoder=bool(args.use_teacher_forcing), with_vision=task == "mmu", device=device)
        super().__init__(cfg, eng, SimpleNamespace(seed=args.seed, parallel_size=args.parallel_size, cfg_weight=args.cfg_weight,
                                                   temperature=args.temperature, use_teacher_forcing=args.use_teacher_forcing,
                                                   debug_max_seq_len=args.debug_max_seq_len, janus_hw=args.janus_hw,
                                                   neg_prompt=getattr(args, "neg_prompt", ""),
                                                   use_neg_box=bool(getattr(args, "use_neg_box", False))), codec=codec)
        self.cli = args
        self.accelerator = accelerator
        self.max_new_tokens = text_new

    # ------------------------------------------------------------------ train.py:90
    def _rows(self):
        a, td = self.cli, self.cli.test_data
        path = td.get("data_file") or td.get("ids_file")
        if path:
            if not os.path.exists(path):
                raise FileNotFoundError(f"test_data.data_file={path!r} does not exist")
            return [json.loads(l) for l in open(path) if l.strip()]
        if not self.synthetic:
            raise PlanGenError("no test data: set test_data.data_file=<jsonl> or test_data.data_name='synthetic'")
        g = torch.Generator().manual_seed(a.seed)
        pick = lambda n: " ".join(_WORDS[int(i)] for i in torch.randint(0, len(_WORDS), (n,), generator=g))
        rows = []
        for i in range(a.test_batch_size * a.max_test_len):
            objs = [pick(2) for _ in range(int(torch.randint(1, 4, (1,), generator=g)))]
            box = lambda: ",".join(str(int(v)) for v in torch.randint(0, 1000, (4,), generator=g))
            gr = "<grounding>" + "".join(f"<ref>{o}</ref><box>[{box()}]</box>" for o in objs) + "</grounding>"
            rows.append({"base_caption": pick(int(torch.randint(4, 12, (1,), generator=g))), "gt_grounding": gr, "image_id": ""})
        return rows

    def collate(self, rows):
        """mmu_collate's inference keys (plangen_base.py:771-795)."""
        T = self.image_token_num_per_image
        b = dict(base_caption=[r.get("base_caption", "") for r in rows], gt_grounding=[r.get("gt_grounding", "") for r in rows],
                 image_id=[r.get("image_id", "") for r in rows], image=None, prompt=[r.get("base_caption", "") for r in rows])
        if all("cond_ids" in r for r in rows):
            uni = [r["cond_ids"] for r in rows]
        else:
            uni = [self.wrap_uni_prompt(c, g)[1].tolist() for c, g in zip(b["base_caption"], b["gt_grounding"])]
        ids, m = self.pad_input_ids(uni)
        b["uni_inputs_ids"], b["uni_attention_mask"] = ids, torch.cat([m, torch.ones((len(rows), T), dtype=m.dtype)], -1)
        if all("stage1_ids" in r for r in rows):
            s1 = [r["stage1_ids"] for r in rows]
        elif self.codec is not None:
            s1 = [self.wrap_uni_prompt(c, "<grounding>", in_stage1=True)[1].tolist() for c in b["base_caption"]]
        else:
            s1 = None
        if s1 is not None:
            b["uni_stage1_inputs_ids"], b["uni_stage1_attention_mask"] = self.pad_input_ids(s1)
        if all("neg_base_caption" in r for r in rows):                  # use_neg_box rows (plangen_base.py:652-670)
            b["neg_base_caption"] = [r["neg_base_caption"] for r in rows]
            b["neg_gt_grounding"] = [r.get("neg_gt_grounding", "") for r in rows]
        for key, col in (("image", "image_pt"), ("edited_image", "edited_image_pt")):     # [3,S,S] tensors in [-1,1] saved with torch.save
            if all(col in r for r in rows):
                b[key] = torch.stack([torch.load(r[col]).float() for r in rows])
        negs = [r["neg_ids"] for r in rows if "neg_ids" in r]
        if len(negs) == len(rows):
            b["neg_inputs_ids"] = negs[0] if all(n == negs[0] for n in negs) else negs
        elif getattr(self.cli, "neg_prompt_ids", None):
            b["neg_inputs_ids"] = list(self.cli.neg_prompt_ids)
        if self.cli.test_data["task_type"] == "mmu":
            b["prepare_inputs_infer"] = self._mmu_inputs(rows)
        return b

    def _mmu_inputs(self, rows):
        """wrap_mmu_prompt's tensors (plangen_base.py:263-290), one image per sample, left-padded (processing_vlm.py batchify).
        With a tokenizer the text goes through the reference's conversation -- ``<|User|>: <image_placeholder>\\n{question}`` /
        empty assistant turn under VLChatProcessor's system prompt -- and the placeholder id is expanded to
        ``<begin_of_image>`` + vit_tokens slots + ``<end_of_image>`` (add_image_token, processing_vlm.py:243-248).
        Rows with ``question_ids`` (pre-tokenised) keep the bare form [first id] + slots + [rest]."""
        from plangen_amd.textproc import wrap_mmu_prompt_ids
        cfg = self.cfg
        P = cfg.vit_tokens
        g = torch.Generator().manual_seed(self.cli.seed + 17)
        seqs = []
        for r in rows:
            if r.get("question_ids"):
                q = r["question_ids"]
                seqs.append(([q[0]] + [0] * P + list(q[1:]), [False] + [True] * P + [False] * (len(q) - 1)))
            else:
                _, ids, slot = wrap_mmu_prompt_ids(self._codec(), r.get("question", r.get("base_caption", "")), P)
                seqs.append((ids, slot))
        L = max(len(s[0]) for s in seqs)
        ids = torch.full((len(rows), L), cfg.pad_id, dtype=torch.long)
        seq_mask = torch.zeros((len(rows), L), dtype=torch.bool)
        attn = torch.zeros((len(rows), L), dtype=torch.int32)
        for i, (s_ids, slot) in enumerate(seqs):
            n = len(s_ids)
            ids[i, L - n:] = torch.tensor(s_ids)
            seq_mask[i, L - n:] = torch.tensor(slot)
            attn[i, L - n:] = 1
        pix = torch.stack([torch.load(r["image_pt"]) if "image_pt" in r else torch.rand(3, cfg.vit_img, cfg.vit_img, generator=g) * 2 - 1
                           for r in rows])[:, None]
        return dict(input_ids=ids, pixel_values=pix, images_seq_mask=seq_mask, images_emb_mask=torch.ones((len(rows), 1, P), dtype=torch.bool),
                    attention_mask=attn)

    def setup_data(self, accelerator=None):
        """Prompt sharding (plangen_base.py:994) over WHOLE batches: rank r owns a contiguous run of test batches, so its first
        row sits on a multiple of test_batch_size and the reference's file names ``{idx*bs+i}`` (:1171-1176), computed with the
        GLOBAL batch index, are the global row index -- no two ranks can write the same file (sharding rows gave
        64 rows / 3 ranks / bs 8 colliding names)."""
        a = self.cli
        rank, ws = world()
        rows = self._rows()
        bs = int(a.test_batch_size)
        nb = (len(rows) + bs - 1) // bs
        lo, hi = shard_range(nb, ws, rank)
        self.batch_offset = lo
        self.row_offset = lo * bs
        rows = rows[lo * bs:hi * bs]
        self.test_dataloader = [self.collate(rows[i:i + bs]) for i in range(0, len(rows), bs)]
        return self.test_dataloader

    # ------------------------------------------------------------------ train.py:91
    def resume(self, accelerator=None):
        """Base Janus-Pro weights + the PlanGen overlay (plangen_base.py:104-113, base_system.py:127-160).  Seeded
        synthetic weights ONLY when asked for (data_name='synthetic' or synthetic=True): a mistyped janus_path or
        resume path is an error, not a silent random-weight run."""
        a = self.cli
        ck = latest_checkpoint(a.out_path) if a.resume == "latest" else a.resume
        if isinstance(ck, int) or (isinstance(ck, str) and ck.isdigit()):      # base_system.py:132-134: resume=<int> -> out_path/checkpoint-<int>
            ck = os.path.join(a.out_path, f"checkpoint-{int(ck)}")
        if a.resume not in (None, "latest") and not os.path.exists(str(ck)):
            raise FileNotFoundError(f"resume={a.resume!r}: {ck!r} does not exist")
        if a.janus_path and os.path.isdir(str(a.janus_path)):
            info = load_checkpoint(self.engine, a.janus_path, overlay=ck, strict=True)
            print(f"loaded {info['loaded'][0]} tensors from {a.janus_path}" + (f" + overlay {ck}" if ck else "")
                  + (f"; {len(info['skipped'])} tensors not owned by this engine" if info["skipped"] else ""))
        elif self.synthetic:
            if ck:
                raise FileNotFoundError(f"resume={ck!r} needs the base weights at janus_path={a.janus_path!r}")
            print("synthetic run: seeded random-init weights (no checkpoint was read)")
            self.engine.init_synthetic(seed=a.seed)
        else:
            raise FileNotFoundError(f"janus_path={a.janus_path!r} is not a directory (set test_data.data_name='synthetic' or synthetic=True "
                                    "for a random-weight plumbing run)")
        return 0

    # ------------------------------------------------------------------ train.py:134-136
    @torch.no_grad()
    def validation(self, global_step=0, accelerator=None, test_mode=True, val_num=None):
        """plangen_base.py:1087-1181, test mode: task dispatch (:1112-1127), uni_generate per batch, and the output tree
        ``{out}/test/{data_name}_{task_type}_{val_num}/{step}/{pr_image,gt_image,image_ids,gt_image_ids}`` with
        ``pr_image/{idx*bs+i}.png`` (``_{t}`` suffix when parallel_size > 1) plus ``{step}_batch/{idx}_layout.json``."""
        a = self.cli
        task = a.test_data["task_type"]
        val_num = val_num or a.max_test_len
        patha = os.path.join(a.out_path, "test", f"{a.test_data['data_name']}_{task}_{val_num}")
        path = os.path.join(patha, f"{global_step}")
        batch_path = os.path.join(patha, f"{global_step}_batch")
        for d in ("gt_image", "pr_image", "image_ids", "gt_image_ids"):
            os.makedirs(os.path.join(path, d), exist_ok=True)
        os.makedirs(batch_path, exist_ok=True)
        kwargs = {}
        if task == "t2i":
            kwargs.update(pred_layout=False, use_uni_prompt_in_t2i=False)
        elif task == "uni_2stage":
            pass
        elif task == "uni":
            kwargs.update(pred_layout=False)
        elif task == "mmu":
            kwargs.update(pred_image=False, is_mmu=True)
        elif task == "plan":
            kwargs.update(pred_image=False)
        else:
            raise PlanGenError(f"unknown task_type {task!r}")
        rank, ws = world()
        bs0 = a.test_batch_size
        n_img, layouts = 0, []
        for idx, batch in enumerate(self.test_dataloader):
            if val_num != -1 and idx >= val_num:
                break
            if idx < int(getattr(a, "test_start", 0)):
                continue
            gidx = getattr(self, "batch_offset", 0) + idx                  # GLOBAL batch index (whole batches are sharded): names == a single-rank run's
            out = self.uni_generate(batch=batch, batch_idx=f"{gidx}", gen_path=batch_path, save_local=True, max_new_tokens=self.max_new_tokens,
                                    **kwargs)
            pr_image, gt_image, image_id = out.get("pr_image"), batch.get("image"), batch["image_id"]
            edited_image = batch.get("edited_image")
            layouts.append(out.get("pr_grounding"))
            bs = len(image_id)
            p = int(a.parallel_size)
            for i in range(bs):
                if pr_image is not None:
                    if image_id[i] != "":
                        save_image(pr_image[i], f"{path}/image_ids/{image_id[i]}.jpg")
                        if gt_image is not None:
                            save_image(gt_image[i], f"{path}/gt_image_ids/{image_id[i]}.jpg")
                    if p > 1:
                        for t in range(p):
                            save_image(pr_image[i * p + t], f"{path}/pr_image/{gidx * bs0 + i}_{t}.png")
                    else:
                        save_image(pr_image[i * p], f"{path}/pr_image/{gidx * bs0 + i}.png")
                    n_img += 1
                if gt_image is not None:
                    save_image(gt_image[i], f"{path}/gt_image/{gidx * bs0 + i}.png")
                if edited_image is not None:                                   # plangen_base.py:1179-1181
                    os.makedirs(f"{path}/edited_image", exist_ok=True)
                    save_image(edited_image[i], f"{path}/edited_image/{gidx * bs0 + i}.png")
        return {"task_type": task, "images": n_img, "batches": len(layouts), "out_dir": path, "batch_dir": batch_path}
