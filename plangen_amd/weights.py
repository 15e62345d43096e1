"""Checkpoint formats of the path (SURVEY.md 8f-4, section 5 'Checkpoint / resume').

* base weights: HF ``deepseek-ai/Janus-Pro-1B`` directory (``*.safetensors`` shards, optionally a
  ``model.safetensors.index.json``) -- what ``AutoModelForCausalLM.from_pretrained(janus_path)`` reads
  (plangen_base.py:95);
* PlanGen overlay: ``checkpoint-*/trainable_model_parameters.pth`` written by ``Base_System.save_para``
  (base_system.py:166-189), a plain ``torch.save`` dict whose keys carry the ``vl_gpt.`` prefix; loaded
  by the reference with ``load_state_dict(strict=False)`` (base_system.py:153-155).

Both are streamed tensor by tensor into ``pg_load_tensor`` (the engine converts to its own layouts);
names the engine does not own (e.g. ``vision_model.*`` when the understanding encoder is disabled) are
reported as skipped, like ``strict=False`` does.
"""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Iterator, List, Optional, Tuple

import torch

from .engine import Engine, PlanGenError


def iter_safetensors(model_dir: str) -> Iterator[Tuple[str, torch.Tensor]]:
    from safetensors import safe_open
    index = os.path.join(model_dir, "model.safetensors.index.json")
    if os.path.exists(index):
        files = sorted(set(json.load(open(index))["weight_map"].values()))
    else:
        files = sorted(os.path.basename(p) for p in glob.glob(os.path.join(model_dir, "*.safetensors")))
    if not files:
        raise PlanGenError(f"no .safetensors files in {model_dir}")
    for fn in files:
        with safe_open(os.path.join(model_dir, fn), framework="pt", device="cpu") as f:
            for k in f.keys():
                yield k, f.get_tensor(k)


def latest_checkpoint(out_dir: str) -> Optional[str]:
    """``resume='latest'``: newest ``checkpoint-{step}`` directory (base_system.py:137-144)."""
    cands = [d for d in glob.glob(os.path.join(out_dir, "checkpoint-*")) if os.path.isdir(d)]
    if not cands:
        return None
    return max(cands, key=lambda d: int(d.rsplit("-", 1)[-1]))


def load_checkpoint(engine: Engine, janus_path: str, overlay: Optional[str] = None, strict: bool = True) -> Dict[str, List[str]]:
    """Load base weights then the PlanGen overlay (later tensors replace earlier ones); the engine's derived
    tables and decode layouts are built ONCE at the end (``Engine.finalize``)."""
    skipped: List[str] = []
    loaded = 0
    sd = {}
    for name, t in iter_safetensors(janus_path):
        sd[name] = t
        if len(sd) >= 64:                         # bounded host memory: hand over in groups, finalize once
            n, sk = engine.load_tensors(sd)
            loaded += n; skipped += sk; sd = {}
    n, sk = engine.load_tensors(sd)
    loaded += n; skipped += sk
    if overlay:
        path = overlay if overlay.endswith(".pth") else os.path.join(overlay, "trainable_model_parameters.pth")
        if not os.path.exists(path):
            raise FileNotFoundError(f"PlanGen overlay not found: {path}")
        ov = torch.load(path, map_location="cpu", weights_only=True)
        n, sk = engine.load_tensors(ov)           # "vl_gpt." prefix is stripped by pg_load_tensor
        if ov and n == 0:
            # e.g. a LoRA-only checkpoint (lora_A / lora_B keys): nothing of it would take effect
            raise PlanGenError(f"overlay {path}: none of its {len(ov)} tensors is a weight of this engine "
                               f"(first key {next(iter(ov))!r}); refusing to run the base weights silently")
        loaded += n; skipped += sk
    engine.finalize(strict)
    return {"loaded": [str(loaded)], "skipped": skipped}
