"""Drop-in facade with the attribute / call surface PlanGen's ``System`` uses on
``self.vl_gpt`` (Janus ``MultiModalityCausalLM``, three_party/Janus/janus/models/
modeling_vlm.py:190-271) -- SURVEY.md section 8b.  Every call lands in the C ABI.

    vl_gpt.language_model.get_input_embeddings()(ids)             plangen_base.py:371,548
    vl_gpt.language_model.model(inputs_embeds=, attention_mask=,  plangen_base.py:571-577
                                use_cache=True, past_key_values=)
    vl_gpt.gen_head(h)                                            plangen_base.py:579
    vl_gpt.prepare_gen_img_embeds(tok)                            plangen_base.py:603
    vl_gpt.gen_vision_model.decode_code(codes, shape=, ...)       plangen_base.py:555
    vl_gpt.gen_vision_model.encode(x)                             plangen_base.py:532
    vl_gpt.language_model.generate(inputs_embeds=, ...)           plangen_base.py:513-523
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from .engine import Engine, PlanGenError


@dataclass
class PastKeyValues:
    """Opaque token standing for the engine-resident KV cache (the reference passes HF's
    DynamicCache back verbatim; here the cache never leaves HBM)."""
    engine: Engine
    epoch: int
    seen: int            # cached positions (padded layout), like DynamicCache.get_seq_length()

    def get_seq_length(self) -> int:
        return self.seen


@dataclass
class ModelOutput:
    last_hidden_state: torch.Tensor
    past_key_values: PastKeyValues


class _Embedding:
    def __init__(self, eng: Engine):
        self.eng = eng

    def __call__(self, ids: torch.Tensor) -> torch.Tensor:
        return self.eng.embed_tokens(ids)


class _LlamaModel:
    """``language_model.model``: prefill on the first call, one decode step on later calls."""

    def __init__(self, eng: Engine, position_mode: int = 0):
        self.eng = eng
        self.epoch = 0
        self.position_mode = position_mode

    def __call__(self, inputs_embeds: torch.Tensor = None, attention_mask: Optional[torch.Tensor] = None,
                 use_cache: bool = True, past_key_values: Optional[PastKeyValues] = None, **kw) -> ModelOutput:
        eng = self.eng
        if inputs_embeds is None:
            raise PlanGenError("language_model.model: inputs_embeds is required (the reference never passes input_ids)")
        R, q, _ = inputs_embeds.shape
        out_dtype = inputs_embeds.dtype if inputs_embeds.dtype in (torch.float32, torch.bfloat16) else torch.float32
        if past_key_values is None:
            if attention_mask is None:
                pad = [0] * R
            else:
                # mask is [R, L+576] at every step (App. B-3): only its first q columns describe the prompt
                pad = Engine.pad_len_from_mask(attention_mask, q)
            hid = eng.prefill_embeds(inputs_embeds, pad, position_mode=self.position_mode, return_hidden=True,
                                     hidden_dtype=out_dtype)
            self.epoch += 1
            return ModelOutput(hid, PastKeyValues(eng, self.epoch, q))
        if past_key_values.epoch != self.epoch or past_key_values.engine is not eng:
            raise PlanGenError("stale past_key_values: the engine holds one KV cache per handle")
        if q != 1:
            raise PlanGenError("decode calls take one new position per row")
        hid = eng.step(inputs_embeds[:, 0, :], hidden_dtype=out_dtype)
        past_key_values.seen += 1
        return ModelOutput(hid[:, None, :], past_key_values)


class _LanguageModel:
    def __init__(self, eng: Engine):
        self.eng = eng
        self.model = _LlamaModel(eng, position_mode=0)
        self._emb = _Embedding(eng)

    def get_input_embeddings(self):
        return self._emb

    def generate(self, inputs_embeds: torch.Tensor = None, attention_mask: Optional[torch.Tensor] = None,
                 pad_token_id: Optional[int] = None, bos_token_id: Optional[int] = None,
                 eos_token_id: Optional[int] = None, max_new_tokens: int = 512, do_sample: bool = False,
                 use_cache: bool = True, min_new_tokens: int = 0, **kw) -> torch.Tensor:
        """HF GenerationMixin.generate, greedy (System.x2t, plangen_base.py:513-523): returns the
        NEW tokens only, int64 [B, n<=max_new_tokens], finished rows padded with eos."""
        if do_sample:
            raise PlanGenError("only greedy decoding (do_sample=False) is what the reference uses")
        if eos_token_id is None:
            raise PlanGenError("eos_token_id is required")
        R, L, _ = inputs_embeds.shape
        pad = [0] * R if attention_mask is None else Engine.pad_len_from_mask(attention_mask, L)
        self.eng.prefill_embeds(inputs_embeds, pad, position_mode=1)
        self.model.epoch += 1          # invalidates any sample_image cache token
        return self.eng.generate_text_greedy(max_new_tokens, int(eos_token_id), min_new_tokens)


class _GenVisionModel:
    def __init__(self, eng: Engine):
        self.eng = eng

    def decode_code(self, code_b: torch.Tensor, shape=None, channel_first: bool = True) -> torch.Tensor:
        """VQModel.decode_code (vq_model.py:505-508): int [B, g*g] -> [B,3,S,S]."""
        cfg = self.eng.cfg
        if shape is not None and (shape[1] != cfg.img_dim or shape[2] != cfg.grid or shape[3] != cfg.grid):
            raise PlanGenError(f"decode_code: shape {list(shape)} does not match the configured VQ ({cfg.img_dim},{cfg.grid},{cfg.grid})")
        if not channel_first:
            # vq_model.py:297-298: channel_first=False makes get_codebook_entry return z_q.view(shape) with shape = (B, H, W, C) -- an NHWC tensor that
            # the reference's own decode() then feeds to post_quant_conv (a Conv2d over dim 1): with H = W = 24 != 8 channels it raises there.  The only call
            # in the reference passes channel_first=True (plangen_base.py:555); refusing the other form is the reference's error behaviour, stated up front.
            raise PlanGenError("decode_code: only channel_first=True (the reference's call) is supported")
        B = shape[0] if shape is not None else code_b.shape[0]
        return self.eng.vq_decode(code_b.reshape(B, -1))

    def encode(self, x: torch.Tensor):
        """VQModel.encode (vq_model.py:494-498) -> (quant, losses, (None, None, indices));
        only the indices are produced (all the reference reads: ``encode(x)[-1][-1]``)."""
        idx = self.eng.vq_encode(x)
        return None, (None, None, None), (None, None, idx)


class MultiModalityCausalLM:
    """What ``self.vl_gpt`` is in PlanGen's System, backed by one MI355X engine."""

    def __init__(self, engine: Engine):
        self.engine = engine
        self.language_model = _LanguageModel(engine)
        self.gen_vision_model = _GenVisionModel(engine)
        self.training = False

    def gen_head(self, h: torch.Tensor) -> torch.Tensor:
        return self.engine.gen_head(h)

    def prepare_gen_img_embeds(self, image_ids: torch.Tensor) -> torch.Tensor:
        return self.engine.gen_embed(image_ids)

    def prepare_inputs_embeds(self, input_ids: torch.Tensor, pixel_values: torch.Tensor,
                              images_seq_mask: torch.Tensor, images_emb_mask: torch.Tensor, **kwargs) -> torch.Tensor:
        """MultiModalityCausalLM.prepare_inputs_embeds (modeling_vlm.py:221-268): SigLIP + aligner on
        the device, then the masked scatter of the image embeddings into the text embeddings."""
        bs, n = pixel_values.shape[0:2]
        images = pixel_values.reshape(bs * n, *pixel_values.shape[2:])
        images_embeds = self.engine.vision_encode(images)                   # [b*n, T2, D]
        images_embeds = images_embeds.reshape(bs, n * images_embeds.shape[1], -1)
        images_emb_mask = images_emb_mask.reshape(bs, -1).to(self.engine.device).bool()
        ids = input_ids.to(self.engine.device).clone()
        ids[ids < 0] = 0
        inputs_embeds = self.language_model.get_input_embeddings()(ids)
        inputs_embeds[images_seq_mask.to(self.engine.device).bool()] = images_embeds[images_emb_mask].to(inputs_embeds.dtype)
        return inputs_embeds

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise PlanGenError("training is out of scope of the MI355X generation path (SURVEY 3.4)")
        return self

    def parameters(self):
        # device / dtype discovery only (plangen_base.py:1183-1189)
        yield torch.empty(0, dtype=self.engine.tdtype, device=self.engine.device)
