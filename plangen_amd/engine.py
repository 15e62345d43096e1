"""Python owner of one ``pg_handle`` (one per process / GPU).

PyTorch-ROCm provides device tensors and the current HIP stream; every compute call goes
through the C ABI of include/plangen_hip.h.  No torch math on the product path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import PG_BF16, PG_F32
from .config import PlanGenConfig


class PlanGenError(RuntimeError):
    pass


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return PG_F32
    if t.dtype == torch.bfloat16:
        return PG_BF16
    raise TypeError(f"unsupported dtype {t.dtype} (need float32 or bfloat16)")


def _torch_dt(code: int):
    return torch.bfloat16 if code == PG_BF16 else torch.float32


class Engine:
    """MI355X engine for the layout->image path.

    dtype 'bf16' (production: bf16 weights / activations / KV, fp32 accumulate, fp32 residual
    stream) or 'f32' (parity mode, BASELINE config 1).
    """

    def __init__(self, cfg: PlanGenConfig, dtype: str = "bf16", max_rows: int = 16, max_prompt: int = 256,
                 max_new: Optional[int] = None, max_images: Optional[int] = None, with_lm_head: bool = False,
                 with_vq_encoder: bool = False, with_vision: bool = False, max_vision_images: Optional[int] = None,
                 device: int = 0, diag: bool = False):
        if not torch.cuda.is_available():
            raise PlanGenError("plangen_amd.Engine needs an MI355X (no CPU fallback)")
        # diag=True (tools/, bench.py's instrumented pass, hazard-screen tests): the handle lives in libplangen_diag.so -- the same object
        # code plus pg_diag_set_option; the product path never passes it
        self.diag = bool(diag)
        self.lib = _lib.load_diag() if diag else _lib.load()
        self.cfg = cfg
        self.dtype = dtype
        self.code = PG_BF16 if dtype == "bf16" else PG_F32
        self.tdtype = _torch_dt(self.code)
        self.device = torch.device("cuda", device)
        self.max_rows = max_rows
        self.max_new = max_new if max_new is not None else cfg.img_tokens
        self.max_images = max_images if max_images is not None else max(1, max_rows // 2)
        c = _lib.pg_config()
        for k in ("hidden", "inter", "n_layers", "n_heads", "head_dim", "vocab", "img_vocab", "img_dim", "grid",
                  "gen_head_dim", "vq_ch", "vq_z", "vq_res_blocks"):
            setattr(c, k, int(getattr(cfg, k)))
        c.vq_levels = len(cfg.vq_ch_mult)
        for i, m in enumerate(cfg.vq_ch_mult):
            c.vq_ch_mult[i] = int(m)
        c.rms_eps = cfg.rms_eps
        c.rope_theta = cfg.rope_theta
        c.compute_dtype = self.code
        c.max_rows, c.max_prompt, c.max_new, c.max_images = max_rows, max_prompt, self.max_new, self.max_images
        c.with_lm_head = int(with_lm_head)
        c.with_vq_encoder = int(with_vq_encoder)
        c.with_vision = int(with_vision)
        for k in ("vit_width", "vit_layers", "vit_heads", "vit_mlp", "vit_patch", "vit_img"):
            setattr(c, k, int(getattr(cfg, k)))
        c.max_vision_images = max_vision_images if max_vision_images is not None else self.max_images
        self._c = c
        h = C.c_void_p()
        rc = self.lib.pg_create(C.byref(h), C.byref(c), device)
        if rc != 0:
            raise PlanGenError(f"pg_create failed ({_lib.STATUS.get(rc, rc)}): {self.lib.pg_last_error(None).decode()}")
        self.h = h
        self.R = 0
        self.L = 0
        self._keep = []       # tensors the library may still be reading asynchronously

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc: int, what: str):
        if rc != 0:
            raise PlanGenError(f"{what} failed ({_lib.STATUS.get(rc, rc)}): {self.lib.pg_last_error(self.h).decode()}")

    @property
    def stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _p(t: Optional[torch.Tensor]) -> C.c_void_p:
        return C.c_void_p(0 if t is None else t.data_ptr())

    def _dev(self, t: torch.Tensor, dtype=None) -> torch.Tensor:
        t = t.to(self.device)
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        return t.contiguous()

    def close(self):
        if getattr(self, "h", None):
            self.lib.pg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_bytes(self) -> int:
        return int(self.lib.pg_device_bytes(self.h))

    def set_option(self, key: str, value: int):
        self._check(self.lib.pg_set_option(self.h, key.encode(), int(value)), "pg_set_option")

    def set_diag_option(self, key: str, value: int):
        """Measurement-only switches of libplangen_diag.so (skip_attn, attn_variant, ...): needs Engine(diag=True)."""
        if not self.diag:
            raise PlanGenError("set_diag_option needs an engine created with diag=True (libplangen_diag.so)")
        self._check(self.lib.pg_diag_set_option(self.h, key.encode(), int(value)), "pg_diag_set_option")

    def timing(self) -> dict:
        t = _lib.pg_timing()
        self._check(self.lib.pg_get_timing(self.h, C.byref(t)), "pg_get_timing")
        return {k: getattr(t, k) for k, _ in t._fields_}

    def class_timing(self) -> dict:
        """Per-kernel-class sums of the last instrumented decode loop (after ``timing()``)."""
        out = {}
        cls = 0
        while True:
            name, ms, n, by = C.c_char_p(), C.c_double(), C.c_int(), C.c_double()
            if self.lib.pg_get_class_timing(self.h, cls, C.byref(name), C.byref(ms), C.byref(n), C.byref(by)) != 0:
                break
            out[name.value.decode()] = {"ms_sum": ms.value, "launches": n.value, "bytes_sum": by.value}
            cls += 1
        return out

    # ------------------------------------------------------------------ weights
    def load_tensors(self, sd: Dict[str, torch.Tensor]) -> Tuple[int, list]:
        """Hand tensors to ``pg_load_tensor`` by their reference state_dict names (HF Janus-Pro-1B keys,
        optionally with PlanGen's ``vl_gpt.`` prefix) WITHOUT finalizing: callers streaming a checkpoint in
        groups call this per group and :meth:`finalize` once.  Returns (#loaded, names the engine does not own)."""
        loaded, skipped = 0, []
        for name, t in sd.items():
            t = t.detach()
            if t.dtype not in (torch.float32, torch.bfloat16):
                t = t.float()
            t = t.cpu().contiguous()
            shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
            rc = self.lib.pg_load_tensor(self.h, name.encode(), C.c_void_p(t.data_ptr()), _dt(t), shape, t.dim())
            if rc == -4:
                skipped.append(name)
                continue
            self._check(rc, f"pg_load_tensor({name})")
            loaded += 1
        return loaded, skipped

    def finalize(self, strict: bool = True) -> int:
        """``pg_finalize_weights``: derived tables + decode weight layouts, once per checkpoint.  strict: raise when a
        required tensor was never loaded; otherwise the engine is told to run with the missing ones reading as zeros
        (``load_state_dict(strict=False)`` semantics of base_system.py:153-155).  Returns the number missing."""
        if not strict:
            self.set_option("allow_partial_weights", 1)
        missing = C.c_int(0)
        self._check(self.lib.pg_finalize_weights(self.h, C.byref(missing), self.stream), "pg_finalize_weights")
        if strict and missing.value:
            raise PlanGenError(f"{missing.value} required tensors missing; first: {self.lib.pg_last_error(self.h).decode()}")
        return missing.value

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> Tuple[int, list]:
        """load_tensors + finalize.  Returns (#loaded, skipped names)."""
        loaded, skipped = self.load_tensors(sd)
        self.finalize(strict)
        return loaded, skipped

    def init_synthetic(self, seed: int = 0, std: float = 0.02):
        """Seeded random-init weights of the configured architecture (bench / smoke: there is
        no checkpoint offline).  N(0, std^2) linears, norm weights 1, L2-normalised codebook
        (SURVEY 8d).  Generated tensor by tensor on the GPU, handed over through the same
        pg_load_tensor path a real checkpoint uses."""
        import math
        cfg = self.cfg
        g = torch.Generator(device=self.device).manual_seed(seed)

        def load(name, t):
            t = t.float().cpu().contiguous()
            shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
            rc = self.lib.pg_load_tensor(self.h, name.encode(), C.c_void_p(t.data_ptr()), PG_F32, shape, t.dim())
            if rc != -4:
                self._check(rc, f"pg_load_tensor({name})")

        def nrm(*shape, s=std):
            return torch.randn(*shape, generator=g, device=self.device) * s

        H, I, HD = cfg.hidden, cfg.inter, cfg.n_heads * cfg.head_dim
        LM = "language_model.model."
        load(LM + "embed_tokens.weight", nrm(cfg.vocab, H))
        for i in range(cfg.n_layers):
            p = f"{LM}layers.{i}."
            for nme, shp in (("self_attn.q_proj", (HD, H)), ("self_attn.k_proj", (HD, H)), ("self_attn.v_proj", (HD, H)),
                             ("self_attn.o_proj", (H, HD)), ("mlp.gate_proj", (I, H)), ("mlp.up_proj", (I, H)),
                             ("mlp.down_proj", (H, I))):
                load(p + nme + ".weight", nrm(*shp))
            load(p + "input_layernorm.weight", torch.ones(H))
            load(p + "post_attention_layernorm.weight", torch.ones(H))
        load(LM + "norm.weight", torch.ones(H))
        load("language_model.lm_head.weight", nrm(cfg.vocab, H))
        G = cfg.gen_head_dim
        load("gen_head.output_mlp_projector.weight", nrm(G, H))
        load("gen_head.output_mlp_projector.bias", nrm(G))
        load("gen_head.vision_head.weight", nrm(cfg.img_vocab, G))
        load("gen_head.vision_head.bias", nrm(cfg.img_vocab))
        load("gen_embed.weight", nrm(cfg.img_vocab, cfg.img_dim, s=1.0))
        load("gen_aligner.layers.0.weight", nrm(H, cfg.img_dim, s=0.3))
        load("gen_aligner.layers.0.bias", nrm(H))
        load("gen_aligner.layers.2.weight", nrm(H, H))
        load("gen_aligner.layers.2.bias", nrm(H))
        V = "gen_vision_model."
        cb = (torch.rand(cfg.img_vocab, cfg.img_dim, generator=g, device=self.device) * 2 - 1) / cfg.img_vocab
        load(V + "quantize.embedding.weight", torch.nn.functional.normalize(cb, dim=-1))
        load(V + "post_quant_conv.weight", nrm(cfg.vq_z, cfg.img_dim, 1, 1, s=0.3))
        load(V + "post_quant_conv.bias", nrm(cfg.vq_z))

        def conv(name, cout, cin, k):
            load(name + ".weight", nrm(cout, cin, k, k, s=1.0 / math.sqrt(cin * k * k)))
            load(name + ".bias", nrm(cout, s=0.02))

        def norm(name, c):
            load(name + ".weight", torch.ones(c))
            load(name + ".bias", torch.zeros(c))

        def res(name, cin, cout):
            norm(name + ".norm1", cin); conv(name + ".conv1", cout, cin, 3)
            norm(name + ".norm2", cout); conv(name + ".conv2", cout, cout, 3)
            if cin != cout:
                conv(name + ".nin_shortcut", cout, cin, 1)

        def attn(name, c):
            norm(name + ".norm", c)
            for t in ("q", "k", "v", "proj_out"):
                conv(name + "." + t, c, c, 1)

        D = V + "decoder."
        nres = len(cfg.vq_ch_mult)
        bin_ = cfg.vq_ch * cfg.vq_ch_mult[-1]
        conv(D + "conv_in", bin_, cfg.vq_z, 3)
        res(D + "mid.0", bin_, bin_); attn(D + "mid.1", bin_); res(D + "mid.2", bin_, bin_)
        for bi, lvl in enumerate(reversed(range(nres))):
            bout = cfg.vq_ch * cfg.vq_ch_mult[lvl]
            for j in range(cfg.vq_res_blocks + 1):
                res(f"{D}conv_blocks.{bi}.res.{j}", bin_, bout)
                bin_ = bout
                if lvl == nres - 1:
                    attn(f"{D}conv_blocks.{bi}.attn.{j}", bin_)
            if lvl != 0:
                conv(f"{D}conv_blocks.{bi}.upsample.conv", bin_, bin_, 3)
        norm(D + "norm_out", bin_)
        conv(D + "conv_out", 3, bin_, 3)
        if self._c.with_vq_encoder:
            E = V + "encoder."
            conv(E + "conv_in", cfg.vq_ch, 3, 3)
            in_mult = (1,) + tuple(cfg.vq_ch_mult)
            b_in = cfg.vq_ch
            for lvl in range(nres):
                b_in = cfg.vq_ch * in_mult[lvl]
                b_out = cfg.vq_ch * cfg.vq_ch_mult[lvl]
                for j in range(cfg.vq_res_blocks):
                    res(f"{E}conv_blocks.{lvl}.res.{j}", b_in, b_out)
                    b_in = b_out
                    if lvl == nres - 1:
                        attn(f"{E}conv_blocks.{lvl}.attn.{j}", b_in)
                if lvl != nres - 1:
                    conv(f"{E}conv_blocks.{lvl}.downsample.conv", b_in, b_in, 3)
            res(E + "mid.0", b_in, b_in); attn(E + "mid.1", b_in); res(E + "mid.2", b_in, b_in)
            norm(E + "norm_out", b_in)
            conv(E + "conv_out", cfg.vq_z, b_in, 3)
            conv(V + "quant_conv", cfg.img_dim, cfg.vq_z, 1)
        if self._c.with_vision:
            Cw, Mh, ps = cfg.vit_width, cfg.vit_mlp, cfg.vit_patch
            VT = "vision_model.vision_tower."

            def lin(name, o, i):
                load(name + ".weight", nrm(o, i)); load(name + ".bias", nrm(o))

            load(VT + "patch_embed.proj.weight", nrm(Cw, 3, ps, ps, s=1.0 / math.sqrt(3 * ps * ps)))
            load(VT + "patch_embed.proj.bias", nrm(Cw))
            load(VT + "pos_embed", nrm(1, cfg.vit_tokens, Cw))
            for i in range(cfg.vit_layers):
                b = f"{VT}blocks.{i}."
                norm(b + "norm1", Cw); lin(b + "attn.qkv", 3 * Cw, Cw); lin(b + "attn.proj", Cw, Cw)
                norm(b + "norm2", Cw); lin(b + "mlp.fc1", Mh, Cw); lin(b + "mlp.fc2", Cw, Mh)
            norm(VT + "norm", Cw)
            lin("aligner.layers.0", H, Cw); lin("aligner.layers.2", H, H)
        missing = C.c_int(0)
        self._check(self.lib.pg_finalize_weights(self.h, C.byref(missing), self.stream), "pg_finalize_weights")
        if missing.value:
            raise PlanGenError(f"{missing.value} tensors missing after init_synthetic: {self.lib.pg_last_error(self.h).decode()}")

    # ------------------------------------------------------------------ language model
    @staticmethod
    def pad_len_from_mask(mask: torch.Tensor, L: int) -> list:
        """Leading-zero count of each row of a LEFT-padded attention mask [R, >=L]."""
        m = mask[:, :L].to("cpu", torch.int64)
        if not bool(((m[:, 1:] - m[:, :-1]) >= 0).all()):
            raise PlanGenError("attention_mask is not left-padded (0...01...1)")
        return (L - m.sum(-1)).tolist()

    @staticmethod
    def uncond_rows_shared(ids: torch.Tensor, pad_len: Sequence[int]) -> bool:
        """Host-side form of the library's probe: every odd (uncond CFG) row carries row 1's padding and ids."""
        R = ids.shape[0]
        if R < 4 or R % 2:
            return False
        if any(int(pad_len[r]) != int(pad_len[1]) for r in range(3, R, 2)):
            return False
        return bool((ids[3::2, int(pad_len[1]):] == ids[1, int(pad_len[1]):]).all())

    def prefill(self, ids: torch.Tensor, pad_len: Sequence[int], position_mode: int = 0,
                return_hidden: bool = False, hidden_dtype=torch.float32, uncond_shared: Optional[bool] = None) -> Optional[torch.Tensor]:
        """uncond_shared: None -> decided here when ``ids`` is still a HOST tensor (the collate's output; exact comparison, no device
        sync inside pg_prefill), probed on the device otherwise; True / False -> the caller's own host-side answer (bench.py:
        rank 0 compares the collated ids once and broadcasts the flag with them)."""
        if uncond_shared is None and not ids.is_cuda and position_mode == 0 and not return_hidden:
            uncond_shared = self.uncond_rows_shared(ids, pad_len)
        self.set_option("uncond_shared_hint", -1 if uncond_shared is None else int(bool(uncond_shared)))    # always sent: never a stale hint
        ids = self._dev(ids, torch.int32)
        R, L = ids.shape
        pl = (C.c_int32 * R)(*[int(v) for v in pad_len])
        out = torch.empty((R, L, self.cfg.hidden), dtype=hidden_dtype, device=self.device) if return_hidden else None
        self._check(self.lib.pg_prefill(self.h, self._p(ids), pl, R, L, position_mode, self._p(out),
                                        _dt(out) if out is not None else PG_F32, self.stream), "pg_prefill")
        self.R, self.L = R, L
        self._keep = [ids]
        return out

    def prefill_embeds(self, embeds: torch.Tensor, pad_len: Sequence[int], position_mode: int = 0,
                       return_hidden: bool = False, hidden_dtype=torch.float32) -> Optional[torch.Tensor]:
        embeds = self._dev(embeds)
        R, L, _ = embeds.shape
        pl = (C.c_int32 * R)(*[int(v) for v in pad_len])
        out = torch.empty((R, L, self.cfg.hidden), dtype=hidden_dtype, device=self.device) if return_hidden else None
        self._check(self.lib.pg_prefill_embeds(self.h, self._p(embeds), _dt(embeds), pl, R, L, position_mode,
                                               self._p(out), _dt(out) if out is not None else PG_F32, self.stream),
                    "pg_prefill_embeds")
        self.R, self.L = R, L
        self._keep = [embeds]
        return out

    def step(self, embeds: torch.Tensor, hidden_dtype=torch.float32) -> torch.Tensor:
        embeds = self._dev(embeds).view(self.R, self.cfg.hidden)
        out = torch.empty((self.R, self.cfg.hidden), dtype=hidden_dtype, device=self.device)
        self._check(self.lib.pg_step(self.h, self._p(embeds), _dt(embeds), self._p(out), _dt(out), self.stream), "pg_step")
        self._keep = [embeds]
        return out

    def gen_head(self, h: torch.Tensor) -> torch.Tensor:
        h = self._dev(h)
        R = h.shape[0]
        out = torch.empty((R, self.cfg.img_vocab), dtype=torch.float32, device=self.device)
        self._check(self.lib.pg_gen_head(self.h, self._p(h), _dt(h), self._p(out), R, self.stream), "pg_gen_head")
        self._keep = [h]
        return out

    def gen_embed(self, tok: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        tok = self._dev(tok.reshape(-1), torch.int32)
        out = torch.empty((tok.numel(), self.cfg.hidden), dtype=dtype, device=self.device)
        self._check(self.lib.pg_gen_embed(self.h, self._p(tok), self._p(out), _dt(out), tok.numel(), self.stream), "pg_gen_embed")
        self._keep = [tok]
        return out

    def embed_tokens(self, ids: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        shape = tuple(ids.shape)
        flat = self._dev(ids.reshape(-1), torch.int32)
        out = torch.empty((flat.numel(), self.cfg.hidden), dtype=dtype, device=self.device)
        self._check(self.lib.pg_embed_tokens(self.h, self._p(flat), self._p(out), _dt(out), flat.numel(), self.stream), "pg_embed_tokens")
        self._keep = [flat]
        return out.view(*shape, self.cfg.hidden)

    def decode_image_tokens(self, T: Optional[int] = None, cfg_weight: float = 5.0, temperature: float = 0.0,
                            seed: int = 0, force_tokens: Optional[torch.Tensor] = None,
                            force_mask: Optional[torch.Tensor] = None, return_logits: bool = False):
        T = self.cfg.img_tokens if T is None else T
        B = self.R // 2
        out = torch.zeros((B, T), dtype=torch.int32, device=self.device)
        ft = self._dev(force_tokens, torch.int32) if force_tokens is not None else None
        fm = self._dev(force_mask, torch.uint8) if force_mask is not None else None
        lg = torch.zeros((T, B, self.cfg.img_vocab), dtype=torch.float32, device=self.device) if return_logits else None
        self._check(self.lib.pg_decode_image_tokens(self.h, T, float(cfg_weight), float(temperature), int(seed),
                                                    self._p(ft), self._p(fm), self._p(out), self._p(lg), self.stream),
                    "pg_decode_image_tokens")
        self._keep = [ft, fm, out, lg]
        return (out, lg) if return_logits else out

    def generate_text_greedy(self, max_new_tokens: int, eos_id: int, min_new_tokens: int = 0) -> torch.Tensor:
        out = torch.full((self.R, max_new_tokens), eos_id, dtype=torch.int64, device=self.device)
        n = C.c_int(0)
        self._check(self.lib.pg_generate_text_greedy(self.h, max_new_tokens, min_new_tokens, eos_id, self._p(out),
                                                     C.byref(n), self.stream), "pg_generate_text_greedy")
        return out[:, :n.value]

    # ------------------------------------------------------------------ VQ
    def vq_decode(self, codes: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        codes = self._dev(codes, torch.int32)
        B = codes.shape[0]
        S = self.cfg.img_size
        out = torch.empty((B, 3, S, S), dtype=dtype, device=self.device)
        self._check(self.lib.pg_vq_decode(self.h, self._p(codes), self._p(out), _dt(out), B, self.stream), "pg_vq_decode")
        self._keep = [codes]
        return out

    def vq_encode(self, img: torch.Tensor) -> torch.Tensor:
        img = self._dev(img)
        B = img.shape[0]
        out = torch.empty((B * self.cfg.img_tokens,), dtype=torch.int64, device=self.device)
        self._check(self.lib.pg_vq_encode(self.h, self._p(img), _dt(img), self._p(out), B, self.stream), "pg_vq_encode")
        self._keep = [img]
        return out

    def vision_encode(self, images: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        """aligner(vision_model(images)): [B,3,S,S] -> [B, P, hidden] (modeling_vlm.py:243-250)."""
        images = self._dev(images)
        B = images.shape[0]
        out = torch.empty((B, self.cfg.vit_tokens, self.cfg.hidden), dtype=dtype, device=self.device)
        self._check(self.lib.pg_vision_encode(self.h, self._p(images), _dt(images), self._p(out), _dt(out), B, self.stream),
                    "pg_vision_encode")
        self._keep = [images]
        return out

    # ------------------------------------------------------------------ test taps
    def debug_read(self, name: str, index: int, numel: int, dtype) -> torch.Tensor:
        out = torch.zeros((numel,), dtype=dtype, device=self.device)       # the library copies min(numel, buffer size): the rest stays zero
        self._check(self.lib.pg_debug_read(self.h, name.encode(), index, self._p(out), out.numel() * out.element_size(),
                                           self.stream), "pg_debug_read")
        return out

    def op_rmsnorm(self, x: torch.Tensor, w: torch.Tensor, eps: float, partial: Optional[torch.Tensor] = None):
        x = self._dev(x, torch.float32).clone()
        M, H = x.shape
        w = self._dev(w, self.tdtype)
        S = 0 if partial is None else partial.shape[0]
        part = self._dev(partial, torch.float32) if partial is not None else None
        out = torch.empty((M, H), dtype=self.tdtype, device=self.device)
        self._check(self.lib.pg_op_rmsnorm(self.h, self._p(x), self._p(part), S, self._p(w), self._p(out), M, H, eps, self.stream), "pg_op_rmsnorm")
        torch.cuda.synchronize()
        return x, out

    def op_gemm(self, a: torch.Tensor, w: torch.Tensor, kind: int = 0) -> torch.Tensor:
        a = self._dev(a, self.tdtype)
        w = self._dev(w, self.tdtype)
        M, K = a.shape
        N = w.shape[0]
        out = torch.zeros((64, M, N), dtype=torch.float32, device=self.device)
        S = C.c_int(0)
        self._check(self.lib.pg_op_gemm(self.h, self._p(a), self._p(w), self._p(out), M, N, K, kind, C.byref(S), self.stream), "pg_op_gemm")
        torch.cuda.synchronize()
        return out[:S.value].sum(0)

    def op_gemm_splits(self, a: torch.Tensor, w: torch.Tensor) -> int:
        """Split-K slab count the decode GEMM picks for this shape under THIS handle's tuning."""
        a = self._dev(a, self.tdtype); w = self._dev(w, self.tdtype)
        M, K = a.shape
        N = w.shape[0]
        out = torch.zeros((64, M, N), dtype=torch.float32, device=self.device)
        S = C.c_int(0)
        self._check(self.lib.pg_op_gemm(self.h, self._p(a), self._p(w), self._p(out), M, N, K, 1, C.byref(S), self.stream), "pg_op_gemm")
        torch.cuda.synchronize()
        return S.value

    def op_swiglu_gemm(self, a: torch.Tensor, w_gate: torch.Tensor, w_up: torch.Tensor) -> torch.Tensor:
        """down-proj input of LlamaMLP: silu(a @ w_gate^T) * (a @ w_up^T) through the decode kernel (tiled weights,
        SwiGLU epilogue).  The gate/up rows are interleaved in blocks of 8 here, like pg_load_tensor does."""
        a = self._dev(a, torch.bfloat16)
        I, K = w_gate.shape
        wgu = torch.stack([w_gate.reshape(I // 8, 8, K), w_up.reshape(I // 8, 8, K)], dim=1).reshape(2 * I, K)
        wgu = self._dev(wgu, torch.bfloat16)
        M = a.shape[0]
        out = torch.empty((M, I), dtype=torch.bfloat16, device=self.device)
        self._check(self.lib.pg_op_swiglu_gemm(self.h, self._p(a), self._p(wgu), self._p(out), M, I, K, self.stream), "pg_op_swiglu_gemm")
        torch.cuda.synchronize()
        return out

    def op_uniform(self, bits: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(u, gumbel) the sampler derives from raw 64-bit RNG outputs (int64 tensor = the bit patterns)."""
        b = self._dev(bits.reshape(-1), torch.int64)
        n = b.numel()
        out = torch.empty((2 * n,), dtype=torch.float32, device=self.device)
        self._check(self.lib.pg_op_uniform(self.h, self._p(b), self._p(out), n, self.stream), "pg_op_uniform")
        torch.cuda.synchronize()
        return out[:n], out[n:]

    def op_conv3x3(self, x_nhwc: torch.Tensor, w_oihw: torch.Tensor, bias: torch.Tensor, residual=None, up: int = 0,
                   stride2: int = 0) -> torch.Tensor:
        x = self._dev(x_nhwc, self.tdtype)
        B, Hi, Wi, Cin = x.shape
        Cout = w_oihw.shape[0]
        w = self._dev(w_oihw.permute(0, 2, 3, 1).reshape(Cout, 9, Cin), self.tdtype)   # [Cout][tap][Cin]
        b = self._dev(bias, torch.float32)
        Ho, Wo = (Hi // 2, Wi // 2) if stride2 else (Hi << up, Wi << up)
        r = self._dev(residual, self.tdtype) if residual is not None else None
        out = torch.empty((B, Ho, Wo, Cout), dtype=self.tdtype, device=self.device)
        self._check(self.lib.pg_op_conv3x3(self.h, self._p(x), self._p(w), self._p(b), self._p(r), self._p(out), B, Hi, Wi,
                                           Cin, Cout, up, stride2, self.stream), "pg_op_conv3x3")
        torch.cuda.synchronize()
        return out

    def op_groupnorm(self, x_nhwc: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, swish: bool) -> torch.Tensor:
        x = self._dev(x_nhwc, torch.float32)          # the VQ skip stream is fp32 in both modes
        B, Hs, Ws, Cc = x.shape
        g = self._dev(gamma, torch.float32)
        b = self._dev(beta, torch.float32)
        out = torch.empty(x.shape, dtype=self.tdtype, device=self.device)
        self._check(self.lib.pg_op_groupnorm(self.h, self._p(x), self._p(g), self._p(b), self._p(out), B, Hs * Ws, Cc,
                                             int(swish), self.stream), "pg_op_groupnorm")
        torch.cuda.synchronize()
        return out
