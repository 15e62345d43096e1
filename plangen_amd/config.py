"""Shapes of the path and the engine capacities."""
from __future__ import annotations

from dataclasses import asdict, dataclass, field
from typing import Tuple


@dataclass
class PlanGenConfig:
    # Janus-Pro-1B language model (tech report Table 1; HF config.json)
    hidden: int = 2048
    inter: int = 5632
    n_layers: int = 24
    n_heads: int = 16
    head_dim: int = 128
    vocab: int = 102400
    rms_eps: float = 1e-6
    rope_theta: float = 10000.0
    # generation head / VQ-16 tokenizer (modeling_vlm.py:36-51, vq_model.py:31-43)
    img_vocab: int = 16384
    img_dim: int = 8
    grid: int = 24
    gen_head_dim: int = 2048
    vq_ch: int = 128
    vq_ch_mult: Tuple[int, ...] = (1, 1, 2, 2, 4)
    vq_z: int = 256
    vq_res_blocks: int = 2
    # SigLIP-L/16-384 understanding encoder (siglip_vit.py:628-637) + aligner
    vit_width: int = 1024
    vit_layers: int = 24
    vit_heads: int = 16
    vit_mlp: int = 4096
    vit_patch: int = 16
    vit_img: int = 384
    eos_id: int = 100001           # conversation.py:306
    pad_id: int = 100002           # <｜▁pad▁｜>; numeric id lives in the HF tokenizer files
    # sampling defaults (cfg/base.py:158-162)
    cfg_weight: float = 5.0
    temperature: float = 1.0
    seed: int = 0

    @property
    def img_tokens(self) -> int:
        return self.grid * self.grid

    @property
    def vit_tokens(self) -> int:
        return (self.vit_img // self.vit_patch) ** 2

    @property
    def img_size(self) -> int:
        return self.grid * (2 ** (len(self.vq_ch_mult) - 1))

    def model_dict(self) -> dict:
        """Fields shared with the test oracle's config."""
        keys = ("hidden inter n_layers n_heads head_dim vocab rms_eps rope_theta img_vocab img_dim grid "
                "gen_head_dim vq_ch vq_ch_mult vq_z vq_res_blocks eos_id pad_id "
                "vit_width vit_layers vit_heads vit_mlp vit_patch vit_img").split()
        d = asdict(self)
        return {k: d[k] for k in keys}

    @staticmethod
    def janus_pro_1b() -> "PlanGenConfig":
        return PlanGenConfig()

    @staticmethod
    def tiny() -> "PlanGenConfig":
        """Small config with the same structure (tests, smoke): 2 layers, 2 heads x 128,
        8x8 image tokens, 3-level VQ decoder -> 32x32 images."""
        return PlanGenConfig(hidden=256, inter=512, n_layers=2, n_heads=2, head_dim=128, vocab=512,
                             img_vocab=256, img_dim=8, grid=8, gen_head_dim=256, vq_ch=64,
                             vq_ch_mult=(1, 2, 2), vq_z=64, eos_id=7, pad_id=3,
                             vit_width=128, vit_layers=2, vit_heads=2, vit_mlp=256, vit_patch=8, vit_img=64)
