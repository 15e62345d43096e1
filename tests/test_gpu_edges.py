"""GPU: edge cases of the path through the C ABI -- minimal and ragged batches, capacity limits,
call-order errors, per-sample negative prompts (no sharing), temperature extremes."""
import os

import numpy as np
import pytest
import torch

from conftest import get_engine
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


def _collate(cfg, cond, neg):
    from plangen_amd.system import t2i_infer_collate_batch
    return t2i_infer_collate_batch(cond, neg, cfg.pad_id, cfg.img_tokens)


def _pad(mask, L):
    return (L - mask[:, :L].sum(-1)).tolist()


def test_single_image_single_token_prompts(tiny_cfg, tiny_weights, ocfg):
    """Smallest batch (one CFG pair) with 1-token prompts: KV slot 0 only, no padding at all."""
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    ids, mask = _collate(tiny_cfg, [[17]], [23])
    ref = R.sample_image(tiny_weights, ocfg, R.embed_tokens(tiny_weights, ids), mask, 5.0, n_tokens=6)
    e.prefill(ids, _pad(mask, 1))
    got = e.decode_image_tokens(T=6, cfg_weight=5.0, temperature=0.0).cpu()
    assert torch.equal(got, ref)


def test_ragged_prompts_and_per_sample_negatives(tiny_cfg, tiny_weights, ocfg):
    """Very uneven prompt lengths and one negative prompt per sample (use_neg_box branch,
    plangen_base.py:652-670): the shared-uncond fast path must NOT trigger."""
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    g = torch.Generator().manual_seed(61)
    cond = [torch.randint(8, tiny_cfg.vocab, (n,), generator=g).tolist() for n in (1, 30, 2, 17)]
    negs = [torch.randint(8, tiny_cfg.vocab, (n,), generator=g).tolist() for n in (4, 1, 9, 3)]
    ids, mask = _collate(tiny_cfg, cond, negs)
    ref = R.sample_image(tiny_weights, ocfg, R.embed_tokens(tiny_weights, ids), mask, 3.0, n_tokens=8)
    e.prefill(ids, _pad(mask, ids.shape[1]))
    got = e.decode_image_tokens(T=8, cfg_weight=3.0, temperature=0.0).cpu()
    assert torch.equal(got, ref)


def test_cfg_weight_zero_and_one(tiny_cfg, tiny_weights, ocfg):
    """w=1 -> pure conditional logits, w=0 -> pure unconditional logits (plangen_base.py:587)."""
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    g = torch.Generator().manual_seed(62)
    cond = [torch.randint(8, tiny_cfg.vocab, (n,), generator=g).tolist() for n in (6, 4)]
    neg = torch.randint(8, tiny_cfg.vocab, (5,), generator=g).tolist()
    ids, mask = _collate(tiny_cfg, cond, neg)
    for w in (0.0, 1.0):
        ref = R.sample_image(tiny_weights, ocfg, R.embed_tokens(tiny_weights, ids), mask, w, n_tokens=5)
        e.prefill(ids, _pad(mask, ids.shape[1]))
        assert torch.equal(e.decode_image_tokens(T=5, cfg_weight=w, temperature=0.0).cpu(), ref)


def test_full_length_decode_fills_kv_capacity_exactly(tiny_cfg, tiny_weights, ocfg):
    """max_prompt-long prompt + all img_tokens steps: the last step writes the last KV slot."""
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    g = torch.Generator().manual_seed(63)
    cond = [torch.randint(8, tiny_cfg.vocab, (96,), generator=g).tolist()]
    neg = torch.randint(8, tiny_cfg.vocab, (96,), generator=g).tolist()
    ids, mask = _collate(tiny_cfg, cond, neg)
    ref = R.sample_image(tiny_weights, ocfg, R.embed_tokens(tiny_weights, ids), mask, 5.0)
    e.prefill(ids, _pad(mask, 96))
    assert torch.equal(e.decode_image_tokens(cfg_weight=5.0, temperature=0.0).cpu(), ref)


def test_capacity_and_order_errors(tiny_cfg, tiny_weights):
    from plangen_amd.engine import Engine, PlanGenError
    e = Engine(tiny_cfg, dtype="f32", max_rows=4, max_prompt=8, max_new=4, max_images=1)
    with pytest.raises(PlanGenError, match="PG_ERR_STATE"):
        e.prefill(torch.zeros((2, 4), dtype=torch.int32), [0, 0])            # weights not finalised
    e.load_state_dict({k: v for k, v in tiny_weights.items()
                       if not k.startswith(("vision_model.", "aligner.", "gen_vision_model.encoder", "gen_vision_model.quant_conv"))}
                      | {}, strict=False)
    ids = torch.randint(8, tiny_cfg.vocab, (2, 6)).int()
    with pytest.raises(PlanGenError, match="PG_ERR_STATE"):
        e.R = 2
        e.decode_image_tokens(T=2)                                            # decode before prefill
    with pytest.raises(PlanGenError, match="PG_ERR_CAPACITY"):
        e.prefill(torch.zeros((6, 4), dtype=torch.int32), [0] * 6)           # rows > max_rows
    with pytest.raises(PlanGenError, match="PG_ERR_CAPACITY"):
        e.prefill(torch.zeros((2, 12), dtype=torch.int32), [0, 0])           # prompt > max_prompt
    with pytest.raises(PlanGenError, match="PG_ERR_ARG"):
        e.prefill(ids, [6, 0])                                               # a row with no real token
    e.prefill(ids, [0, 2])
    with pytest.raises(PlanGenError, match="PG_ERR_CAPACITY"):
        e.decode_image_tokens(T=7)                                            # T-1 > max_new
    toks = e.decode_image_tokens(T=5, temperature=0.0)
    assert toks.shape == (1, 5)
    with pytest.raises(PlanGenError, match="PG_ERR_STATE"):
        e.decode_image_tokens(T=2)                                            # needs a fresh prefill
    with pytest.raises(PlanGenError):
        e.vq_decode(torch.zeros((2, tiny_cfg.img_tokens), dtype=torch.int32))  # images > max_images
    # out-of-range ids / codes are clamped, never read out of bounds
    e.prefill(torch.full((2, 6), 10 ** 6, dtype=torch.int32), [0, 0])
    img = e.vq_decode(torch.full((1, tiny_cfg.img_tokens), 10 ** 6, dtype=torch.int32))
    assert torch.isfinite(img).all()
    e.close()


def test_temperature_limits(tiny_cfg, tiny_weights):
    """Very low temperature sampling collapses onto the greedy tokens; sampling never emits an
    out-of-range id."""
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    g = torch.Generator().manual_seed(64)
    cond = [torch.randint(8, tiny_cfg.vocab, (n,), generator=g).tolist() for n in (5, 7)]
    neg = torch.randint(8, tiny_cfg.vocab, (4,), generator=g).tolist()
    ids, mask = _collate(tiny_cfg, cond, neg)
    pad = _pad(mask, ids.shape[1])
    e.prefill(ids, pad)
    greedy = e.decode_image_tokens(T=6, cfg_weight=5.0, temperature=0.0).cpu()
    e.prefill(ids, pad)
    cold = e.decode_image_tokens(T=6, cfg_weight=5.0, temperature=1e-3, seed=5).cpu()
    assert torch.equal(greedy[:, 0], cold[:, 0])            # first token: same logits, T -> 0
    e.prefill(ids, pad)
    hot = e.decode_image_tokens(T=6, cfg_weight=5.0, temperature=50.0, seed=5).cpu()
    assert (hot >= 0).all() and (hot < tiny_cfg.img_vocab).all()


def test_cli_validation_writes_images(tmp_path):
    """python train.py --cfg ... --opt test=True ... -> System.validation (tiny config, synthetic prompts)."""
    import train
    out = train.parse_args(["--cfg", "project/plangen/cfg/uni/h_text_ump+oimsam.py", "--opt", "test=True", "tiny=True",
                            "resume=None", f"out_path='{tmp_path}'", "test_batch_size=2", "max_test_len=2", "max_prompt=64",
                            "janus_path=None"])
    import importlib
    System = importlib.import_module(out.system_cls_path).System
    m = System(out, None)
    m.setup_data(None)
    m.resume(None)
    res = m.validation(0)
    assert res["images"] == 4
    files = os.listdir(res["out_dir"])
    assert sum(f.endswith((".png", ".pt")) for f in files) == 4 and sum(f.endswith("_tokens.json") for f in files) == 2
