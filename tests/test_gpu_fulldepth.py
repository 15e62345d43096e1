"""GPU: Janus-Pro-1B WIDTH AND DEPTH (24 layers x hidden 2048) against the transformers-driven fixture
(tests/golden/sample_image_fulldepth.npz, oracle/make_golden.py::golden_full_depth): 2 CFG pairs, L = 64, 16 greedy steps.
What the 2-layer full-width fixtures cannot show: rounding accumulated over 24 layers, and the per-layer strides of the KV cache and of
the weight tables at the real depth.  PG_F32: tokens bit-exact free-running, logits within 5e-3.  PG_BF16: teacher-forced, measured
error statistics printed; bounds stated at the asserts."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden
from fullwidth_cfg import FULLW
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu

_S = {}


def _setup():
    if not _S:
        from plangen_amd.config import PlanGenConfig
        g = load_golden("sample_image_fulldepth.npz")
        kw = dict(FULLW, n_layers=24)
        ocfg = R.OracleCfg(**kw)
        W = R.make_weights(ocfg, seed=6)
        ws = float(sum(v.double().abs().sum() for v in W.values()))
        assert abs(ws - float(g["wsum"])) < 1e-6 * ws
        _S.update(W=W, g=g, cfg=PlanGenConfig(**kw))
    return _S


def _run(dtype, force=None):
    from plangen_amd.engine import Engine
    s = _setup()
    g = s["g"]
    e = Engine(s["cfg"], dtype=dtype, max_rows=4, max_prompt=64, max_new=32, max_images=1)
    e.load_state_dict(s["W"])
    try:
        ids = torch.from_numpy(g["ids"].astype(np.int32))
        pad = [int(p) for p in g["pad"]]
        e.prefill(ids, pad, position_mode=0)
        toks, logits = e.decode_image_tokens(T=g["tokens"].shape[1], cfg_weight=5.0, temperature=0.0, force_tokens=force, return_logits=True)
        return toks.cpu(), logits.cpu()
    finally:
        e.close()


def test_fulldepth_f32_tokens_bit_exact_and_logits():
    g = _setup()["g"]
    toks, logits = _run("f32")
    assert np.array_equal(toks.numpy(), g["tokens"])
    vsel = torch.from_numpy(g["vsel"]).long()
    err = (logits[:, :, vsel] - torch.from_numpy(g["sel_logits"])).abs().max().item()
    tv, ti = logits.topk(4, dim=-1)
    assert err < 5e-3 and (tv - torch.from_numpy(g["top_v"])).abs().max().item() < 5e-3, err
    print(f"24-layer f32: max |logit err| {err:.2e}")


def test_fulldepth_bf16_teacher_forced():
    g = _setup()["g"]
    gold = torch.from_numpy(g["tokens"])
    toks, logits = _run("bf16", force=gold.contiguous())
    vsel = torch.from_numpy(g["vsel"]).long()
    d = (logits[:, :, vsel] - torch.from_numpy(g["sel_logits"])).abs()
    top_v = torch.from_numpy(g["top_v"])
    margin = top_v[..., 0] - top_v[..., 1]
    got = toks.t()
    agree = got == gold.t()
    stats = {"logit_abs_err_max": float(d.max()), "p99": float(np.percentile(d.numpy(), 99)), "p50": float(np.percentile(d.numpy(), 50)),
             "logit_std": float(torch.from_numpy(g["sel_logits"]).std()), "agreement": float(agree.float().mean()), "margin_median": float(margin.median())}
    print("24-layer bf16 teacher-forced:", json.dumps(stats))
    # 24 layers of bf16 GEMM inputs under CFG weight 5 (logit std 2.7): bounds = 1.5x the values measured on MI355X (like the other full-size files)
    assert stats["logit_abs_err_max"] < BF16_MAX and stats["p99"] < BF16_P99, stats
    flips = (~agree) & (margin > 2 * stats["logit_abs_err_max"])
    assert not flips.any()


BF16_MAX, BF16_P99 = 0.40, 0.27       # 1.5x measured on MI355X (rounds 3-4): max 0.266, p99 0.177, p50 0.043 at a logit std of 2.75; agreement 93.8 %
