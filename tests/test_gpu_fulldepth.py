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
    """PG_BF16 accepted RELATIVE TO THE REFERENCE'S OWN bf16 arithmetic (round 6): tests/golden/sample_image_fulldepth_bf16ref.npz holds
    E_ref = |reference-bf16 - reference-fp32| of this very loop under torch.autocast(bfloat16) with fp32 master weights (plangen_base.py:95,360;
    oracle/make_golden_bf16ref.py: max 0.342 / p99 0.238 / p50 0.060, agreement 0.9375); tests/bf16ref.py asserts E_hip <= K x E_ref per statistic."""
    import bf16ref
    g = _setup()["g"]
    gold = torch.from_numpy(g["tokens"])
    toks, logits = _run("bf16", force=gold.contiguous())
    bf16ref.check_image_loop("sample_image_fulldepth", logits, toks, g, "24-layer bf16 teacher-forced")
