"""GPU: greedy text / layout-token decode (a11, f1) at THE REAL CONFIGURATION against tests/golden/generate_fullconfig.npz
(oracle/make_golden.py::golden_text_full_config: transformers ``LlamaForCausalLM.generate`` driven like plangen_base.py:513-523).

Janus-Pro-1B width, depth (24 layers) and vocabulary (102 400, untied lm_head behind 24 layers), 6 left-padded prompts of 24-96 tokens
(positions = mask cumsum), 24 greedy steps, at least one row stopping at EOS.  The 2-layer text fixtures (tests/test_gpu_fullvocab.py,
test_gpu_fullwidth.py) cannot show rounding accumulated over the real depth in front of a 102 400-way argmax.
PG_F32: ids bit-exact (stopped and un-stopped runs).  PG_BF16 (round 6): accepted relative to the REFERENCE'S OWN bf16 arithmetic
(tests/golden/generate_fullconfig_bf16ref.npz: LlamaForCausalLM under torch.autocast(bfloat16), fp32 master weights, plangen_base.py:95,360) -- logit-level
at every prompt position, id-level through the forced-oracle protocol.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu

_S = {}
N_NEW = 24


def _setup():
    if "W" not in _S:
        from plangen_amd.config import PlanGenConfig
        g = load_golden("generate_fullconfig.npz")
        ocfg = R.OracleCfg()
        W = R.make_weights(ocfg, seed=int(g["seed_w"]), with_lm_head=True)
        ws = float(sum(v.double().abs().sum() for v in W.values()))
        assert abs(ws - float(g["wsum"])) < 1e-6 * ws, "seeded weights drifted from the ones the fixture was generated with"
        _S.update(W=W, g=g, cfg=PlanGenConfig.janus_pro_1b(), ocfg=ocfg)
    return _S


def _x2t(dtype, eos, **kw):
    from plangen_amd.engine import Engine
    from plangen_amd.system import System
    s = _setup()
    g = s["g"]
    e = Engine(s["cfg"], dtype=dtype, max_rows=8, max_prompt=96, max_new=32, max_images=1, with_lm_head=True)
    e.load_state_dict(s["W"])
    try:
        sysm = System(s["cfg"], e)
        ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"].astype(np.int32))
        emb = sysm.vl_gpt.language_model.get_input_embeddings()(ids.to(e.device))
        return sysm.vl_gpt.language_model.generate(inputs_embeds=emb, attention_mask=mask.to(e.device), eos_token_id=eos,
                                                   max_new_tokens=N_NEW, **kw).cpu()
    finally:
        e.close()


def test_fullconfig_text_greedy_f32_matches_hf_generate():
    g = _setup()["g"]
    eos, unused = int(g["eos"]), int(g["unused_eos"])
    assert float(g["min_margin"]) > 1e-3          # no near tie in the fixture: fp32 on another summation order cannot legitimately flip an id
    out = _x2t("f32", eos)
    ref = g["out"].astype(np.int64)
    n = out.shape[1]                                  # generation may stop when every row is done; the fixture pads to N with EOS
    assert np.array_equal(out.numpy(), ref[:, :n]) and (ref[:, n:] == eos).all()
    assert 1 <= int((ref == eos).any(1).sum()) < ref.shape[0]
    probe = _x2t("f32", unused, min_new_tokens=N_NEW)      # un-stopped run
    assert np.array_equal(probe.numpy(), g["probe"].astype(np.int64))


def _text_ref():
    import json
    g = load_golden("generate_fullconfig_bf16ref.npz")
    return g, json.loads(str(g["stats"]))


def test_fullconfig_text_bf16_prompt_logits_vs_the_references_own_bf16():
    """Logit-level anchor of the text path in the production dtype (round 6): next-token logits at all 417 real prompt positions (24 layers,
    positions = mask cumsum, final norm, lm_head 2048 -> 102 400 through the decode GEMM kernels) against the fp32 reference, accepted relative to
    E_ref = |reference-bf16 - reference-fp32| of ``LlamaForCausalLM`` under torch.autocast(bfloat16) on the same prompts: E_hip <= K x E_ref per
    statistic and p99 |hip_bf16 - ref_bf16| <= p99 E_ref (tests/bf16ref.py).  No bound here was measured on this build."""
    import bf16ref
    from plangen_amd.engine import Engine
    s = _setup()
    e = Engine(s["cfg"], dtype="bf16", max_rows=8, max_prompt=96, max_new=32, max_images=1, with_lm_head=True)
    e.load_state_dict(s["W"])
    try:
        bf16ref.check_text_prompt_logits("generate_fullconfig", e, s["W"]["language_model.lm_head.weight"], s["g"], "bf16 text path, 24 layers x vocab 102 400")
    finally:
        e.close()


def test_fullconfig_text_greedy_bf16_vs_oracle_logits():
    """Id-level protocol (no logits cross the boundary of ``generate``): the engine's free-running bf16 ids are forced into the fp32 oracle; gap = fp32 best
    logit - fp32 logit of the engine's token.  A flip at margin m needs two logit errors that differ by m, so the worst gap is bounded by twice the
    reference-bf16's own worst logit error on this fixture (prompt_logits.all_columns.max of generate_fullconfig_bf16ref.npz); the reference-bf16's own
    ids, through the same protocol, are printed beside it (worst gap 0.031, agreement 0.903 -- a max over ~14 flipped steps, not a tolerance)."""
    import bf16ref
    _, E = _text_ref()
    TEXT_TOL = 2 * bf16ref.K_MAX * E["prompt_logits"]["all_columns"]["max"]
    s = _setup()
    g = s["g"]
    unused = int(g["unused_eos"])
    out = _x2t("bf16", unused, min_new_tokens=N_NEW)
    assert out.shape == (g["ids"].shape[0], N_NEW)
    W, ocfg = s["W"], s["ocfg"]
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"].astype(np.int32))
    _, logits = R.generate_text_greedy(W, ocfg, R.embed_tokens(W, ids), mask, N_NEW, unused, min_new_tokens=N_NEW, force_tokens=out, return_logits=True)
    lg = logits.permute(1, 0, 2)
    lg[:, :, unused] = float("-inf")
    gap = lg.max(-1).values - torch.gather(lg, 2, out[..., None]).squeeze(-1)
    agree = (gap == 0).float().mean().item()
    same_as_fp32 = float((out.numpy() == g["probe"].astype(np.int64)).mean())
    print(f"bf16 full-configuration text greedy: argmax agreement with the oracle on its own prefix {agree:.3f} (reference-bf16: {E['argmax_agreement_on_own_prefix']:.3f}), "
          f"worst logit gap {gap.max().item():.4f} (reference-bf16: {E['worst_logit_gap']:.4f}; bound 2 x E_ref logit max = {TEXT_TOL:.3f}), mean gap {gap.mean().item():.5f} "
          f"(reference-bf16: {E['gap_mean']:.5f}), free-running ids equal to the fp32 sequence {same_as_fp32:.3f} (reference-bf16: {E['free_running_ids_equal_fp32']:.3f})")
    assert gap.max().item() < TEXT_TOL, gap.max().item()
    assert agree >= E["argmax_agreement_on_own_prefix"] - 0.03          # 144 samples: one flip = 0.7 %
