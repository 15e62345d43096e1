"""GPU: greedy text decode (a11) at the REAL vocabulary against tests/golden/generate_fullvocab.npz
(oracle/make_golden.py::golden_text_full_vocab: transformers `LlamaForCausalLM.generate` driven like plangen_base.py:513-523).

Janus-Pro-1B width, 2 layers, vocab 102 400: the 2048 -> 102 400 `lm_head` skinny GEMM (419 MB of weights in bf16),
`text_scan_kernel` over 16 chunks of 6 400 columns, `text_argmax_kernel`, ids above 65 535 (EOS of the fixture = 67 852, stored
as int32), the 102 400-row embedding gather for the next step.  12 left-padded prompts of 24..96 tokens, 12 greedy steps, one row
stops at EOS and pads with it.  PG_F32: ids bit-exact.  PG_BF16: the engine's ids forced into the fp32 oracle, every token within
TEXT_TOL of the oracle's best logit (measured value printed).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from fullwidth_cfg import FULLV
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu

_S = {}
N_NEW = 12


def _setup():
    if "W" not in _S:
        from plangen_amd.config import PlanGenConfig
        g = load_golden("generate_fullvocab.npz")
        ocfg = R.OracleCfg(**FULLV)
        W = R.make_weights(ocfg, seed=11)
        ws = float(sum(v.double().abs().sum() for v in W.values()))
        assert abs(ws - float(g["wsum"])) < 1e-6 * ws, "seeded weights drifted from the ones the fixture was generated with"
        _S.update(W=W, g=g, cfg=PlanGenConfig(**FULLV), ocfg=ocfg)
    return _S


def _engine(dtype):
    from plangen_amd.engine import Engine
    s = _setup()
    if dtype not in _S:
        for k in [k for k in ("f32", "bf16") if k in _S]:
            _S.pop(k).close()
        e = Engine(s["cfg"], dtype=dtype, max_rows=16, max_prompt=96, max_new=16, max_images=1, with_lm_head=True)
        e.load_state_dict(s["W"])
        _S[dtype] = e
    return _S[dtype]


def _x2t(dtype, eos, **kw):
    from plangen_amd.system import System
    s = _setup()
    g = s["g"]
    e = _engine(dtype)
    sysm = System(s["cfg"], e)
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"].astype(np.int32))
    emb = sysm.vl_gpt.language_model.get_input_embeddings()(ids.to(e.device))
    return sysm.vl_gpt.language_model.generate(inputs_embeds=emb, attention_mask=mask.to(e.device), eos_token_id=eos,
                                               max_new_tokens=N_NEW, **kw).cpu()


def test_fullvocab_text_greedy_f32_matches_hf_generate():
    g = _setup()["g"]
    eos = int(g["eos"])
    assert eos >= 65536
    out = _x2t("f32", eos)
    ref = g["out"].astype(np.int64)
    assert out.shape[0] == ref.shape[0]
    n = out.shape[1]                                  # generation may stop when every row is done; the fixture pads to N
    assert np.array_equal(out.numpy(), ref[:, :n]) and (ref[:, n:] == eos).all()
    assert (ref == eos).any(1).sum() >= 1 and (ref >= 65536).sum() > 20
    # un-stopped run (the model's own EOS never appears): equals the probe ids, whose argmax columns hit all 16 scan chunks
    probe = _x2t("f32", _setup()["cfg"].eos_id, min_new_tokens=N_NEW)
    assert np.array_equal(probe.numpy(), g["probe"].astype(np.int64))
    assert len(set((g["probe"].reshape(-1) // 6400).tolist())) == 16


def test_fullvocab_text_greedy_bf16_vs_oracle_logits():
    import bf16ref
    TEXT_TOL, AGREE_MIN, E = bf16ref.text_id_bounds("generate_fullvocab")      # 2 x the reference-bf16's own worst logit error; its agreement - 0.03 (round 6)
    s = _setup()
    g = s["g"]
    out = _x2t("bf16", s["cfg"].eos_id, min_new_tokens=N_NEW)
    assert out.shape == (12, N_NEW)
    W, ocfg = s["W"], s["ocfg"]
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"].astype(np.int32))
    _, logits = R.generate_text_greedy(W, ocfg, R.embed_tokens(W, ids), mask, N_NEW, s["cfg"].eos_id, min_new_tokens=N_NEW,
                                       force_tokens=out, return_logits=True)
    lg = logits.permute(1, 0, 2)
    lg[:, :, s["cfg"].eos_id] = float("-inf")
    gap = lg.max(-1).values - torch.gather(lg, 2, out[..., None]).squeeze(-1)
    agree = (gap == 0).float().mean().item()
    print(f"bf16 full-vocabulary text greedy: argmax agreement {agree:.3f}, worst logit gap {gap.max().item():.4f}, "
          f"ids >= 65536: {(out >= 65536).sum().item()} of {out.numel()}")
    print(f"reference-bf16 through the same protocol: agreement {E['argmax_agreement_on_own_prefix']:.3f}, worst gap {E['worst_logit_gap']:.4f}; bound {TEXT_TOL:.3f}")
    assert gap.max().item() < TEXT_TOL, gap.max().item()
    assert agree >= AGREE_MIN, (agree, AGREE_MIN)
    assert (out >= 65536).any() and out.max().item() < 102400
    bf16ref.check_text_prompt_logits("generate_fullvocab", _engine("bf16"), s["W"]["language_model.lm_head.weight"], g, "bf16 text path, 2 layers x vocab 102 400")
