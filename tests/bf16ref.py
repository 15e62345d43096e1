"""Reference-relative bf16 acceptance (round 6).

tests/golden/*_bf16ref.npz (oracle/make_golden_bf16ref.py) hold what the REFERENCE'S OWN production arithmetic -- the transformers-driven loop of
plangen_base.py:567-607 under ``torch.autocast(bfloat16)`` over fp32 master weights (plangen_base.py:95,360) -- does against its fp32 arithmetic on
the very inputs of the fp32 fixtures:  E_ref = |ref_bf16 - ref_fp32|.  The engine's production dtype is accepted when, statistic by statistic,

    E_hip = |hip_bf16 - ref_fp32|  <=  K * E_ref               (K = 1.05 on p50 / p99 / p99.9 / mean, K_MAX = 1.25 on maxima; the verdict allowed up to 1.25)
    p99 |hip_bf16 - ref_bf16|      <=  K_CROSS * p99 E_ref     (how far the engine's bf16 is from the reference's bf16, in units of that one's distance to its own fp32)

and its teacher-forced argmax agreement is within AGREE_SLACK of the reference-bf16's.  E_ref is recomputed HERE from the stored reference-bf16
logits on exactly the (images, steps) a test runs, so a test on a slice of a fixture is compared with the reference's error on that slice.
No constant in this file was measured on this build.
"""
import json

import numpy as np
import torch

from conftest import load_golden

K = 1.05                # distribution statistics (p50 / p99 / p99.9 / mean): the engine's bf16 must be NO WORSE than the reference's own bf16, to within the 5 % by which
                        # two implementations of the SAME arithmetic differ on these samples (text path: fp32 residual stream in both -> measured ratios 0.99-1.01)
K_MAX = 1.25            # maxima (an extreme-value statistic of 10^3-10^5 samples moves by tens of percent between two equally good roundings): the verdict's cap
K_CROSS = 1.25          # p99 |hip_bf16 - ref_bf16| over p99 E_ref.  Two different bf16 roundings of one fp32 computation are sqrt(E_hip^2 + E_ref^2 - 2 cov) apart:
                        # <= 1.0 is only reachable when E_hip << E_ref; measured on MI355X (round 6) 1.04-1.10 on the image loop at E_hip = 0.55-0.74 E_ref and
                        # 1.16-1.18 on the text path at E_hip = E_ref (independent roundings would give 1.17-1.25 and 1.41: the shared bf16 weight rounding correlates
                        # them).  Reported in every test's printed ratios and in DESIGN.md section 2 as a finding; bounded at the verdict's cap.
AGREE_SLACK = 0.02      # teacher-forced argmax agreement may sit this far below the reference-bf16's (bf16 logits tie often) -- or two samples' worth on
                        # the small fixtures (24 layers x 2 images x 16 steps = 32 samples: one flipped step is 0.031), see agree_slack()
STATS = ("max", "p999", "p99", "p50", "mean")


def load(name):
    g = load_golden(name + "_bf16ref.npz")
    return g, json.loads(str(g["stats"]))


def bf16_bits(a):
    """int16 bit patterns -> float32 tensor (the fixtures store reference-bf16 values exactly)."""
    return torch.from_numpy(np.ascontiguousarray(a).astype(np.int16)).view(torch.bfloat16).float()


def pct(x, q):
    return float(np.percentile(np.asarray(x, dtype=np.float64), q))


def err_stats(d):
    d = d.float().numpy().reshape(-1)
    return {"max": float(d.max()), "p999": pct(d, 99.9), "p99": pct(d, 99), "p50": pct(d, 50), "mean": float(d.mean())}


def agree_slack(n_samples):
    return max(AGREE_SLACK, 2.0 / max(int(n_samples), 1))


def limit(stat_name):
    return K_CROSS if stat_name.startswith("vs_ref_bf16") else (K_MAX if "max" in stat_name else K)


def over_limit(ratios):
    return {k: v for k, v in ratios.items() if v > limit(k)}


def hidden_bound(name, key="prefill_last"):
    """K x the reference-bf16's own max |hidden error| at the prompt's positions (LlamaModel under autocast vs fp32), for prefill checks."""
    _, E = load(name)
    return K_MAX * E["hidden"][key]["max"]


def check_image_loop(name, logits, toks, g32, what, images=None, steps=None):
    """logits [T, B, V] fp32 (teacher-forced on the fp32 fixture's tokens), toks [B, T] argmax; g32 = the fp32 fixture; images = slice of the
    fixture's images the run covers (None: all), steps = number of leading steps (None: all).  Returns the report dict."""
    gref, Eall = load(name)
    images = slice(None) if images is None else images
    T = int(g32["tokens"].shape[1]) if steps is None else int(steps)
    sel_all = torch.from_numpy(gref["sel_steps"]).long()
    keep = sel_all < T
    sel = sel_all[keep]
    vsel = torch.from_numpy(gref["vsel"]).long()
    assert abs(float(gref["wsum"]) - float(g32["wsum"])) < 1e-6 * float(g32["wsum"])
    if "sel_logits" in g32:
        assert np.array_equal(gref["vsel"], g32["vsel"])
        ref32 = torch.from_numpy(g32["sel_logits"])[keep][:, images]
        top_v = torch.from_numpy(g32["top_v"])[:T, images]; top_i = torch.from_numpy(g32["top_i"]).long()[:T, images]
    else:                                                                   # tiny fixture: whole logit rows
        full = torch.from_numpy(g32["logits"])[:T, images]
        ref32 = full[sel]; top_v, top_i = full.topk(4, dim=-1)
    refbf = bf16_bits(gref["ref_bf16_sel_logits"])[keep][:, images]
    gold = torch.from_numpy(g32["tokens"])[images, :T]
    ref_tok = torch.from_numpy(gref["ref_bf16_tf_tokens"])[images, :T]
    ref_top1 = torch.from_numpy(gref["ref_bf16_at_fp32_top1"])[:T, images]
    logits = logits.float().cpu(); toks = toks.cpu()
    mine = logits[sel][:, :, vsel]
    d_hip, d_ref, d_x = (mine - ref32).abs(), (refbf - ref32).abs(), (mine - refbf).abs()
    top_hip = (logits[:T].gather(2, top_i[..., :1]).squeeze(-1) - top_v[..., 0]).abs()
    top_ref = (ref_top1 - top_v[..., 0]).abs()
    agree = (toks[:, :T] == gold).t()
    agree_ref = (ref_tok == gold).t()
    margin = top_v[..., 0] - top_v[..., 1]
    H = {"all": err_stats(d_hip), "vs_ref_bf16": err_stats(d_x), "top1_value_err_max_all_steps": float(top_hip.max()), "teacher_forced_agreement": float(agree.float().mean())}
    E = {"all": err_stats(d_ref), "top1_value_err_max_all_steps": float(top_ref.max()), "teacher_forced_agreement": float(agree_ref.float().mean())}
    ratios = {k: H["all"][k] / E["all"][k] for k in STATS}
    ratios["top1_value_err_max"] = H["top1_value_err_max_all_steps"] / E["top1_value_err_max_all_steps"]
    if T > 400:
        late = sel >= 400
        H["late_steps_ge_400"], H["early_steps_lt_400"] = err_stats(d_hip[late]), err_stats(d_hip[~late])
        E["late_steps_ge_400"], E["early_steps_lt_400"] = err_stats(d_ref[late]), err_stats(d_ref[~late])
        for k in ("max", "p99", "p50"):
            ratios["late_" + k] = H["late_steps_ge_400"][k] / E["late_steps_ge_400"][k]
            ratios["early_" + k] = H["early_steps_lt_400"][k] / E["early_steps_lt_400"][k]
    ratios["vs_ref_bf16_p99_over_Eref_p99"] = H["vs_ref_bf16"]["p99"] / E["all"]["p99"]
    rep = {"what": what, "K": K, "E_hip": H, "E_ref": E, "ratio_E_hip_over_E_ref": {k: round(v, 3) for k, v in ratios.items()},
           "ref_bf16_free_running_agreement_with_fp32": Eall["free_running"]["agreement_with_fp32_tokens"]}
    print(f"{what}: reference-relative bf16:", json.dumps(rep))
    bad = over_limit(ratios)
    assert not bad, f"E_hip exceeds K x E_ref (K = {K} on quantiles, {K_MAX} on maxima, {K_CROSS} on the distance to the reference's bf16; a finding, not a tolerance to widen): {bad}"
    assert H["teacher_forced_agreement"] >= E["teacher_forced_agreement"] - agree_slack(agree.numel()), (H["teacher_forced_agreement"], E["teacher_forced_agreement"], agree.numel())
    # a flipped argmax is only legitimate inside twice the engine's OWN worst error (consistency of the two measurements, not a tolerance)
    err_bound = max(H["all"]["max"], H["top1_value_err_max_all_steps"])
    assert not ((~agree) & (margin > 2 * err_bound)).any(), rep
    return rep


def check_text_prompt_logits(name, e, lm_head_w, g32, what):
    """Logit-level anchor of the TEXT path (a11): next-token logits at every real prompt position -- engine ``e`` (bf16): packed prefill with
    positions = mask cumsum (position_mode 1), final norm, then lm_head through the decode GEMM kernels (pg_op_gemm, <= 128 rows per call) --
    against the fp32 reference, accepted relative to E_ref = |reference-bf16 - reference-fp32| of ``LlamaForCausalLM`` under torch.autocast(bfloat16)
    on the same prompts (tests/golden/<name>_bf16ref.npz, oracle/make_golden_bf16ref.py::text_anchor)."""
    gr, E = load(name)
    ids, mask = torch.from_numpy(g32["ids"].astype(np.int32)), torch.from_numpy(g32["mask"].astype(np.int32))
    L = ids.shape[1]
    real = mask[:, :L].bool()
    pad = [int(L - m.sum()) for m in mask[:, :L]]
    csel = torch.from_numpy(gr["csel"]).long()
    ref32 = torch.from_numpy(gr["prompt_sel_fp32"])                                   # [P, 128]
    refbf = bf16_bits(gr["prompt_sel_ref_bf16"])
    top1 = torch.from_numpy(gr["prompt_top1"])                                        # [P, 2]: fp32 top-1 value, index
    hid = e.prefill(ids, pad, position_mode=1, return_hidden=True).float().cpu()[real]
    assert hid.shape[0] == ref32.shape[0] == int(E["prompt_logits"]["positions"])
    lg = torch.cat([e.op_gemm(hid[i:i + 128], lm_head_w).cpu() for i in range(0, hid.shape[0], 128)])
    d_hip, d_ref, d_x = (lg[:, csel] - ref32).abs(), (refbf - ref32).abs(), (lg[:, csel] - refbf).abs()
    H, R_ = err_stats(d_hip), err_stats(d_ref)
    top_hip = float((lg.gather(1, top1[:, 1:].long()).squeeze(1) - top1[:, 0]).abs().max())
    agree = float((lg.argmax(-1) == top1[:, 1].long()).float().mean())
    ratios = {k: round(H[k] / R_[k], 3) for k in STATS}
    ratios["top1_value_err_max"] = round(top_hip / E["prompt_logits"]["top1_value_err_max"], 3)
    ratios["vs_ref_bf16_p99_over_Eref_p99"] = round(pct(d_x.numpy(), 99) / R_["p99"], 3)
    rep = {"what": what, "positions": int(hid.shape[0]), "E_hip": H, "E_ref": R_, "ratio_E_hip_over_E_ref": ratios, "argmax_agreement": agree,
           "ref_bf16_argmax_agreement": E["prompt_logits"]["argmax_agreement"]}
    print(f"{what}: reference-relative bf16, prompt-position logits:", json.dumps(rep))
    bad = over_limit(ratios)
    assert not bad, f"E_hip exceeds K x E_ref (K = {K} on quantiles, {K_MAX} on maxima, {K_CROSS} on the distance to the reference's bf16; a finding, not a tolerance to widen): {bad}"
    assert agree >= E["prompt_logits"]["argmax_agreement"] - AGREE_SLACK
    return rep


def text_id_bounds(name):
    """Bounds for the id-level protocol (the engine's free-running ids forced into the fp32 oracle; gap = fp32 best logit - fp32 logit of the engine's
    token): a flip at margin m needs two logit errors that differ by m, so the worst gap is bounded by TWICE the reference-bf16's own worst logit
    error on the fixture; agreement within 0.03 of the reference-bf16's (<= 1 280 samples, often 144: one flip = 0.7 %)."""
    _, E = load(name)
    return 2 * K_MAX * E["prompt_logits"]["all_columns"]["max"], E["argmax_agreement_on_own_prefix"] - 0.03, E


def check_vision(name, feat_err, aligned_err, what, policy="cuda_policy"):
    """SigLIP tower + aligner (a13): |hip_bf16 - ref_fp32| of the features / aligned embeddings at the fixture's tokens against the same statistic of
    the reference's own classes under torch.autocast(bfloat16) with bf16 pixels (modeling_vlm.py:249-250; tests/golden/<name>_bf16ref.npz)."""
    _, E = load(name)
    rep = {"what": what}
    ratios = {}
    for key, d in (("features_at_fixture_tokens", feat_err), ("aligned_at_fixture_tokens", aligned_err)):
        H = err_stats(d)
        rep[key] = {"E_hip": H, "E_ref": E[policy][key]}
        for k in STATS:
            ratios[key.split("_")[0] + "_" + k] = round(H[k] / E[policy][key][k], 3)
    rep["ratio_E_hip_over_E_ref"] = ratios
    print(f"{what}: reference-relative bf16:", json.dumps(rep))
    bad = over_limit(ratios)
    assert not bad, f"E_hip exceeds K x E_ref (K = {K} on quantiles, {K_MAX} on maxima, {K_CROSS} on the distance to the reference's bf16; a finding, not a tolerance to widen): {bad}"
    return rep
