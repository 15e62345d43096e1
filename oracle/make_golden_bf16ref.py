"""Reference-bf16 anchors: what the REFERENCE'S OWN production arithmetic does against its fp32 arithmetic.  TEST INFRASTRUCTURE ONLY.

The reference runs the whole path under ``torch.autocast(device_type='cuda', dtype=torch.bfloat16)`` (plangen_base.py:360) over fp32 master
weights (``from_pretrained`` without ``torch_dtype``, plangen_base.py:95).  Every other fixture in tests/golden/ is pure fp32, so until round 6
the production dtype of this build (PG_BF16) was only bounded by "1.5x what this build measured on itself".  This script runs THE SAME
transformers-driven loops that made the fp32 fixtures (plangen_base.py:567-607 for image tokens, :513-523 for text, vq_model.py:505-508 for
pixels) under ``torch.autocast('cpu', dtype=torch.bfloat16)`` with fp32 master weights and stores

  * E_ref = |reference-bf16 - reference-fp32| statistics (p50 / p99 / max, early vs late steps, teacher-forced argmax agreement, the
    free-running bf16 sequence's agreement with the fp32 tokens), and
  * the reference-bf16 logits at the fp32 fixture's (step, column) selection, so the GPU tests can also measure |hip_bf16 - ref_bf16|.

GPU tests then assert  E_hip = |hip_bf16 - ref_fp32| <= K * E_ref  per statistic (K stated in the test, <= 1.25).

What "autocast on CPU" shares with "autocast on CUDA" and what it does not (torch 2.10 cast policies, aten/src/ATen/autocast_mode.cpp):
  * identical: linear / matmul / conv2d / SDPA run in bf16 with fp32 accumulation; the HF LlamaRMSNorm casts to fp32 by itself; embedding
    lookups stay fp32; type promotion (fp32 + bf16 -> fp32) is device independent.  Consequence visible in the loop: the PREFILL residual stream
    is fp32 (embedding output) but every DECODE step's residual stream is bf16, because ``prepare_gen_img_embeds`` ends in an autocast Linear
    (modeling_vlm.py:270-271) and bf16 + bf16 stays bf16; gen_head's logits and the CFG mix (plangen_base.py:585) are bf16 too.
  * differs: CUDA's policy forces group_norm / layer_norm / softmax to fp32, the CPU policy leaves them in the input dtype.  Llama is not
    affected (SDPA hides its softmax on both).  The VQ decoder is (GroupNorm + swish in front of every conv): both variants are stored --
    ``cpu_policy`` (autocast as is) and ``cuda_policy`` (group_norm wrapped to compute and return fp32, softmax likewise, as CUDA autocast does).

Needs /root/reference and the fp32 fixtures; run as  python -m oracle.make_golden_bf16ref [name ...].
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from oracle import ref_cpu as R
from oracle import make_golden as MG

OUT = MG.OUT
BF = dict(device_type="cpu", dtype=torch.bfloat16)


def _pct(x, q):
    return float(np.percentile(np.asarray(x, dtype=np.float64), q))


def err_stats(d):
    d = d.float().numpy().reshape(-1)
    return {"max": float(d.max()), "p999": _pct(d, 99.9), "p99": _pct(d, 99), "p50": _pct(d, 50), "mean": float(d.mean())}


@torch.no_grad()
def autocast_sample_image(model, W, ids, mask, T, force=None, cfg_weight=5.0, log_every=64, what=""):
    """plangen_base.py:567-607 (greedy parity mode: argmax instead of multinomial) under the reference's autocast, fp32 master weights.
    force [B, T] int: teacher forcing -- the token fed back at step i is force[:, i], the argmax is still recorded.
    Returns (argmax tokens [B, T] int32, CFG-mixed logits [T, B, V] float32 -- exactly the bf16 values the reference would hold)."""
    B = ids.shape[0] // 2
    toks = torch.zeros((B, T), dtype=torch.int32)
    out_logits, outputs = [], None
    t0 = time.time()
    with torch.autocast(**BF):
        inputs_embeds = model.get_input_embeddings()(ids.long())                       # fp32 (embedding is not an autocast op)
        for i in range(T):
            outputs = model(inputs_embeds=inputs_embeds, attention_mask=mask, use_cache=True, past_key_values=outputs.past_key_values if i != 0 else None)
            if i == 0:
                autocast_sample_image.prefill_hidden = outputs.last_hidden_state.float().clone()      # [R, L, H]: the prompt's hidden states under autocast
            logits = R.gen_head(W, outputs.last_hidden_state[:, -1, :])               # vision_head: Linear -> GELU -> Linear, bf16 under autocast
            logits = logits[1::2] + cfg_weight * (logits[0::2] - logits[1::2])         # plangen_base.py:585 on bf16 tensors
            assert logits.dtype == torch.bfloat16
            out_logits.append(logits.float())
            nxt = torch.argmax(logits, dim=-1, keepdim=True)
            toks[:, i] = nxt.squeeze(-1).int()
            if force is not None:
                nxt = force[:, i:i + 1].long()
            nxt = torch.cat([nxt.unsqueeze(1), nxt.unsqueeze(1)], dim=1).view(-1)
            inputs_embeds = R.prepare_gen_img_embeds(W, nxt).unsqueeze(1)              # bf16: every decode step's residual stream is bf16
            assert inputs_embeds.dtype == torch.bfloat16
            if log_every and i % log_every == 0:
                print(f"  {what} autocast step {i} ({time.time() - t0:.0f} s)", flush=True)
    return toks, torch.stack(out_logits)


def image_loop_anchor(name, cfg, W, g, sel_steps=None, free_steps=None):
    """g = the fp32 fixture (ids, pad, tokens, top_v, top_i, vsel, sel_logits[, sel_steps]).  Teacher-forced on the fp32 tokens, then free-running."""
    torch.set_num_threads(8)
    ids = torch.from_numpy(g["ids"].astype(np.int32))
    pad = torch.from_numpy(g["pad"].astype(np.int64))
    Rr, L = ids.shape
    gold = torch.from_numpy(g["tokens"]).int()
    B, T = gold.shape
    mask = torch.ones((Rr, L + cfg.img_tokens), dtype=torch.int32)
    for r in range(Rr):
        mask[r, :int(pad[r])] = 0
    sel = torch.arange(T) if sel_steps is None else torch.from_numpy(np.asarray(sel_steps)).long()
    if "vsel" in g:
        vsel = torch.from_numpy(g["vsel"]).long()
        ref32 = torch.from_numpy(g["sel_logits"])                                           # [S, B, |vsel|]
        top_v, top_i = torch.from_numpy(g["top_v"]), torch.from_numpy(g["top_i"]).long()    # [T, B, 4]
    else:                                                                                    # the tiny fixture stores whole logit rows
        full = torch.from_numpy(g["logits"])
        vsel = torch.arange(full.shape[-1]); ref32 = full
        top_v, top_i = full.topk(4, dim=-1)
    model = MG.hf_llama(cfg, W)

    tf_tok, tf_logits = autocast_sample_image(model, W, ids, mask, T, force=gold, what=name + " teacher-forced")
    ph = autocast_sample_image.prefill_hidden
    hid = {}
    if "prefill_last" in g:                                                              # last prompt position of every row (all rows are real there)
        hid["prefill_last"] = err_stats((ph[:, -1] - torch.from_numpy(g["prefill_last"])).abs())
    if "prefill_hidden" in g and "hid_rows" in g:                                        # full-width fixture: 6 rows x selected positions (real ones only)
        rows, pos = torch.from_numpy(g["hid_rows"]).long(), torch.from_numpy(g["pos_sel"]).long()
        real = (pos[None, :] >= pad[rows][:, None])
        hid["prefill_hidden_real_positions"] = err_stats((ph[rows][:, pos] - torch.from_numpy(g["prefill_hidden"])).abs()[real])
    elif "prefill_hidden" in g:                                                          # tiny fixture: every position
        real = torch.arange(L)[None, :] >= pad[:, None]
        hid["prefill_hidden_real_positions"] = err_stats((ph - torch.from_numpy(g["prefill_hidden"])).abs()[real])
    d = (tf_logits[sel][:, :, vsel] - ref32).abs()
    top_err = (tf_logits.gather(2, top_i[..., :1]).squeeze(-1) - top_v[..., 0]).abs()   # the bf16 value of fp32's top-1 column, every step
    agree = (tf_tok == gold).t()                                                        # [T, B]
    margin = top_v[..., 0] - top_v[..., 1]
    stats = {"all": err_stats(d), "top1_value_err_max_all_steps": float(top_err.max()), "top1_value_err_p99": _pct(top_err.numpy(), 99),
             "teacher_forced_agreement": float(agree.float().mean()), "logit_std": float(ref32.std()), "margin_median": float(margin.median()),
             "largest_margin_of_a_flip": float(margin[~agree].max()) if (~agree).any() else 0.0}
    stats["hidden"] = hid
    if T > 400:
        late = sel >= 400
        stats["late_steps_ge_400"] = dict(err_stats(d[late]), agreement=float(agree[400:].float().mean()))
        stats["early_steps_lt_400"] = dict(err_stats(d[~late]), agreement=float(agree[:400].float().mean()))

    nfree = T if free_steps is None else min(T, free_steps)
    fr_tok, _ = autocast_sample_image(model, W, ids, mask, nfree, force=None, what=name + " free-running")
    same = (fr_tok[:, :nfree] == gold[:, :nfree])
    first_div = [int((~same[b]).nonzero()[0]) if (~same[b]).any() else nfree for b in range(B)]
    stats["free_running"] = {"steps": nfree, "agreement_with_fp32_tokens": float(same.float().mean()), "first_divergence_step": first_div if B <= 8 else None,
                             "median_first_divergence": float(np.median(first_div))}
    print(name, json.dumps(stats, indent=1))
    np.savez_compressed(os.path.join(OUT, name + "_bf16ref.npz"), stats=json.dumps(stats), sel_steps=sel.numpy().astype(np.int32), vsel=vsel.numpy().astype(np.int32),
                        ref_bf16_sel_logits=tf_logits[sel][:, :, vsel].to(torch.bfloat16).view(torch.int16).numpy(),     # bf16 bit patterns (exact)
                        ref_bf16_tf_tokens=tf_tok.numpy(), ref_bf16_free_tokens=fr_tok.numpy().astype(np.int32),
                        ref_bf16_at_fp32_top1=tf_logits.gather(2, top_i[..., :1]).squeeze(-1).numpy(),       # [T, B]: the bf16 value of fp32's top-1 column
                        wsum=float(g["wsum"]))
    return stats


def anchor_fullconfig():
    g = np.load(os.path.join(OUT, "sample_image_fullconfig.npz"))
    cfg = R.OracleCfg()
    W = R.make_weights(cfg, seed=int(g["seed_w"]), with_lm_head=False)
    assert abs(MG.wsum(W) - float(g["wsum"])) < 1e-6 * float(g["wsum"])
    image_loop_anchor("sample_image_fullconfig", cfg, W, g, sel_steps=g["sel_steps"])
    vq_anchor("sample_image_fullconfig", cfg, W, torch.from_numpy(g["tokens"]))


def anchor_fulldepth():
    g = np.load(os.path.join(OUT, "sample_image_fulldepth.npz"))
    cfg = R.OracleCfg(**dict(MG.FULLW, n_layers=24))
    W = R.make_weights(cfg, seed=6)
    assert abs(MG.wsum(W) - float(g["wsum"])) < 1e-6 * float(g["wsum"])
    image_loop_anchor("sample_image_fulldepth", cfg, W, g)


def anchor_fullwidth():
    g = np.load(os.path.join(OUT, "sample_image_fullwidth.npz"))
    cfg = R.OracleCfg(**MG.FULLW)
    W = R.make_weights(cfg, seed=3)
    assert abs(MG.wsum(W) - float(g["wsum"])) < 1e-6 * float(g["wsum"])
    image_loop_anchor("sample_image_fullwidth", cfg, W, g)


def anchor_b8_long():
    g = np.load(os.path.join(OUT, "sample_image_b8_long.npz"))
    cfg = R.OracleCfg(**MG.FULLW)
    W = R.make_weights(cfg, seed=3)
    image_loop_anchor("sample_image_b8_long", cfg, W, g)


def anchor_tiny():
    g = np.load(os.path.join(OUT, "sample_image_tiny.npz"), allow_pickle=True)
    cfg = R.OracleCfg(**MG.TINY)
    W = R.make_weights(cfg, seed=1)
    assert abs(MG.wsum(W) - float(g["wsum"])) < 1e-6 * float(g["wsum"])
    L = g["ids"].shape[1]
    g = dict(g); g["pad"] = (L - g["mask"][:, :L].sum(-1)).astype(np.int32)
    image_loop_anchor("sample_image_tiny", cfg, W, g)


@torch.no_grad()
def anchor_prefill_long():
    """prefill_long_fullwidth.npz (8 rows x 640 positions, positions = mask cumsum): the prompt's hidden states under the reference's autocast."""
    torch.set_num_threads(8)
    g = np.load(os.path.join(OUT, "prefill_long_fullwidth.npz"))
    cfg = R.OracleCfg(**MG.FULLW)
    W = R.make_weights(cfg, seed=3)
    assert abs(MG.wsum(W) - float(g["wsum"])) < 1e-6 * float(g["wsum"])
    model = MG.hf_llama(cfg, W)
    ids = torch.from_numpy(g["ids"].astype(np.int32)); pad = torch.from_numpy(g["pad"].astype(np.int64))
    L = ids.shape[1]
    mask = (torch.arange(L)[None, :] >= pad[:, None]).int()
    pos = (mask.long().cumsum(-1) - 1).clamp(min=0)
    with torch.autocast(**BF):
        out = model(inputs_embeds=model.get_input_embeddings()(ids.long()), attention_mask=mask, position_ids=pos, use_cache=False).last_hidden_state.float()
    ps = torch.from_numpy(g["pos_sel"]).long()
    real = ps[None, :] >= pad[:, None]
    d = (out[:, ps] - torch.from_numpy(g["hidden"])).abs()[real]
    stats = {"hidden": {"prefill_hidden_real_positions": err_stats(d)}, "hidden_abs_max": float(torch.from_numpy(g["hidden"]).abs().max())}
    print("prefill_long_fullwidth", json.dumps(stats))
    np.savez_compressed(os.path.join(OUT, "prefill_long_fullwidth_bf16ref.npz"), stats=json.dumps(stats), wsum=float(g["wsum"]))


def anchor_vq_full():
    """The stand-alone full-size VQ fixture (vq_full.npz: one image of seeded codes through the reference's VQ-16)."""
    g = np.load(os.path.join(OUT, "vq_full.npz"))
    cfg = R.OracleCfg(n_layers=0, vocab=8)          # only the VQ part is used (as in make_golden.golden_vq_full)
    W = R.make_weights(cfg, seed=2, with_lm_head=False)
    assert abs(MG.wsum(MG.sub(W, "gen_vision_model.")) - float(g["wsum"])) < 1e-6 * float(g["wsum"])
    return vq_anchor("vq_full", cfg, W, torch.from_numpy(g["codes"].astype(np.int32)))


# ------------------------------------------------------------------------------------------------------------------ greedy text (a11)
@torch.no_grad()
def text_anchor(name, cfg, W, g, N, unused_eos):
    """System.x2t -> language_model.generate (plangen_base.py:513-523) under the reference's autocast: the un-stopped greedy run of
    ``LlamaForCausalLM.generate``; its ids are then forced into the FP32 oracle and measured exactly like the engine's bf16 ids are in the GPU
    tests: gap = (fp32 best logit) - (fp32 logit of the bf16 path's token), 0 where the argmax agrees."""
    torch.set_num_threads(8)
    lm = MG.hf_llama(cfg, W, causal_lm=True)
    ids, mask = torch.from_numpy(g["ids"].astype(np.int32)), torch.from_numpy(g["mask"].astype(np.int32))
    with torch.autocast(**BF):
        emb = lm.get_input_embeddings()(ids.long())
        out = lm.generate(inputs_embeds=emb, attention_mask=mask, pad_token_id=unused_eos, bos_token_id=1, eos_token_id=unused_eos,
                          max_new_tokens=N, min_new_tokens=N, do_sample=False, use_cache=True)
    assert out.shape == (ids.shape[0], N), out.shape
    # Logit-level anchor with MANY samples (the id-level "gap" below is a max over the handful of steps whose argmax flips): next-token logits at
    # every REAL prompt position (the prompt is teacher forcing by construction), fp32 vs autocast, positions = mask cumsum as in generate().
    L = ids.shape[1]
    m64 = mask[:, :L].long()
    pos = (m64.cumsum(-1) - 1).clamp(min=0)
    real = m64.bool()
    gsel = torch.Generator().manual_seed(97)
    csel = torch.randperm(cfg.vocab, generator=gsel)[:128].sort().values
    emb32 = lm.get_input_embeddings()(ids.long())
    lg32 = lm(inputs_embeds=emb32, attention_mask=m64, position_ids=pos, use_cache=False).logits[real]                 # [P, V] fp32
    with torch.autocast(**BF):
        lgbf = lm(inputs_embeds=emb32, attention_mask=m64, position_ids=pos, use_cache=False).logits[real]
    assert lgbf.dtype == torch.bfloat16
    t1v, t1i = lg32.max(-1)
    d_all = (lgbf.float() - lg32).abs()
    prompt_stats = {"positions": int(real.sum()), "all_columns": err_stats(d_all), "sel_columns": err_stats(d_all[:, csel]),
                    "top1_value_err_max": float((lgbf.float().gather(1, t1i[:, None]).squeeze(1) - t1v).abs().max()),
                    "argmax_agreement": float((lgbf.float().argmax(-1) == t1i).float().mean()), "logit_std": float(lg32.std())}
    prompt_sel32 = lg32[:, csel].clone(); prompt_selbf = lgbf[:, csel].clone(); prompt_top1 = torch.stack([t1v, t1i.float()], 1)
    del lm, lg32, lgbf, d_all
    _, logits = R.generate_text_greedy(W, cfg, R.embed_tokens(W, ids), mask, N, unused_eos, min_new_tokens=N, force_tokens=out, return_logits=True)
    lg = logits.permute(1, 0, 2).clone()
    lg[:, :, unused_eos] = float("-inf")
    gap = lg.max(-1).values - torch.gather(lg, 2, out[..., None]).squeeze(-1)
    probe = torch.from_numpy(g["probe"].astype(np.int64))
    stats = {"worst_logit_gap": float(gap.max()), "gap_p99": _pct(gap.numpy(), 99), "gap_mean": float(gap.mean()),
             "argmax_agreement_on_own_prefix": float((gap == 0).float().mean()),
             "free_running_ids_equal_fp32": float((out == probe[:, :N]).float().mean()), "prompt_logits": prompt_stats}
    print(name, "text, reference-bf16 ids in the fp32 oracle:", json.dumps(stats))
    np.savez_compressed(os.path.join(OUT, name + "_bf16ref.npz"), stats=json.dumps(stats), ref_bf16_ids=out.numpy().astype(np.int32), wsum=float(g["wsum"]),
                        csel=csel.numpy().astype(np.int32), prompt_sel_fp32=prompt_sel32.numpy(), prompt_top1=prompt_top1.numpy(),
                        prompt_sel_ref_bf16=prompt_selbf.view(torch.int16).numpy())
    return stats


def anchor_text_fullconfig():
    g = np.load(os.path.join(OUT, "generate_fullconfig.npz"))
    cfg = R.OracleCfg()
    W = R.make_weights(cfg, seed=int(g["seed_w"]), with_lm_head=True)
    assert abs(MG.wsum(W) - float(g["wsum"])) < 1e-6 * float(g["wsum"])
    text_anchor("generate_fullconfig", cfg, W, g, N=g["probe"].shape[1], unused_eos=int(g["unused_eos"]))


def anchor_text_fullvocab():
    g = np.load(os.path.join(OUT, "generate_fullvocab.npz"))
    cfg = R.OracleCfg(**MG.FULLV)
    W = R.make_weights(cfg, seed=11)
    assert abs(MG.wsum(W) - float(g["wsum"])) < 1e-6 * float(g["wsum"])
    text_anchor("generate_fullvocab", cfg, W, g, N=g["probe"].shape[1], unused_eos=cfg.eos_id)


def anchor_text_fullwidth():
    g = np.load(os.path.join(OUT, "generate_fullwidth.npz"))
    cfg = R.OracleCfg(**MG.FULLW)
    W = R.make_weights(cfg, seed=3)
    assert abs(MG.wsum(W) - float(g["wsum"])) < 1e-6 * float(g["wsum"])
    text_anchor("generate_fullwidth", cfg, W, g, N=g["probe"].shape[1], unused_eos=cfg.eos_id)


# ------------------------------------------------------------------------------------------------------------------ VQ-16 pixels
class _cuda_autocast_policy_for_norms:
    """CUDA autocast runs group_norm, layer_norm and softmax in fp32 (autocast_mode.cpp, fp32 cast policy); the CPU policy does not list them.
    Wrap the three functionals so the reference module executes what it would under the reference's ``device_type='cuda'`` autocast."""

    def __enter__(self):
        import torch.nn.functional as F
        self.F, self.gn, self.sm, self.ln = F, F.group_norm, F.softmax, F.layer_norm

        def gn(x, num_groups, weight=None, bias=None, eps=1e-5):
            with torch.autocast(device_type="cpu", enabled=False):
                return self.gn(x.float(), num_groups, weight, bias, eps)

        def sm(x, dim=None, _stacklevel=3, dtype=None):
            with torch.autocast(device_type="cpu", enabled=False):
                return self.sm(x.float(), dim=dim, dtype=dtype)

        def ln(x, normalized_shape, weight=None, bias=None, eps=1e-5):
            with torch.autocast(device_type="cpu", enabled=False):
                return self.ln(x.float(), normalized_shape, weight, bias, eps)

        F.group_norm, F.softmax, F.layer_norm = gn, sm, ln
        return self

    def __exit__(self, *a):
        self.F.group_norm, self.F.softmax, self.F.layer_norm = self.gn, self.sm, self.ln


@torch.no_grad()
def vq_anchor(name, cfg, W, tokens):
    """The reference's OWN ``VQ_models["VQ-16"].decode_code`` (vq_model.py:505-508; Upsample's explicit bf16 cast :417-421) on the same tokens in fp32 and
    under autocast: pixel MSE of reference-bf16 against reference-fp32 = the share of north_star's 1e-4 budget the reference's own dtype uses."""
    m = MG.ref_vq_module()
    vq = m.VQ_models["VQ-16"]().eval()
    missing = vq.load_state_dict(MG.sub(W, "gen_vision_model."), strict=False)
    assert all(k.startswith(("encoder.", "quant_conv", "quantize.codebook_used")) for k in missing.missing_keys), missing
    B = tokens.shape[0]
    shape = [B, 8, cfg.grid, cfg.grid]
    img32 = vq.decode_code(tokens.int(), shape=shape)
    res = {}
    with torch.autocast(**BF):
        img_cpu = vq.decode_code(tokens.int(), shape=shape)
    with _cuda_autocast_policy_for_norms(), torch.autocast(**BF):
        img_cuda = vq.decode_code(tokens.int(), shape=shape)
    for k, im in (("cpu_policy", img_cpu), ("cuda_policy", img_cuda)):
        assert im.dtype == torch.bfloat16, im.dtype
        d = im.float() - img32
        res[k] = {"pixel_mse": float((d ** 2).mean()), "max_abs": float(d.abs().max()), "per_image_mse": [float((d[b] ** 2).mean()) for b in range(B)]}
    res["image_std"] = float(img32.std())
    print(name, "VQ-16 decode_code, reference-bf16 vs reference-fp32:", json.dumps(res))
    pooled = torch.nn.functional.avg_pool2d(img_cuda.float(), 8)
    np.savez_compressed(os.path.join(OUT, name + "_vq_bf16ref.npz"), stats=json.dumps(res), pooled_cuda_policy=pooled.numpy())
    return res


# ------------------------------------------------------------------------------------------------------------------ SigLIP tower + aligner (a13)
@torch.no_grad()
def siglip_anchor(name, cfg, W, img, select_layer):
    """The reference's own VisionTransformer / CLIPVisionTower classes (stand-in PatchEmbed / Mlp as in make_golden: parity stays 'unpinned' in the
    strict sense) + the mlp_gelu aligner, fp32 vs the reference's autocast: ``prepare_inputs_embeds`` casts the pixels to bf16 itself
    (modeling_vlm.py:249) and runs ``aligner(vision_model(images))`` under autocast (:250).  Features = tower output (after the final LayerNorm)."""
    torch.set_num_threads(8)
    g = np.load(os.path.join(OUT, name + ".npz"))
    assert abs(MG.wsum(W) - float(g["wsum"])) < 1e-6 * float(g["wsum"])
    sv, ce = MG.ref_siglip_modules()
    tower = ce.CLIPVisionTower(model_name="siglip_large_patch16_384", image_size=cfg.vit_img, select_feature="same", select_layer=select_layer).eval()
    tower.vision_tower.attn_pool = None                                  # plangen_base.py:105-106
    MG._load_vit_weights(tower.vision_tower, W, cfg)
    al = lambda f: R.mlp_gelu_projector(f, W["aligner.layers.0.weight"], W["aligner.layers.0.bias"], W["aligner.layers.2.weight"], W["aligner.layers.2.bias"])
    f32 = tower(img); a32 = al(f32)
    tok = torch.from_numpy(g["tok"]).long()
    assert (f32[:, tok] - torch.from_numpy(g["features"])).abs().max().item() < 1e-4
    res = {}
    for pol in ("cpu_policy", "cuda_policy"):
        ctx = _cuda_autocast_policy_for_norms() if pol == "cuda_policy" else torch.autocast(device_type="cpu", enabled=False)
        with ctx, torch.autocast(**BF):
            fb = tower(img.to(torch.bfloat16)); ab = al(fb)
        res[pol] = {"features": err_stats((fb.float() - f32).abs()), "aligned": err_stats((ab.float() - a32).abs()),
                    "features_at_fixture_tokens": err_stats((fb.float()[:, tok] - f32[:, tok]).abs()),
                    "aligned_at_fixture_tokens": err_stats((ab.float()[:, tok] - a32[:, tok]).abs())}
    res["feat_std"], res["aligned_absmax"] = float(f32.std()), float(a32.abs().max())
    print(name, "SigLIP tower + aligner, reference-bf16 vs reference-fp32:", json.dumps(res))
    np.savez_compressed(os.path.join(OUT, name + "_bf16ref.npz"), stats=json.dumps(res), wsum=float(g["wsum"]))
    return res


def anchor_siglip_fulldepth():
    cfg = R.OracleCfg(**dict(MG.VISW, vit_layers=24))
    siglip_anchor("siglip_fulldepth", cfg, R.make_weights(cfg, seed=12, with_vision=True), MG.siglip_fullwidth_images(cfg, n=1, seed=43), -1)


def anchor_siglip_fullwidth():
    cfg = R.OracleCfg(**MG.VISW)
    siglip_anchor("siglip_fullwidth", cfg, R.make_weights(cfg, seed=7, with_vision=True), MG.siglip_fullwidth_images(cfg), cfg.vit_layers)


# ------------------------------------------------------------------------------------------------------------------ VQ-16 encoder (a14)
@torch.no_grad()
def anchor_vq_encode():
    """t2i encodes ``gt_image.bfloat16()`` under autocast (plangen_base.py:530): the reference's OWN VQ_models["VQ-16"].encode on the fixture's image in
    fp32 and under autocast -- share of the 576 indices that survive, and the fp32 distance gap at the indices that do not."""
    torch.set_num_threads(8)
    g = np.load(os.path.join(OUT, "vq_full_encode.npz"))
    cfg = R.OracleCfg(n_layers=0, vocab=8)
    W = R.make_weights(cfg, seed=4, with_lm_head=False, with_encoder=True)
    assert abs(MG.wsum(MG.sub(W, "gen_vision_model.")) - float(g["wsum"])) < 1e-6 * float(g["wsum"])
    m = MG.ref_vq_module()
    vq = m.VQ_models["VQ-16"]().eval()
    missing = vq.load_state_dict(MG.sub(W, "gen_vision_model."), strict=False)
    assert all("codebook_used" in k for k in missing.missing_keys) and not missing.unexpected_keys, missing
    x = torch.from_numpy(g["image_u8"]).float() / 127.5 - 1.0
    idx32 = vq.encode(x)[2][-1].reshape(-1)
    assert np.array_equal(idx32.numpy(), g["idx"].astype(np.int64))
    gap = torch.from_numpy(g["gap"])
    res = {}
    for pol in ("cpu_policy", "cuda_policy"):
        ctx = _cuda_autocast_policy_for_norms() if pol == "cuda_policy" else torch.autocast(device_type="cpu", enabled=False)
        with ctx, torch.autocast(**BF):
            idxb = vq.encode(x.bfloat16())[2][-1].reshape(-1)
        diff = idxb != idx32
        res[pol] = {"indices_equal_fp32": float((~diff).float().mean()), "mismatches": int(diff.sum()),
                    "largest_fp32_gap_at_a_mismatch": float(gap[diff].max()) if diff.any() else 0.0, "median_gap": float(gap.median())}
    print("vq_full_encode, reference-bf16 vs reference-fp32:", json.dumps(res))
    np.savez_compressed(os.path.join(OUT, "vq_full_encode_bf16ref.npz"), stats=json.dumps(res), wsum=float(g["wsum"]))
    return res


def main():
    names = sys.argv[1:] or ["fulldepth", "fullwidth", "b8_long", "fullconfig", "text_fullconfig", "text_fullvocab", "text_fullwidth"]
    for n in names:
        globals()["anchor_" + n]()


if __name__ == "__main__":
    main()
