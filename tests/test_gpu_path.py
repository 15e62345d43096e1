"""GPU: the hot path (prefill -> 576-step CFG decode loop -> VQ decode) through the C ABI,
against (a) the golden vectors produced by the reference modules / transformers and (b) the
CPU oracle on the same seeded inputs.

Tolerances
  * PG_F32 mode: indices bit-exact; logits / hidden within 2e-3 abs (fp32 summation order).
  * PG_BF16 mode (bf16 operands, fp32 accumulate): teacher-forced protocol -- the oracle's
    token is fed back at every step, CFG-mixed logits must agree within LOGIT_TOL_BF16 and
    the argmax must be identical wherever the oracle's top-1 margin exceeds 2x that tolerance.
  * VQ-decoded pixels: MSE <= 1e-4 (north_star), outputs in ~[-1, 1].
"""
import numpy as np
import pytest
import torch

from conftest import get_engine, load_golden
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu

LOGIT_TOL_F32 = 2e-3
# PG_BF16 bounds (round 6) = K x what the REFERENCE'S OWN bf16 arithmetic does on this fixture (tests/golden/sample_image_tiny_bf16ref.npz: the same loop under
# torch.autocast(bfloat16), fp32 master weights, plangen_base.py:95,360; E_ref logits max 0.0094 at logit std 0.11, prefill hidden max 0.022) -- tests/bf16ref.py
import bf16ref
_EREF = bf16ref.load("sample_image_tiny")[1]
LOGIT_TOL_BF16 = bf16ref.K_MAX * _EREF["all"]["max"]
HIDDEN_TOL_BF16 = bf16ref.hidden_bound("sample_image_tiny", "prefill_hidden_real_positions")
PIXEL_MSE = 1e-4


def _golden():
    g = load_golden("sample_image_tiny.npz")
    return g, torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])


def _pad(mask, L):
    return (L - mask[:, :L].sum(-1)).tolist()


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-4), ("bf16", HIDDEN_TOL_BF16)])
def test_prefill_hidden_matches_transformers_fixture(tiny_cfg, tiny_weights, dtype, tol):
    g, ids, mask = _golden()
    e = get_engine(tiny_cfg, tiny_weights, dtype)
    L = ids.shape[1]
    hid = e.prefill(ids, _pad(mask, L), position_mode=0, return_hidden=True).cpu()
    ref = torch.from_numpy(g["prefill_hidden"])
    real = mask[:, :L].bool()
    assert (hid - ref)[real].abs().max() < tol
    assert hid[~real].abs().max() == 0          # pad slots are skipped, reported as zeros


def test_kv_cache_matches_oracle(tiny_cfg, tiny_weights, ocfg):
    g, ids, mask = _golden()
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    L = ids.shape[1]
    pad = _pad(mask, L)
    e.prefill(ids, pad, position_mode=0)
    pos = torch.arange(L)[None].expand(ids.shape[0], -1)
    _, cache = R.llama_forward(tiny_weights, ocfg, R.embed_tokens(tiny_weights, ids), mask, pos)
    slots = 96 + tiny_cfg.img_tokens
    for layer in range(tiny_cfg.n_layers):
        for name, ref in (("kcache", cache.k[layer]), ("vcache", cache.v[layer])):
            buf = e.debug_read(name, layer, 8 * tiny_cfg.n_heads * slots * 128, torch.float32).cpu()
            buf = buf.view(8, tiny_cfg.n_heads, slots, 128)
            for r in range(ids.shape[0]):
                n = L - pad[r]
                if r % 2 == 1 and r != 1:
                    # uncond rows share one negative prompt: its K/V is stored once (row 1) and
                    # aliased -- legitimate only because the oracle's K/V are identical too
                    assert torch.equal(ref[r, :, pad[r]:], ref[1, :, pad[1]:])
                    continue
                assert (buf[r, :, :n] - ref[r, :, pad[r]:]).abs().max() < 1e-4, (name, layer, r)


def test_decode_tokens_f32_bit_exact_vs_reference_loop(tiny_cfg, tiny_weights):
    g, ids, mask = _golden()
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    e.prefill(ids, _pad(mask, ids.shape[1]), position_mode=0)
    toks, logits = e.decode_image_tokens(cfg_weight=5.0, temperature=0.0, return_logits=True)
    assert np.array_equal(toks.cpu().numpy(), g["tokens"])
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < LOGIT_TOL_F32


def test_shared_uncond_prompt_equals_private_copies(tiny_cfg, tiny_weights):
    """Storing the batch-constant negative prompt's K/V once must not change a single token."""
    g, ids, mask = _golden()
    for dtype in ("f32", "bf16"):
        e = get_engine(tiny_cfg, tiny_weights, dtype)
        outs = []
        for share in (1, 0):
            e.set_option("share_uncond", share)
            e.prefill(ids, _pad(mask, ids.shape[1]), position_mode=0)
            outs.append(e.decode_image_tokens(cfg_weight=5.0, temperature=0.0, return_logits=True))
        e.set_option("share_uncond", 1)
        assert torch.equal(outs[0][0], outs[1][0]), dtype
        # keys are split differently over waves -> same math, different fp32 summation order
        assert (outs[0][1] - outs[1][1]).abs().max() < (1e-4 if dtype == "f32" else 5e-2), dtype


def test_decode_graph_equals_eager(tiny_cfg, tiny_weights):
    g, ids, mask = _golden()
    for dtype in ("f32", "bf16"):
        e = get_engine(tiny_cfg, tiny_weights, dtype)
        outs = []
        for use_graph in (1, 0):
            e.set_option("use_graph", use_graph)
            e.prefill(ids, _pad(mask, ids.shape[1]), position_mode=0)
            outs.append(e.decode_image_tokens(cfg_weight=5.0, temperature=0.0).cpu())
        e.set_option("use_graph", 0)          # back to the default (stream launches)
        assert torch.equal(outs[0], outs[1]), dtype


def test_decode_bf16_teacher_forced(tiny_cfg, tiny_weights):
    g, ids, mask = _golden()
    e = get_engine(tiny_cfg, tiny_weights, "bf16")
    e.prefill(ids, _pad(mask, ids.shape[1]), position_mode=0)
    gold_tok = torch.from_numpy(g["tokens"])
    toks, logits = e.decode_image_tokens(cfg_weight=5.0, temperature=0.0, force_tokens=gold_tok, return_logits=True)
    ref = torch.from_numpy(g["logits"])                    # [T, B, V]
    err = (logits.cpu() - ref).abs().max().item()
    print(f"tiny bf16 teacher-forced: max |logit err| {err:.4f} (bound = the reference-bf16's own max error {LOGIT_TOL_BF16:.4f})")
    g32 = dict(g); g32["pad"] = np.array(_pad(mask, ids.shape[1]))
    bf16ref.check_image_loop("sample_image_tiny", logits.cpu(), toks.cpu(), g32, "tiny bf16 teacher-forced")
    assert err < LOGIT_TOL_BF16, err
    top2 = ref.topk(2, dim=-1).values
    decisive = (top2[..., 0] - top2[..., 1]) > 2 * LOGIT_TOL_BF16   # [T, B]
    got = toks.cpu().t()                                   # [T, B]
    assert torch.equal(got[decisive], gold_tok.t()[decisive])
    # most steps agree outright
    assert (got == gold_tok.t()).float().mean() > 0.8


def test_stepwise_facade_loop_equals_fused_loop(tiny_cfg, tiny_weights):
    from plangen_amd.system import System
    g, ids, mask = _golden()
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    sysm = System(tiny_cfg, e)
    emb = sysm.vl_gpt.language_model.get_input_embeddings()(ids.to(e.device))
    toks = sysm.sample_image_stepwise(emb, mask.to(e.device), 5.0, n_tokens=12)
    assert np.array_equal(toks.cpu().numpy(), g["tokens"][:, :12])


def test_teacher_forcing_edit_region(tiny_cfg, tiny_weights, ocfg):
    """use_teacher_forcing branch (plangen_base.py:593-598) against the oracle."""
    from plangen_amd.system import System
    g, ids, mask = _golden()
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    sysm = System(tiny_cfg, e)
    gen = torch.Generator().manual_seed(5)
    B, T = ids.shape[0] // 2, 16
    gt = torch.randint(0, tiny_cfg.img_vocab, (B, T), generator=gen).int()
    region = (torch.rand(B, T, generator=gen) > 0.5).int()
    ref = R.sample_image(tiny_weights, ocfg, R.embed_tokens(tiny_weights, ids), mask, 5.0, n_tokens=T,
                         edit_region=region, gt_labels=gt)
    got = sysm.sample_image(ids, mask, 5.0, 0.0, edit_region=region, gt_labels=gt, n_tokens=T)
    assert torch.equal(got.cpu(), ref)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_vq_decode_matches_reference_module(tiny_cfg, tiny_weights, dtype):
    g = load_golden("vq_tiny.npz")
    e = get_engine(tiny_cfg, tiny_weights, dtype)
    img = e.vq_decode(torch.from_numpy(g["codes"])).cpu()
    ref = torch.from_numpy(g["image"])
    mse = ((img - ref) ** 2).mean().item()
    if dtype == "f32":
        assert (img - ref).abs().max() < 2e-4
    assert mse <= PIXEL_MSE, mse


def test_vq_encode_indices_bit_exact(tiny_cfg, tiny_weights):
    g = load_golden("vq_tiny.npz")
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    idx = e.vq_encode(torch.from_numpy(g["enc_in"])).cpu()
    assert np.array_equal(idx.numpy(), g["enc_idx"])


def test_t2i_end_to_end_f32(tiny_cfg, tiny_weights):
    from plangen_amd.system import System
    g, ids, mask = _golden()
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    sysm = System(tiny_cfg, e)
    dec, mask_image = sysm.t2i(ids, mask, cfg_weight=5.0, temperature=0.0)
    toks = sysm.last_generated_tokens
    assert mask_image is None                                   # no teacher forcing (plangen_base.py:561-562)
    assert np.array_equal(toks.cpu().numpy(), g["tokens"])
    assert ((dec.cpu() - torch.from_numpy(g["image"])) ** 2).mean().item() <= PIXEL_MSE


def test_text_greedy_matches_hf_generate(tiny_cfg, tiny_weights):
    from plangen_amd.system import System
    g = load_golden("generate_tiny.npz")
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    sysm = System(tiny_cfg, e)
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
    emb = sysm.vl_gpt.language_model.get_input_embeddings()(ids.to(e.device))
    sysm.cfg.eos_id = int(g["eos"])
    try:
        out = sysm.x2t(emb, mask.to(e.device), max_new_tokens=10)
    finally:
        sysm.cfg.eos_id = 7
    assert np.array_equal(out.cpu().numpy(), g["out"])


def test_gen_head_and_gen_embed_ops(tiny_cfg, tiny_weights):
    g = load_golden("proj_tiny.npz")
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    out = e.gen_embed(torch.from_numpy(g["ids"])).cpu()
    assert np.abs(out.numpy() - g["out"]).max() < 1e-5          # vs reference projector.py
    gs = load_golden("sample_image_tiny.npz")
    h = torch.from_numpy(gs["last_hidden"][0])
    logits = e.gen_head(h).cpu()
    ref = R.gen_head(tiny_weights, h)
    assert (logits - ref).abs().max() < 1e-4


def test_sampling_is_seeded_and_follows_softmax(tiny_cfg, tiny_weights):
    """temperature>0: Gumbel-max == multinomial(softmax(logits/T)); check determinism per seed
    and that the empirical distribution of the first token tracks softmax of the oracle logits."""
    g, ids, mask = _golden()
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    pad = _pad(mask, ids.shape[1])
    draws = []
    for seed in range(200):
        e.prefill(ids, pad, position_mode=0)
        draws.append(e.decode_image_tokens(T=1, cfg_weight=5.0, temperature=1.0, seed=seed).cpu()[:, 0])
    draws = torch.stack(draws)                                  # [200, B]
    e.prefill(ids, pad, position_mode=0)
    again = e.decode_image_tokens(T=1, cfg_weight=5.0, temperature=1.0, seed=7).cpu()[:, 0]
    assert torch.equal(again, draws[7])
    p = torch.softmax(torch.from_numpy(g["logits"][0]), dim=-1)  # [B, V]
    for b in range(p.shape[0]):
        top = p[b].topk(5).indices
        emp = torch.stack([(draws[:, b] == t).float().mean() for t in top])
        assert (emp - p[b][top]).abs().max() < 0.12


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-4), ("bf16", 6e-2)])
def test_vision_encoder_matches_oracle(tiny_cfg, tiny_weights, ocfg, dtype, tol):
    """SigLIP + aligner (a13).  Oracle = restatement of siglip_vit.py / clip_encoder.py
    (parity unpinned: timm absent, see oracle/ref_cpu.py)."""
    e = get_engine(tiny_cfg, tiny_weights, dtype)
    g = torch.Generator().manual_seed(31)
    img = torch.rand(3, 3, tiny_cfg.vit_img, tiny_cfg.vit_img, generator=g) * 2 - 1
    ref = R.vision_encode(tiny_weights, ocfg, img)
    out = e.vision_encode(img).cpu()
    assert out.shape == ref.shape == (3, tiny_cfg.vit_tokens, tiny_cfg.hidden)
    assert (out - ref).abs().max() < tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-4), ("bf16", 6e-2)])
def test_vision_encoder_matches_third_party_fixture(tiny_cfg, tiny_weights, dtype, tol):
    """a13 against the committed cross-check fixture (transformers.SiglipVisionModel output through the aligner): an
    independent implementation of the published architecture, not the reference's timm class (parity stays unpinned)."""
    g = load_golden("siglip_tiny_crosscheck.npz")
    e = get_engine(tiny_cfg, tiny_weights, dtype)
    out = e.vision_encode(torch.from_numpy(g["images"])).cpu()
    ref = torch.from_numpy(g["aligned"])
    assert (out - ref).abs().max() < tol * max(1.0, ref.abs().max().item())


def test_mmu_path_prepare_inputs_embeds_then_generate(tiny_cfg, tiny_weights, ocfg):
    """task_type='mmu' (plangen_base.py:365-366, :513-523): image -> SigLIP -> scatter into the text
    embeddings -> greedy text decode; ids bit-exact vs the oracle in fp32."""
    from plangen_amd.system import System
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    sysm = System(tiny_cfg, e)
    g = torch.Generator().manual_seed(32)
    B, P = 2, tiny_cfg.vit_tokens
    Ltxt = [5, 3]
    L = P + max(Ltxt) + 2
    ids = torch.full((B, L), tiny_cfg.pad_id, dtype=torch.int64)
    seq_mask = torch.zeros((B, L), dtype=torch.bool)
    attn = torch.zeros((B, L), dtype=torch.int32)
    for b in range(B):
        n = 1 + P + Ltxt[b] + 1
        row = torch.randint(8, tiny_cfg.vocab, (n,), generator=g)
        ids[b, L - n:] = row
        seq_mask[b, L - n + 1: L - n + 1 + P] = True          # image placeholder slots
        attn[b, L - n:] = 1
    pix = torch.rand(B, 1, 3, tiny_cfg.vit_img, tiny_cfg.vit_img, generator=g) * 2 - 1
    emb_mask = torch.ones((B, 1, P), dtype=torch.bool)
    ref_emb = R.prepare_inputs_embeds(tiny_weights, ocfg, ids, pix, seq_mask, emb_mask)
    emb = sysm.vl_gpt.prepare_inputs_embeds(input_ids=ids, pixel_values=pix, images_seq_mask=seq_mask, images_emb_mask=emb_mask)
    assert (emb.cpu() - ref_emb).abs().max() < 2e-4
    ref = R.generate_text_greedy(tiny_weights, ocfg, ref_emb, attn, 8, tiny_cfg.eos_id)
    out = sysm.x2t(emb, attn.to(e.device), max_new_tokens=8)
    assert np.array_equal(out.cpu().numpy(), ref.numpy())


def test_two_decode_lanes_equal_one_lane(tiny_cfg, tiny_weights, ocfg):
    """The batch split into two row-range lanes on two streams (large-batch decode path) must give
    the same tokens as one lane; fp32 mode is bit-exact (per-row math does not depend on M)."""
    from plangen_amd.system import t2i_infer_collate_batch
    g = torch.Generator().manual_seed(41)
    cond = [torch.randint(8, tiny_cfg.vocab, (n,), generator=g).tolist() for n in (9, 12, 5, 7)]
    neg = torch.randint(8, tiny_cfg.vocab, (6,), generator=g).tolist()
    ids, mask = t2i_infer_collate_batch(cond, neg, tiny_cfg.pad_id, tiny_cfg.img_tokens)
    pad = _pad(mask, ids.shape[1])
    ref = R.sample_image(tiny_weights, ocfg, R.embed_tokens(tiny_weights, ids), mask, 5.0, n_tokens=10)
    for dtype in ("f32", "bf16"):
        e = get_engine(tiny_cfg, tiny_weights, dtype)
        outs = []
        for lanes in (1, 2):
            e.set_option("lanes", lanes)
            for use_graph in (1, 0):
                e.set_option("use_graph", use_graph)
                e.prefill(ids, pad, position_mode=0)
                outs.append(e.decode_image_tokens(T=10, cfg_weight=5.0, temperature=0.0).cpu())
        e.set_option("lanes", -1)
        e.set_option("use_graph", 0)          # back to the default (stream launches)
        for o in outs[1:]:
            assert torch.equal(o, outs[0]), dtype
        if dtype == "f32":
            assert torch.equal(outs[0], ref)
    # sampling is lane-independent too (RNG keyed on the global image index)
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    samp = []
    for lanes in (1, 2):
        e.set_option("lanes", lanes)
        e.prefill(ids, pad, position_mode=0)
        samp.append(e.decode_image_tokens(T=10, cfg_weight=5.0, temperature=1.0, seed=3).cpu())
    e.set_option("lanes", -1)
    assert torch.equal(samp[0], samp[1])


def test_uni_2stage_layout_then_image(tiny_cfg, tiny_weights, ocfg):
    """task_type='uni_2stage' (plangen_base.py:1117-1118): greedy layout tokens, host hand-off, then
    the CFG image loop; both stages bit-exact vs the oracle in fp32."""
    from plangen_amd.system import System, pad_input_ids
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    sysm = System(tiny_cfg, e)
    sysm.args.temperature = 0.0
    g = torch.Generator().manual_seed(51)
    stage1 = [torch.randint(8, tiny_cfg.vocab, (n,), generator=g).tolist() for n in (7, 4)]
    neg = torch.randint(8, tiny_cfg.vocab, (5,), generator=g).tolist()
    ids1, mask1 = pad_input_ids(stage1, tiny_cfg.pad_id)

    def layout_to_prompt(i, new_ids):            # stand-in for decode_plan_text_batch + wrap_uni_prompt
        keep = [t for t in new_ids if t != tiny_cfg.eos_id][:6]
        return stage1[i] + keep + [9]

    batch = dict(uni_stage1_inputs_ids=ids1, uni_stage1_attention_mask=mask1, neg_inputs_ids=neg)
    out = sysm.uni_generate(batch, pred_layout=True, layout_to_prompt=layout_to_prompt, max_new_tokens=8)
    ref_layout = R.generate_text_greedy(tiny_weights, ocfg, R.embed_tokens(tiny_weights, ids1), mask1, 8, tiny_cfg.eos_id)
    assert np.array_equal(out["pr_layout_ids"].cpu().numpy(), ref_layout.numpy())
    cond = [layout_to_prompt(i, r) for i, r in enumerate(ref_layout.tolist())]
    cids, cmask = R.t2i_infer_collate_batch(cond, neg, tiny_cfg.pad_id, tiny_cfg.img_tokens)
    ref_tok, ref_img = R.t2i(tiny_weights, ocfg, cids, cmask, 5.0)
    assert np.array_equal(out["pr_tokens"].cpu().numpy(), ref_tok.numpy())
    assert ((out["pr_image"].cpu() - ref_img) ** 2).mean().item() <= PIXEL_MSE


def test_checkpoint_formats_roundtrip(tiny_cfg, tiny_weights, tmp_path):
    """HF safetensors base + PlanGen '.pth' overlay with the 'vl_gpt.' prefix (base_system.py:153-189)."""
    from safetensors.torch import save_file
    from plangen_amd.engine import Engine
    from plangen_amd.weights import load_checkpoint
    keep = {k: v.contiguous() for k, v in tiny_weights.items()
            if not k.startswith(("vision_model.", "aligner.", "gen_vision_model.encoder", "gen_vision_model.quant_conv"))}
    save_file(keep, str(tmp_path / "model.safetensors"))
    bias = tiny_weights["gen_head.vision_head.bias"] + 1.5
    ck = tmp_path / "checkpoint-7"
    ck.mkdir()
    torch.save({"vl_gpt.gen_head.vision_head.bias": bias}, str(ck / "trainable_model_parameters.pth"))
    e = Engine(tiny_cfg, dtype="f32", max_rows=8, max_prompt=16, max_images=1, with_lm_head=True)
    info = load_checkpoint(e, str(tmp_path), overlay=str(ck))
    assert info["skipped"] == []
    g = load_golden("sample_image_tiny.npz")
    h = torch.from_numpy(g["last_hidden"][0])
    W2 = dict(tiny_weights); W2["gen_head.vision_head.bias"] = bias
    assert (e.gen_head(h).cpu() - R.gen_head(W2, h)).abs().max() < 1e-4
    e.close()


def test_text_greedy_bf16_vs_oracle_logits(tiny_cfg, tiny_weights, ocfg):
    """bf16 check for pg_generate_text_greedy.  The engine's own greedy ids are forced into the fp32
    oracle; at every step the engine's token must be the oracle's argmax or lie within TEXT_TOL of it (bf16 rounding of
    O(1) lm_head logits), and most steps agree outright."""
    from plangen_amd.system import System
    TEXT_TOL = 0.08
    g = load_golden("generate_tiny.npz")
    e = get_engine(tiny_cfg, tiny_weights, "bf16")
    sysm = System(tiny_cfg, e)
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
    emb = sysm.vl_gpt.language_model.get_input_embeddings()(ids.to(e.device))
    n = 12
    out = sysm.vl_gpt.language_model.generate(inputs_embeds=emb, attention_mask=mask.to(e.device), eos_token_id=tiny_cfg.eos_id,
                                              max_new_tokens=n, min_new_tokens=n).cpu()           # EOS suppressed: n steps for every row
    assert out.shape == (ids.shape[0], n)
    ref_ids, logits = R.generate_text_greedy(tiny_weights, ocfg, R.embed_tokens(tiny_weights, ids), mask, n, tiny_cfg.eos_id,
                                             min_new_tokens=n, force_tokens=out, return_logits=True)     # [n, B, V]
    lg = logits.permute(1, 0, 2)                                                                    # [B, n, V]
    lg[:, :, tiny_cfg.eos_id] = float("-inf")
    gap = lg.max(-1).values - torch.gather(lg, 2, out[..., None]).squeeze(-1)                       # 0 where the engine picked the oracle argmax
    assert gap.max().item() < TEXT_TOL, gap
    agree = (gap == 0).float().mean().item()
    print(f"bf16 text greedy: argmax agreement {agree:.3f}, worst logit gap {gap.max().item():.4f}")
    assert agree > 0.9


def test_vq_encode_bf16_indices_near_ties_only(tiny_cfg, tiny_weights, ocfg):
    """VERDICT r1 item 8 / missing 7: the reference encodes gt_image.bfloat16() under autocast (plangen_base.py:530-532).
    bf16 engine indices vs the fp32 oracle: every mismatch must be a near tie of the oracle's own code distances."""
    g = load_golden("vq_tiny.npz")
    e = get_engine(tiny_cfg, tiny_weights, "bf16")
    x = torch.from_numpy(g["enc_in"])
    idx = e.vq_encode(x.to(torch.bfloat16)).cpu()
    ref = torch.from_numpy(g["enc_idx"])
    h = R._conv(tiny_weights, "gen_vision_model.quant_conv", R.vq_encoder(tiny_weights, ocfg, x))
    z = torch.nn.functional.normalize(h.permute(0, 2, 3, 1).reshape(-1, ocfg.img_dim), dim=-1)
    emb = torch.nn.functional.normalize(tiny_weights["gen_vision_model.quantize.embedding.weight"], dim=-1)
    d = (z ** 2).sum(1, keepdim=True) + (emb ** 2).sum(1) - 2 * z @ emb.t()
    gap = torch.gather(d, 1, idx[:, None]).squeeze(1) - d.min(1).values
    agree = (idx == ref).float().mean().item()
    print(f"bf16 VQ encode: index agreement {agree:.3f}, worst distance gap of a mismatch {gap.max().item():.4f} (distances span {d.min().item():.2f}..{d.max().item():.2f})")
    assert gap.max().item() < 0.05 and agree > 0.7
