"""GPU: the SMALL-BATCH kernel instantiations (BASELINE configs[1] bs=8, the bs=16/32 per-rank batches of configs[2]/[3])
against the transformers-driven fixtures, through engines with DEFAULT options.

The dispatch depends on the row count M = 2 x images (launch_attn_decode_fused / the decode GEMM dispatch in engine.hip):

  * M x heads <= 512 (bs <= 16): attn_decode_fused_kernel<bf16,5,8,16> -- 8 waves x 5 x 4 = 160 keys per block iteration; the
    first 160 keys of a segment go through the peeled chunk, the software-pipelined loop (run_pipe) only starts beyond them,
    and the shared-prefix loop only beyond 160 SHARED keys;
  * bs = 32: attn_decode_fused_kernel<bf16,6,4,16> (96 keys per iteration) on 64 rows;
  * the 16- / 32- / 64-row decode GEMM blocks (gemm_sk4_kernel<1|2|4, NCK, ...>), rmsnorm512_kernel, the M <= 16 sampler.

Two references:

  1. pairs [:8] / [:16] / [:32] of tests/golden/sample_image_fullwidth.npz (rows are independent, so a slice of the 64-pair
     fixture IS the reference of the smaller batch): 48 steps, cond rows 160-256 + 48 keys, shared 96-token negative prompt;
  2. tests/golden/sample_image_b8_long.npz (oracle/make_golden.py::golden_small_batch): 8 pairs, L = 256, ONE shared
     200-token negative prompt, T = 288 greedy steps -> cond rows reach 448-544 keys (peeled chunk + 2-3 run_pipe iterations),
     uncond rows 200 shared keys (peeled chunk + one iteration of the shared-prefix loop for waves 0-1) + 288 private keys
     (2 run_pipe iterations).

Tolerances are the full-width ones (tests/test_gpu_fullwidth.py): PG_F32 tokens bit-exact free-running, logits within 3e-3;
PG_BF16 teacher-forced: E_hip <= K x E_ref per statistic, E_ref = the reference's own autocast-bf16 error on the same images / steps (tests/bf16ref.py).
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from test_gpu_fullwidth import (FULLW, LOGIT_TOL_BF16_MAX, LOGIT_TOL_F32, _setup)

pytestmark = pytest.mark.gpu

_E = {}


def _engine(dtype, max_rows):
    from plangen_amd.engine import Engine
    s = _setup()
    key = (dtype, max_rows)
    if key not in _E:
        for k in [k for k in _E if isinstance(k, tuple)]:          # one full-width engine at a time
            _E.pop(k).close()
        e = Engine(s["cfg"], dtype=dtype, max_rows=max_rows, max_prompt=256, max_new=s["cfg"].img_tokens, max_images=1)
        e.load_state_dict(s["W"])
        _E[key] = e
    return _E[key]


def _bf16_stats(logits, toks, g, sl, T, name="sample_image_fullwidth"):
    """Teacher-forced bf16 statistics against fixture g restricted to images sl; accepted relative to the REFERENCE'S OWN bf16 arithmetic on the same
    images and steps (tests/bf16ref.py, tests/golden/<name>_bf16ref.npz)."""
    import bf16ref
    bf16ref.check_image_loop(name, logits, toks, g, f"bf16 {name} images {sl.start}:{sl.stop}, {T} steps", images=sl, steps=T)
    vsel = torch.from_numpy(g["vsel"]).long().to(logits.device)
    d = (logits[:, :, vsel].cpu() - torch.from_numpy(g["sel_logits"][:T, sl])).abs()
    top_v = torch.from_numpy(g["top_v"][:T, sl])
    margin = top_v[..., 0] - top_v[..., 1]
    gold = torch.from_numpy(g["tokens"][sl, :T])
    got = toks.cpu().t()
    agree = got == gold.t()
    decisive = margin > 2 * LOGIT_TOL_BF16_MAX
    flips = (~agree) & (margin > 2 * float(d.amax(dim=(1, 2)).max()))
    stats = {"steps": T, "images": int(gold.shape[0]), "logit_abs_err_max": float(d.max()),
             **{f"p{q}": float(np.percentile(d.numpy(), q)) for q in (50, 99, 99.9)},
             "err_max_last_32_steps": float(d[-32:].max()),
             "teacher_forced_agreement": float(agree.float().mean()), "decisive_share": float(decisive.float().mean())}
    assert stats["logit_abs_err_max"] < LOGIT_TOL_BF16_MAX, stats
    assert torch.equal(got[decisive], gold.t()[decisive]), stats
    assert not flips.any(), stats
    return stats


@pytest.mark.parametrize("pairs", [8, 16, 32])
def test_fixture_slices_f32_tokens_bit_exact(pairs):
    """Free-running greedy loop on the first `pairs` CFG pairs of the 64-pair fixture, engine sized for exactly that batch."""
    g = _setup()["g"]
    e = _engine("f32", 2 * pairs)
    ids = torch.from_numpy(g["ids"][:2 * pairs].astype(np.int32))
    pad = [int(p) for p in g["pad"][:2 * pairs]]
    e.prefill(ids, pad, position_mode=0)
    toks, logits = e.decode_image_tokens(T=48, cfg_weight=5.0, temperature=0.0, return_logits=True)
    assert np.array_equal(toks.cpu().numpy(), g["tokens"][:pairs])
    vsel = torch.from_numpy(g["vsel"]).long().to(logits.device)
    err = (logits[:, :, vsel].cpu() - torch.from_numpy(g["sel_logits"][:, :pairs])).abs().max().item()
    assert err < LOGIT_TOL_F32, err
    tv, ti = logits.topk(4, dim=-1)
    assert np.array_equal(ti[..., 0].cpu().numpy(), g["top_i"][:, :pairs, 0])


@pytest.mark.parametrize("pairs", [8, 16, 32])
def test_fixture_slices_bf16_teacher_forced(pairs):
    g = _setup()["g"]
    e = _engine("bf16", 2 * pairs)
    ids = torch.from_numpy(g["ids"][:2 * pairs].astype(np.int32))
    pad = [int(p) for p in g["pad"][:2 * pairs]]
    gold = torch.from_numpy(g["tokens"][:pairs])
    e.prefill(ids, pad, position_mode=0)
    toks, logits = e.decode_image_tokens(T=48, cfg_weight=5.0, temperature=0.0, force_tokens=gold.contiguous(), return_logits=True)
    stats = _bf16_stats(logits, toks, g, slice(0, pairs), 48)
    print(f"bf16 teacher-forced, {pairs} pairs:", json.dumps(stats))


def _long():
    if "long" not in _E:
        g = dict(load_golden("sample_image_b8_long.npz"))
        assert abs(float(g["wsum"]) - float(_setup()["g"]["wsum"])) < 1e-6 * float(g["wsum"])     # same seeded weights
        _E["long"] = g
    return _E["long"]


def test_long_fixture_key_counts_iterate_the_small_batch_loops():
    """The fixture's shape is what makes this file worth its GPU minutes: check it (pure host arithmetic)."""
    g = _long()
    pad, T = g["pad"], g["tokens"].shape[1]
    real = 256 - pad
    assert g["ids"].shape == (16, 256) and T >= 256
    assert int(g["neg_len"]) > 160 and (real[1::2] == int(g["neg_len"])).all()          # shared prefix beyond the peeled 160 keys
    assert (real[0::2] + T - 1 > 160 * 2).all()                                          # every cond row: peel + >= 2 pipelined iterations
    assert (real[0::2] + T - 1).max() > 160 * 3                                          # the longest: >= 3
    assert T - 1 >= 160 * 1 + 1                                                          # uncond private stream iterates


def test_long_fixture_f32_tokens_bit_exact():
    """288 free-running greedy steps x 8 images on a 16-row engine: every token equals the transformers-driven loop's."""
    g = _long()
    e = _engine("f32", 16)
    ids = torch.from_numpy(g["ids"].astype(np.int32))
    pad = [int(p) for p in g["pad"]]
    T = g["tokens"].shape[1]
    e.prefill(ids, pad, position_mode=0)
    toks, logits = e.decode_image_tokens(T=T, cfg_weight=5.0, temperature=0.0, return_logits=True)
    got = toks.cpu().numpy()
    if not np.array_equal(got, g["tokens"]):
        bad = np.argwhere(got != g["tokens"])
        b, t = bad[bad[:, 1].argmin()]
        margin = float(g["top_v"][t, b, 0] - g["top_v"][t, b, 1])
        raise AssertionError(f"first divergence image {b} step {t}, reference top-1 margin {margin:.2e}")
    vsel = torch.from_numpy(g["vsel"]).long().to(logits.device)
    err = (logits[:, :, vsel].cpu() - torch.from_numpy(g["sel_logits"])).abs().max().item()
    assert err < LOGIT_TOL_F32, err
    tv, ti = logits.topk(4, dim=-1)
    assert (tv.cpu() - torch.from_numpy(g["top_v"])).abs().max().item() < LOGIT_TOL_F32


@pytest.mark.parametrize("graph", [0, 1])
def test_long_fixture_bf16_teacher_forced(graph):
    """attn_decode_fused_kernel<bf16,5,8,16> with its pipelined private loop AND its shared-prefix loop iterating, gemm_sk4 16-row
    blocks, rmsnorm512, the M <= 16 sampler: 288 teacher-forced steps vs the fixture (stream launches and graph replay)."""
    g = _long()
    e = _engine("bf16", 16)
    ids = torch.from_numpy(g["ids"].astype(np.int32))
    pad = [int(p) for p in g["pad"]]
    T = g["tokens"].shape[1]
    gold = torch.from_numpy(g["tokens"])
    e.set_option("use_graph", graph)
    try:
        e.prefill(ids, pad, position_mode=0)
        toks, logits = e.decode_image_tokens(T=T, cfg_weight=5.0, temperature=0.0, force_tokens=gold.contiguous(), return_logits=True)
    finally:
        e.set_option("use_graph", 0)
    stats = _bf16_stats(logits, toks, g, slice(0, 8), T, name="sample_image_b8_long")
    print(f"bf16 long small-batch fixture (graph={graph}):", json.dumps(stats))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir) and not graph:
        json.dump(stats, open(os.path.join(out_dir, "smallbatch_bf16_stats.json"), "w"), indent=1)


def test_long_fixture_bf16_shared_prefix_equals_private_copies():
    """Shared-prefix branch of the 8-wave kernel (200 shared keys: peeled chunk + the non-pipelined shared loop) vs private
    copies of the negative prompt: same tokens under teacher forcing, logits equal up to the key-split summation order."""
    g = _long()
    e = _engine("bf16", 16)
    ids = torch.from_numpy(g["ids"].astype(np.int32))
    pad = [int(p) for p in g["pad"]]
    gold = torch.from_numpy(g["tokens"][:, :64]).contiguous()
    outs = []
    for share in (1, 0):
        e.set_option("share_uncond", share)
        e.prefill(ids, pad, position_mode=0)
        outs.append(e.decode_image_tokens(T=64, cfg_weight=5.0, temperature=0.0, force_tokens=gold, return_logits=True))
    e.set_option("share_uncond", 1)
    assert (outs[0][1] - outs[1][1]).abs().max().item() < 0.05
    assert (outs[0][0] == outs[1][0]).float().mean().item() > 0.97


def test_smallbatch_release():
    for k in list(_E):
        v = _E.pop(k)
        if hasattr(v, "close"):
            v.close()
