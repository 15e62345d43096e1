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
    """python train.py --cfg ... --opt test=True ... -> System.validation (tiny config, synthetic prompts, bf16 default);
    the per-task dispatch and the output tree are covered in tests/test_gpu_cli.py."""
    import train
    out = train.parse_args(["--cfg", "project/plangen/cfg/uni/h_text_ump+oimsam.py", "--opt", "test=True", "tiny=True",
                            "resume=None", f"out_path='{tmp_path}'", "test_batch_size=2", "max_test_len=2", "max_prompt=200",
                            "janus_path=None"])
    import importlib
    System = importlib.import_module(out.system_cls_path).System
    m = System(out, None)
    m.setup_data(None)
    m.resume(None)
    res = m.validation(0)
    assert res["images"] == 4 and res["task_type"] == "uni"
    files = os.listdir(os.path.join(res["out_dir"], "pr_image"))
    assert sorted(files) == ["0.png", "1.png", "2.png", "3.png"]
    m.engine.close()


def test_decode_graph_survives_fresh_buffers_seeds_and_temperatures(tiny_cfg, tiny_weights, ocfg):
    """the decode-step graph is keyed on shapes only; seeds, temperatures, T and the caller's
    output / forcing tensors reach it through library-owned device memory.  Calls with fresh tensors must keep
    producing the right tokens (greedy == oracle), seeded sampling must be reproducible, and seeds must matter."""
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    e.set_option("use_graph", 1)          # off by default since the end of round 2 (stream launches measure faster); this test is about the graph
    g = torch.Generator().manual_seed(71)
    cond = [torch.randint(8, tiny_cfg.vocab, (n,), generator=g).tolist() for n in (7, 11)]
    neg = torch.randint(8, tiny_cfg.vocab, (5,), generator=g).tolist()
    ids, mask = _collate(tiny_cfg, cond, neg)
    pad = _pad(mask, ids.shape[1])
    ref = R.sample_image(tiny_weights, ocfg, R.embed_tokens(tiny_weights, ids), mask, 5.0, n_tokens=10)
    outs = []
    for seed, temp in ((0, 0.0), (3, 1.0), (4, 1.0), (3, 1.0), (9, 0.0)):
        e.prefill(ids, pad)
        outs.append(e.decode_image_tokens(T=10, cfg_weight=5.0, temperature=temp, seed=seed).cpu())
    assert torch.equal(outs[0], ref) and torch.equal(outs[4], ref)
    assert torch.equal(outs[1], outs[3]) and not torch.equal(outs[1], outs[2])
    # forcing tensors change between calls (teacher forcing) on the same graph
    force = torch.randint(0, tiny_cfg.img_vocab, (2, 10), generator=g).int()
    ref_f = R.sample_image(tiny_weights, ocfg, R.embed_tokens(tiny_weights, ids), mask, 5.0, n_tokens=10, force_tokens=force)
    e.prefill(ids, pad)
    assert torch.equal(e.decode_image_tokens(T=10, cfg_weight=5.0, temperature=0.0, force_tokens=force).cpu(), ref_f)
    e.prefill(ids, pad)
    assert torch.equal(e.decode_image_tokens(T=7, cfg_weight=5.0, temperature=0.0).cpu(), ref[:, :7])
    e.set_option("use_graph", 0)


def test_sampled_frequencies_follow_softmax_chi_square(tiny_cfg, tiny_weights, ocfg):
    """chi-square test of sampled token frequencies against softmax(mixed logits / T) on the small
    vocabulary (V=256).  Teacher forcing fixes the context, so the distribution of step t of image b is known
    from the oracle; 600 seeds x 3 steps x 2 images, categories pooled into bins of expected count >= 20."""
    from scipy import stats
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    g = torch.Generator().manual_seed(72)
    cond = [torch.randint(8, tiny_cfg.vocab, (n,), generator=g).tolist() for n in (9, 6)]
    neg = torch.randint(8, tiny_cfg.vocab, (4,), generator=g).tolist()
    ids, mask = _collate(tiny_cfg, cond, neg)
    pad = _pad(mask, ids.shape[1])
    T, N, temp = 3, 600, 1.3
    force = torch.randint(0, tiny_cfg.img_vocab, (2, T), generator=g).int()
    _, logits = R.sample_image(tiny_weights, ocfg, R.embed_tokens(tiny_weights, ids), mask, 2.0, n_tokens=T,
                               force_tokens=force, return_logits=True)           # [T, B, V]
    draws = []
    for seed in range(N):
        e.prefill(ids, pad)
        draws.append(e.decode_image_tokens(T=T, cfg_weight=2.0, temperature=temp, seed=seed, force_tokens=force).cpu())
    draws = torch.stack(draws)                                                   # [N, B, T]
    p = torch.softmax(logits.double() / temp, dim=-1)                            # [T, B, V]
    for t in range(T):
        for b in range(2):
            exp = p[t, b] * N
            obs = torch.bincount(draws[:, b, t].long(), minlength=tiny_cfg.img_vocab).double()
            # pool categories (most probable first) into bins of expected count >= 20
            order = torch.argsort(exp, descending=True)
            o, x, co, cx = [], [], 0.0, 0.0
            for v in order.tolist():
                co += obs[v].item(); cx += exp[v].item()
                if cx >= 20:
                    o.append(co); x.append(cx); co = cx = 0.0
            if cx > 0:
                o[-1] += co; x[-1] += cx
            o, x = torch.tensor(o), torch.tensor(x)
            assert len(o) >= 8, len(o)
            chi2 = ((o - x) ** 2 / x).sum().item()
            pval = 1 - stats.chi2.cdf(chi2, df=len(o) - 1)
            assert pval > 1e-4, (t, b, chi2, pval)


def test_transposed_or_missing_weights_are_refused(tiny_cfg, tiny_weights):
    """a tensor with the right element count and the wrong shape is rejected; an engine with
    required tensors missing refuses to run unless the caller opted in with strict=False."""
    from plangen_amd.engine import Engine, PlanGenError
    e = Engine(tiny_cfg, dtype="f32", max_rows=2, max_prompt=8, max_images=1)
    name = "language_model.model.layers.0.mlp.down_proj.weight"
    with pytest.raises(PlanGenError, match="shape"):
        e.load_tensors({name: tiny_weights[name].t().contiguous()})
    part = {k: v for k, v in tiny_weights.items() if "layers.1." not in k}
    e.load_tensors(part)
    with pytest.raises(PlanGenError, match="missing"):
        e.finalize(strict=True)
    with pytest.raises(PlanGenError):
        e.prefill(torch.tensor([[9, 10], [9, 11]], dtype=torch.int32), [0, 0])
    assert e.finalize(strict=False) > 0                      # opt-in: missing tensors read as zeros
    e.prefill(torch.tensor([[9, 10], [9, 11]], dtype=torch.int32), [0, 0])
    toks = e.decode_image_tokens(T=2, cfg_weight=5.0, temperature=0.0)
    assert toks.shape == (1, 2)
    e.close()


def test_two_handles_keep_their_own_tuning(tiny_cfg, tiny_weights):
    """pg_set_option is per handle."""
    from plangen_amd.engine import Engine
    a = get_engine(tiny_cfg, tiny_weights, "bf16")
    b = Engine(tiny_cfg, dtype="bf16", max_rows=4, max_prompt=16, max_images=1)
    b.load_state_dict({k: v for k, v in tiny_weights.items()
                       if not k.startswith(("vision_model.", "aligner.", "gen_vision_model.encoder", "gen_vision_model.quant_conv", "language_model.lm_head"))})
    g = torch.Generator().manual_seed(5)
    x = torch.randn(100, 2048, generator=g).bfloat16()
    w = (torch.randn(1024, 2048, generator=g) * 0.05).bfloat16()
    b.set_option("split_target_big", 4096)                  # b wants many split-K slabs at M >= 96, a keeps its default
    S_a = a.op_gemm_splits(x, w)
    S_b = b.op_gemm_splits(x, w)
    assert S_b > S_a
    assert a.op_gemm_splits(x, w) == S_a
    b.close()


def test_bench_line_contract_tiny():
    """`bench.py` prints ONE JSON line with the driver's keys, the dominant-kernel roofline (HIP-event timed), the per-class
    table and the graph-replayed GEMM + norm phase (tiny config, a few seconds)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--tiny", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True and j["vs_baseline"] is None
    assert "workload" in j["config"] and j["value"] > 0
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert "decode_attention" in r["classes"] and "decode_gemm_qkv" in r["classes"]
    ph = r["decode_gemm_norm_phase"]
    assert ph["ms_per_step"] > 0 and ph["events_ms_per_step"] > 0 and "method" in ph


def test_bf16_flash_prefill_at_tile_boundaries(tiny_cfg, tiny_weights):
    """Both MFMA prefill-attention kernels (128-query LDS-DMA / transpose-read and 64-query) on prompt lengths that straddle their
    64-key and 64 / 128-query tiles (1, 2, 63, 64, 65, 127, 128, 129, 257 real tokens, left-padded): per-position hidden states agree
    with the fp32 engine within the bf16 bound and with each other."""
    L = 288
    lens = [1, 2, 63, 64, 65, 127, 129, 257]
    eb = get_engine(tiny_cfg, tiny_weights, "bf16", max_prompt=L)
    ef = get_engine(tiny_cfg, tiny_weights, "f32", max_prompt=L)
    g = torch.Generator().manual_seed(77)
    ids = torch.randint(8, tiny_cfg.vocab, (len(lens), L), generator=g).int()
    pad = [L - n for n in lens]
    ref = ef.prefill(ids, pad, position_mode=0, return_hidden=True).float().cpu()
    outs = {}
    for variant in (2, 1):
        eb.set_option("prefill_attn", variant)
        outs[variant] = eb.prefill(ids, pad, position_mode=0, return_hidden=True).float().cpu()
    eb.set_option("prefill_attn", 2)
    scale = ref.abs().max().item()
    for r, n in enumerate(lens):
        for variant in (2, 1):
            err = (outs[variant][r, L - n:] - ref[r, L - n:]).abs().max().item()
            assert err < 0.05 * scale, (variant, n, err, scale)
            assert outs[variant][r, :L - n].abs().max().item() == 0.0 if n < L else True
    assert (outs[2] - outs[1]).abs().max().item() < 0.03 * scale


def test_persistent_chain_skeleton_barriers_complete():
    """The measured skeleton of the persistent decode chain (chain.hip, profiles/r04_c): 256 workgroups, three hierarchical grid
    barriers per launch, run-ahead register-ring weight stream.  Every spin is bounded; no barrier may give up and the launch
    sequence must finish (a lost arrival would show as give-ups, not as a hang)."""
    import ctypes as C
    from conftest import ROOT
    from plangen_amd import _lib
    lib = _lib.load_diag()
    lib.pg_bench_chain_skeleton.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint)]
    for mode in (1, 3, 5):
        us, err = C.c_float(0), C.c_uint(0)
        assert lib.pg_bench_chain_skeleton(3, 60, mode, C.byref(us), C.byref(err)) == 0
        assert err.value == 0 and 0 < us.value < 200, (mode, us.value, err.value)
