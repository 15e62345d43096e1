"""GPU: the CLI row: `train.py --opt test=True test_data.task_type=...` dispatches
uni / uni_2stage / mmu / plan like plangen_base.py:1112-1127 and writes the reference's output tree (:1099-1105,
:1162-1181, :416-420); the stage-1 -> stage-2 hand-off through the tokenizer equals the oracle; t2i returns the
edit-region mask under teacher forcing (:557-560)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, get_engine
from oracle import ref_cpu as R
from plangen_amd import textproc as T

pytestmark = pytest.mark.gpu


def _args(tmp_path, task, **over):
    import train
    opts = ["test=True", "tiny=True", "test_batch_size=2", "max_test_len=2", "dtype='f32'", "temperature=0.0", f"out_path={str(tmp_path)!r}",
            f"test_data.task_type={task!r}", "max_new_tokens=12", "max_prompt=160"] + [f"{k}={v!r}" for k, v in over.items()]
    return train.parse_args(["--cfg", os.path.join(ROOT, "project/plangen/cfg/uni/h_text_ump+oimsam.py"), "--opt", *opts])


@pytest.mark.parametrize("task", ["uni", "uni_2stage", "mmu", "plan"])
def test_validation_dispatch_and_output_tree(tmp_path, task):
    from project.plangen.plangen_base import System
    a = _args(tmp_path, task)
    m = System(a, None)
    m.setup_data(None)
    m.resume(None)
    r = m.validation(0)
    base = os.path.join(str(tmp_path), "test", f"synthetic_{task}_2")
    assert r["out_dir"] == os.path.join(base, "0") and os.path.isdir(os.path.join(base, "0_batch"))
    for d in ("gt_image", "pr_image", "image_ids", "gt_image_ids"):
        assert os.path.isdir(os.path.join(base, "0", d))
    lay = [json.load(open(os.path.join(base, "0_batch", f"{i}_layout.json"))) for i in range(2)]
    assert all(set(l) == {"base_caption", "gt_grounding", "pr_grounding"} and len(l["base_caption"]) == 2 for l in lay)
    pngs = sorted(os.listdir(os.path.join(base, "0", "pr_image")))
    if task in ("uni", "uni_2stage"):
        assert pngs == ["0.png", "1.png", "2.png", "3.png"] and r["images"] == 4          # {idx*bs+i}.png
        from PIL import Image
        assert Image.open(os.path.join(base, "0", "pr_image", "0.png")).size == (32, 32)
    else:
        assert pngs == [] and r["images"] == 0                                              # pred_image=False
    if task == "uni":
        assert lay[0]["pr_grounding"] == ""                                                 # pred_layout=False (:418)
    else:
        assert all(isinstance(t, str) for t in lay[0]["pr_grounding"])
        if task != "mmu":
            assert all(t.startswith("<grounding>") and t.endswith("</grounding>") for t in lay[0]["pr_grounding"])
    m.engine.close()


def test_t2i_task_and_unknown_task_fail_like_the_reference(tmp_path):
    from plangen_amd.engine import PlanGenError
    from project.plangen.plangen_base import System
    a = _args(tmp_path, "t2i")
    m = System(a, None)
    m.setup_data(None)
    m.resume(None)
    with pytest.raises(PlanGenError, match="use_uni_prompt_in_t2i"):       # the reference hits `assert False` on this branch (:645-648)
        m.validation(0)
    m.engine.close()


def test_uni_2stage_through_the_tokenizer_equals_oracle(tiny_cfg, tiny_weights, ocfg):
    """Stage 1 greedy layout ids -> decode_plan_text_batch -> wrap_uni_prompt -> pad -> CFG collate -> image loop, on the
    engine vs the oracle (fp32: ids, prompts and image tokens identical)."""
    from plangen_amd.system import System, pad_input_ids
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    codec = T.TagWordCodec(tiny_cfg.vocab, eos_id=tiny_cfg.eos_id, pad_id=tiny_cfg.pad_id)
    sysm = System(tiny_cfg, e, codec=codec)
    sysm.args.temperature = 0.0
    caps = ["a red cat on the table", "two dogs"]
    s1 = [sysm.wrap_uni_prompt(c, "<grounding>", in_stage1=True)[1].tolist() for c in caps]
    ids1, mask1 = pad_input_ids(s1, tiny_cfg.pad_id)
    neg = sysm.wrap_uni_prompt("", "")[1].tolist()
    batch = dict(base_caption=caps, gt_grounding=["", ""], uni_stage1_inputs_ids=ids1, uni_stage1_attention_mask=mask1, neg_inputs_ids=neg)
    out = sysm.uni_generate(batch, pred_layout=True, max_new_tokens=10)
    # oracle: same steps with the restated rules
    ref_ids = R.generate_text_greedy(tiny_weights, ocfg, R.embed_tokens(tiny_weights, ids1), mask1, 10, tiny_cfg.eos_id)
    assert np.array_equal(out["pr_layout_ids"].cpu().numpy(), ref_ids.numpy())
    ref_gr = [R.decode_plan_text(codec.decode(r)) for r in ref_ids.tolist()]
    assert out["pr_grounding"] == ref_gr
    cond = [codec.encode(R.wrap_uni_prompt_text(c, g)) for c, g in zip(caps, ref_gr)]
    cids, cmask = R.t2i_infer_collate_batch(cond, neg, tiny_cfg.pad_id, tiny_cfg.img_tokens)
    ref_tok, ref_img = R.t2i(tiny_weights, ocfg, cids, cmask, 5.0)
    assert np.array_equal(out["pr_tokens"].cpu().numpy(), ref_tok.numpy())
    assert ((out["pr_image"].cpu() - ref_img) ** 2).mean().item() <= 1e-4


def test_mmu_answers_cut_at_eos(tiny_cfg, tiny_weights):
    from plangen_amd.system import System
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    codec = T.TagWordCodec(tiny_cfg.vocab, eos_id=tiny_cfg.eos_id, pad_id=tiny_cfg.pad_id)
    sysm = System(tiny_cfg, e, codec=codec)
    a, b = codec.encode("a cat")[1:], codec.encode("two dogs")[1:]
    rows = [a + [tiny_cfg.eos_id] + b, b]
    assert sysm.decode_mmu_text_batch(rows) == ["a cat", "two dogs"]          # cut at the FIRST eos (:316-322)
    assert sysm.trans_gr_to_creati("<ref>x</ref><box>[0,0,500,1000]</box>") == ([[0.0, 0.0, 0.5, 1.0]], ["x"])


def test_t2i_returns_edit_mask_under_teacher_forcing(tiny_cfg, tiny_weights):
    """use_teacher_forcing (plangen_base.py:528-532, :557-560): gt image -> VQ encode -> forced tokens outside the edit
    region; second return value = the region map resized to janus_hw."""
    from plangen_amd.system import System, t2i_infer_collate_batch
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    sysm = System(tiny_cfg, e)
    sysm.args.use_teacher_forcing = True
    sysm.args.temperature = 0.0
    g = torch.Generator().manual_seed(3)
    cond = [torch.randint(8, tiny_cfg.vocab, (n,), generator=g).tolist() for n in (6, 9)]
    neg = torch.randint(8, tiny_cfg.vocab, (4,), generator=g).tolist()
    ids, mask = t2i_infer_collate_batch(cond, neg, tiny_cfg.pad_id, tiny_cfg.img_tokens)
    gt = torch.rand(2, 3, tiny_cfg.img_size, tiny_cfg.img_size, generator=g) * 2 - 1
    region = (torch.rand(2, tiny_cfg.img_tokens, generator=g) > 0.5).int()
    dec, mask_image = sysm.t2i(ids, mask, gt_image=gt, edit_region=region)
    assert dec.shape == (2, 3, tiny_cfg.img_size, tiny_cfg.img_size)
    ref = torch.nn.functional.interpolate(region.reshape(2, 1, tiny_cfg.grid, tiny_cfg.grid).repeat(1, 3, 1, 1).float(),
                                          size=(tiny_cfg.img_size,) * 2, mode="bilinear", align_corners=False, antialias=True)
    assert mask_image.shape == dec.shape and torch.allclose(mask_image.cpu(), ref, atol=1e-6)
    # outside the edit region the emitted tokens are the ground-truth labels
    labels = e.vq_encode(gt).reshape(2, -1).cpu()
    toks = sysm.last_generated_tokens.cpu()
    assert torch.equal(toks[region == 0], labels[region == 0].int())


def test_teacher_forcing_with_parallel_size_forces_only_the_first_replica(tiny_cfg, tiny_weights):
    """The reference's forcing loop runs over ``len(batch['edit_region'])`` = the B un-replicated rows (plangen_base.py:593-598), so
    with parallel_size = p only the first replica of every image is teacher-forced; replicas 2..p sample freely.  Matched (round 4):
    replica 0 carries the ground-truth labels outside the edit region, replica 1 equals an UNFORCED run of the same prompts, and the
    returned mask_image keeps B rows (:557-560)."""
    from plangen_amd.system import System, t2i_infer_collate_batch
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    sysm = System(tiny_cfg, e)
    sysm.args.temperature, sysm.args.parallel_size = 0.0, 2
    g = torch.Generator().manual_seed(5)
    cond = [torch.randint(8, tiny_cfg.vocab, (n,), generator=g).tolist() for n in (6, 9)]
    neg = torch.randint(8, tiny_cfg.vocab, (4,), generator=g).tolist()
    ids, mask = t2i_infer_collate_batch(cond, neg, tiny_cfg.pad_id, tiny_cfg.img_tokens)
    gt = torch.rand(2, 3, tiny_cfg.img_size, tiny_cfg.img_size, generator=g) * 2 - 1
    region = (torch.rand(2, tiny_cfg.img_tokens, generator=g) > 0.5).int()
    sysm.args.use_teacher_forcing = False
    sysm.t2i(ids, mask)
    free = sysm.last_generated_tokens.cpu().clone()                    # [4, T]: replicas laid out [all images] x p (:547)
    assert torch.equal(free[:2], free[2:])
    sysm.args.use_teacher_forcing = True
    dec, mask_image = sysm.t2i(ids, mask, gt_image=gt, edit_region=region)
    toks = sysm.last_generated_tokens.cpu()
    labels = e.vq_encode(gt).reshape(2, -1).cpu().int()
    assert dec.shape[0] == 4 and mask_image.shape[0] == 2
    assert torch.equal(toks[:2][region == 0], labels[region == 0])     # replica 0: forced outside the edit region
    assert torch.equal(toks[2:], free[2:])                             # replica 1: not forced at all, as in the reference
    assert not torch.equal(toks[:2], toks[2:])
    sysm.args.use_teacher_forcing = False
    sysm.args.parallel_size = 1


def test_parallel_size_replica_layout_and_file_names(tmp_path, tiny_cfg, tiny_weights, ocfg):
    """parallel_size=2 (plangen_base.py:547, :1171-1176): t2i replicates the CFG batch as ``torch.cat([tokens] * p)`` -- replicas
    laid out [all pairs] x p -- and validation saves ``pr_image[i*p + t]`` as ``pr_image/{idx*bs+i}_{t}.png``.  Greedy: every
    replica of a sample equals the oracle's image of that sample; the files carry the reference's names and indexing."""
    from plangen_amd.system import System, t2i_infer_collate_batch
    from project.plangen.plangen_base import System as CliSystem
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    sysm = System(tiny_cfg, e)
    sysm.args.temperature, sysm.args.parallel_size = 0.0, 2
    g = torch.Generator().manual_seed(5)
    cond = [torch.randint(8, tiny_cfg.vocab, (n,), generator=g).tolist() for n in (7, 5)]
    neg = torch.randint(8, tiny_cfg.vocab, (4,), generator=g).tolist()
    ids, mask = t2i_infer_collate_batch(cond, neg, tiny_cfg.pad_id, tiny_cfg.img_tokens)
    dec, _ = sysm.t2i(ids, mask)
    ref_tok, ref_img = R.t2i(tiny_weights, ocfg, ids, mask, 5.0)
    toks = sysm.last_generated_tokens.cpu()
    assert dec.shape[0] == 4 and toks.shape[0] == 4
    assert torch.equal(toks[:2], ref_tok) and torch.equal(toks[2:], ref_tok)            # [sample 0, sample 1] x 2 replicas (:547)
    assert ((dec[:2].cpu() - ref_img) ** 2).mean().item() <= 1e-4 and torch.equal(dec[:2], dec[2:])
    # through the CLI: names {idx*bs+i}_{t}.png, content pr_image[i*p+t] (the reference's own indexing of that layout)
    a = _args(tmp_path, "uni", parallel_size=2)
    m = CliSystem(a, None)
    m.setup_data(None)
    m.resume(None)
    r = m.validation(0)
    pngs = sorted(os.listdir(os.path.join(r["out_dir"], "pr_image")))
    assert pngs == sorted(f"{k}_{t}.png" for k in range(4) for t in range(2))
    m.engine.close()


def test_use_neg_box_builds_one_negative_prompt_per_sample(tiny_cfg, tiny_weights, ocfg):
    """use_neg_box (plangen_base.py:652-670): the uncond row of sample i is wrap_uni_prompt(neg_base_caption[i],
    neg_gt_grounding[i]); image tokens equal the oracle driven with the same per-sample negative prompts, and differ from the
    shared-negative-prompt run."""
    from plangen_amd.system import System, pad_input_ids
    e = get_engine(tiny_cfg, tiny_weights, "f32")
    codec = T.TagWordCodec(tiny_cfg.vocab, eos_id=tiny_cfg.eos_id, pad_id=tiny_cfg.pad_id)
    sysm = System(tiny_cfg, e, codec=codec)
    sysm.args.temperature, sysm.args.use_neg_box = 0.0, True
    caps, grs = ["a red cat", "two dogs on a field"], ["<grounding><ref>cat</ref><box>[1,2,300,400]</box></grounding>", "<grounding></grounding>"]
    ncaps, ngrs = ["blurry", "low quality photo"], ["<grounding><ref>cat</ref><box>[5,5,900,900]</box></grounding>", ""]
    uni = [sysm.wrap_uni_prompt(c, g)[1].tolist() for c, g in zip(caps, grs)]
    ids, m = pad_input_ids(uni, tiny_cfg.pad_id)
    T_ = tiny_cfg.img_tokens
    batch = dict(base_caption=caps, gt_grounding=grs, neg_base_caption=ncaps, neg_gt_grounding=ngrs, uni_inputs_ids=ids,
                 uni_attention_mask=torch.cat([m, torch.ones((2, T_), dtype=m.dtype)], -1))
    out = sysm.uni_generate(batch, pred_layout=False)
    negs = [codec.encode(R.wrap_uni_prompt_text(c, g)) for c, g in zip(ncaps, ngrs)]
    cids, cmask = R.t2i_infer_collate_batch(uni, negs, tiny_cfg.pad_id, T_)
    assert len(negs[0]) != len(negs[1])                                        # genuinely per-sample rows
    ref_tok, _ = R.t2i(tiny_weights, ocfg, cids, cmask, 5.0)
    assert np.array_equal(out["pr_tokens"].cpu().numpy(), ref_tok.numpy())
    sysm.args.use_neg_box = False
    shared = sysm.uni_generate(dict(batch, neg_inputs_ids=sysm.wrap_uni_prompt("", "")[1].tolist()), pred_layout=False)
    assert not torch.equal(shared["pr_tokens"], out["pr_tokens"])
    sysm.args.use_neg_box = True
    with pytest.raises(Exception, match="use_neg_box"):
        sysm.uni_generate({k: v for k, v in batch.items() if not k.startswith("neg_")}, pred_layout=False)


def test_edited_image_and_gt_image_are_written(tmp_path, tiny_cfg):
    """plangen_base.py:1167-1181: gt_image/{idx*bs+i}.png and edited_image/{idx*bs+i}.png next to pr_image when the batch has them."""
    S = tiny_cfg.img_size
    g = torch.Generator().manual_seed(9)
    rows = []
    for i in range(4):
        pi, pe = str(tmp_path / f"img{i}.pt"), str(tmp_path / f"ed{i}.pt")
        torch.save(torch.rand(3, S, S, generator=g) * 2 - 1, pi)
        torch.save(torch.rand(3, S, S, generator=g) * 2 - 1, pe)
        rows.append({"cond_ids": torch.randint(8, tiny_cfg.vocab, (5 + i,), generator=g).tolist(), "neg_ids": [1, 9, 10],
                     "image_id": f"id{i}" if i % 2 == 0 else "", "image_pt": pi, "edited_image_pt": pe})
    f = tmp_path / "rows.jsonl"
    f.write_text("\n".join(json.dumps(r) for r in rows))
    from project.plangen.plangen_base import System as CliSystem
    a = _args(tmp_path, "uni", synthetic=True)
    a.test_data = dict(a.test_data, data_file=str(f), data_name="edit")
    m = CliSystem(a, None)
    m.setup_data(None)
    m.resume(None)
    r = m.validation(0)
    for d in ("pr_image", "gt_image", "edited_image"):
        assert sorted(os.listdir(os.path.join(r["out_dir"], d))) == ["0.png", "1.png", "2.png", "3.png"], d
    assert sorted(os.listdir(os.path.join(r["out_dir"], "image_ids"))) == ["id0.jpg", "id2.jpg"]
    assert sorted(os.listdir(os.path.join(r["out_dir"], "gt_image_ids"))) == ["id0.jpg", "id2.jpg"]
    m.engine.close()


def test_mmu_prompt_rows_use_the_chat_template(tmp_path):
    """mmu rows with text go through wrap_mmu_prompt's conversation + image-token expansion."""
    from project.plangen.plangen_base import System as CliSystem
    a = _args(tmp_path, "mmu")
    m = CliSystem(a, None)
    dl = m.setup_data(None)
    pin = dl[0]["prepare_inputs_infer"]
    c = m.codec
    boi, img, eoi = c.token_id(T.IMAGE_START_TAG), c.token_id(T.IMAGE_TAG), c.token_id(T.IMAGE_END_TAG)
    P = m.cfg.vit_tokens
    for i in range(pin["input_ids"].shape[0]):
        row = pin["input_ids"][i][pin["attention_mask"][i].bool()].tolist()
        k = row.index(boi)
        assert row[k + 1:k + 1 + P] == [img] * P and row[k + 1 + P] == eoi
        assert pin["images_seq_mask"][i][pin["attention_mask"][i].bool()].tolist() == [t == img for t in row]
        text = c.decode(row[:k])
        assert text.startswith(T.MMU_SYSTEM_PROMPT) and text.endswith("<|User|>: ")
        assert c.decode(row[k + 2 + P:]).endswith("<|Assistant|>:")
    m.resume(None)
    r = m.validation(0)
    assert r["batches"] == 2
    m.engine.close()


def test_cli_end_to_end_on_real_format_files(tmp_path, tiny_cfg, tiny_weights):
    """The command line a PlanGen user runs, on files in the reference's formats: `janus_path` holding HF safetensors weights AND
    HF tokenizer files (HFCodec), `resume=<int>` -> `out_path/checkpoint-<int>/trainable_model_parameters.pth` overlay
    (base_system.py:132-134, :153-155), JSONL TEXT rows (base_caption / gt_grounding / image_id), task uni_2stage: stage-1 layout
    decode -> decode_plan_text_batch -> wrap_uni_prompt through the tokenizer -> CFG image decode -> PNG tree + layout JSON."""
    from safetensors.torch import save_file
    from test_text_cpu import _tiny_hf_tokenizer_dir
    from project.plangen.plangen_base import System as CliSystem
    jp = _tiny_hf_tokenizer_dir(tmp_path)                                        # tokenizer.json etc. (vocabulary < tiny_cfg.vocab)
    keep = {k: v.contiguous() for k, v in tiny_weights.items()
            if not k.startswith(("vision_model.", "aligner.", "gen_vision_model.encoder", "gen_vision_model.quant_conv"))}
    save_file(keep, os.path.join(jp, "model.safetensors"))
    out = tmp_path / "out"
    ck = out / "checkpoint-3"
    ck.mkdir(parents=True)
    torch.save({"vl_gpt.gen_head.vision_head.bias": tiny_weights["gen_head.vision_head.bias"] + 0.25}, str(ck / "trainable_model_parameters.pth"))
    rows = [{"base_caption": c, "gt_grounding": "<grounding><ref>a cat</ref><box>[1,2,300,400]</box></grounding>", "image_id": f"im{i}"}
            for i, c in enumerate(["a red cat on the table", "two dogs playing in a field", "a cat", "the table"])]
    f = tmp_path / "rows.jsonl"
    f.write_text("\n".join(json.dumps(r) for r in rows))
    a = _args(out, "uni_2stage", janus_path=jp, resume=3)
    a.test_data = dict(a.test_data, data_file=str(f), data_name="creati")
    m = CliSystem(a, None)
    assert type(m.codec).__name__ == "HFCodec" and m.cfg.eos_id == m.codec.eos_token_id and m.cfg.pad_id == m.codec.pad_id
    m.setup_data(None)
    assert m.resume(None) == 0
    r = m.validation(0)
    base = os.path.join(str(out), "test", "creati_uni_2stage_2", "0")
    assert r["out_dir"] == base and r["images"] == 4
    assert sorted(os.listdir(os.path.join(base, "pr_image"))) == ["0.png", "1.png", "2.png", "3.png"]
    assert sorted(os.listdir(os.path.join(base, "image_ids"))) == [f"im{i}.jpg" for i in range(4)]
    lay = json.load(open(os.path.join(str(out), "test", "creati_uni_2stage_2", "0_batch", "0_layout.json")))
    assert lay["base_caption"] == [rows[0]["base_caption"], rows[1]["base_caption"]]
    assert all(t.startswith("<grounding>") and t.endswith("</grounding>") for t in lay["pr_grounding"])
    m.engine.close()
