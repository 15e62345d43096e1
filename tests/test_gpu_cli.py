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
