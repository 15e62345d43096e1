"""CPU: the text plumbing either side of the hot path: chat template vs the fixture generated
by the reference's own conversation.py, the layout / answer post-processing rules vs the oracle restatement
(plangen_base.py:296-325, :460-473), the offline codec, and the CLI driver's error behaviour."""
import json
import os
from types import SimpleNamespace

import pytest
import torch

from conftest import GOLDEN
from oracle import ref_cpu as R
from plangen_amd import textproc as T


def test_chat_template_matches_reference_conversation_py():
    cases = json.load(open(os.path.join(GOLDEN, "text_golden.json")))
    assert len(cases) >= 5
    for c in cases:
        if "mmu_question" in c:
            continue
        assert T.wrap_uni_prompt_text(c["caption"], c["grounding"], c["in_stage1"]) == c["prompt"], c
    assert T.wrap_t2i_prompt_text("two dogs playing") == "<|User|>: two dogs playing\n\n<|Assistant|>:<begin_of_image>"


@pytest.mark.parametrize("text", [
    "<ref>a cat</ref><box>[12,40,500,620]</box></grounding> trailing <ref>x</ref>",
    "</grounding></grounding>", "no closing tag at all", "", "<ref>a</ref><box>[1,2,3,4]</box><ref>b c</ref><box>[10,20,30,40]</box></grounding>",
])
def test_plan_text_rules_match_oracle(text):
    assert T.cut_plan_text(text) == R.decode_plan_text(text)
    full = T.cut_plan_text(text)
    assert T.trans_gr_to_creati(full) == R.trans_gr_to_creati(full)
    assert T.get_pr_grounding_part("junk " + full).startswith("<grounding>")


def test_box_regex_values():
    boxes, names = T.trans_gr_to_creati("<grounding><ref>a red car</ref><box>[120,400,530,880]</box><ref>tree</ref><box>[0,0,1000,1000]</box></grounding>")
    assert names == ["a red car", "tree"] and boxes == [[0.12, 0.4, 0.53, 0.88], [0.0, 0.0, 1.0, 1.0]]
    assert T.trans_gr_to_creati("<grounding></grounding>") == ([], [])
    with pytest.raises(ValueError):
        T.trans_gr_to_creati("<ref>a</ref><box>[1,2,3]</box>")          # the reference's map(int, ...) unpack fails the same way


def test_cut_at_eos_matches_oracle():
    for ids in ([5, 9, 7, 4, 7], [7], [1, 2, 3], []):
        assert T.cut_at_eos(ids, 7) == R.cut_mmu_ids(ids, 7)


def test_tagword_codec_round_trip_and_stage1_drop():
    c = T.TagWordCodec(512)
    gr = "<grounding><ref>a small red cat</ref><box>[12,40,500,620]</box></grounding>"
    prompt, ids = T.wrap_uni_prompt_ids(c, "a cat on the table", gr)
    assert c.decode(ids) == prompt and ids[0] == c.bos_token_id and max(ids) < 512
    p1, ids1 = T.wrap_uni_prompt_ids(c, "a cat on the table", "<grounding>", in_stage1=True)
    assert p1.endswith("<grounding><｜end▁of▁sentence｜>") and c.decode(ids1).endswith("<grounding>")      # last token (EOS tag) dropped
    # ids of a layout survive decode -> cut -> parse
    layout_ids = c.encode("<ref>a dog</ref><box>[1,2,3,4]</box></grounding> and more")[1:]
    text = T.cut_plan_text(c.decode(layout_ids))
    assert T.trans_gr_to_creati(text) == ([[0.001, 0.002, 0.003, 0.004]], ["a dog"])
    with pytest.raises(ValueError):
        small = T.TagWordCodec(40)
        small.encode("many different words exceed this tiny vocabulary quickly indeed")


def _cli(**over):
    import train
    from conftest import ROOT
    opts = ["test=True", "tiny=True", "test_batch_size=2", "max_test_len=1"] + [f"{k}={v!r}" for k, v in over.items()]
    return train.parse_args(["--cfg", os.path.join(ROOT, "project/plangen/cfg/uni/h_text_ump+oimsam.py"), "--opt", *opts])


def test_cli_rejects_unknown_task_before_touching_the_gpu():
    from plangen_amd.engine import PlanGenError
    from project.plangen.plangen_base import System
    a = _cli()
    a.test_data = dict(a.test_data, task_type="segmentation")
    with pytest.raises(PlanGenError, match="task_type"):
        System(a, None)


def test_cli_resume_errors_are_loud(tmp_path):
    """a missing explicit resume path / a missing janus_path must not fall back to random weights."""
    from project.plangen.plangen_base import System
    s = object.__new__(System)
    s.engine = None
    s.synthetic = False
    s.cli = SimpleNamespace(out_path=str(tmp_path), resume=str(tmp_path / "checkpoint-9"), janus_path=str(tmp_path / "nope"))
    with pytest.raises(FileNotFoundError, match="resume"):
        s.resume()
    s.cli.resume = None
    with pytest.raises(FileNotFoundError, match="janus_path"):
        s.resume()


def test_mmu_chat_template_matches_reference_conversation_py():
    """wrap_mmu_prompt's text (plangen_base.py:263-279) as VLChatProcessor renders it -- the reference's conversation.py under the
    processor's own system prompt (fixture: oracle/make_golden.py::golden_text); ADVICE r2: mmu prompts were bare caption tokens."""
    cases = [c for c in json.load(open(os.path.join(GOLDEN, "text_golden.json"))) if "mmu_question" in c]
    assert len(cases) >= 3
    for c in cases:
        assert T.wrap_mmu_prompt_text(c["mmu_question"], c["mmu_answer"]) == c["prompt"], c
    assert cases[0]["prompt"].startswith("You are a helpful language and vision assistant.") and cases[0]["prompt"].endswith("<|Assistant|>:")
    assert "<|User|>: <image_placeholder>\nDescribe the layout of the image.\n\n<|Assistant|>:" in cases[0]["prompt"]


def test_image_placeholder_expansion_like_add_image_token():
    """processing_vlm.py:243-248 (add_special_token=False): the placeholder id is replaced by boi + P slots + eoi."""
    assert T.expand_image_tokens([1, 5, 9, 6], image_id=9, image_start_id=20, image_end_id=21, num_image_tokens=3) == [1, 5, 20, 9, 9, 9, 21, 6]
    assert T.expand_image_tokens([9, 9], 9, 20, 21, 1) == [20, 9, 21, 20, 9, 21]
    c = T.TagWordCodec(512)
    prompt, ids, slots = T.wrap_mmu_prompt_ids(c, "what is there?", 4)
    img, boi, eoi = c.token_id(T.IMAGE_TAG), c.token_id(T.IMAGE_START_TAG), c.token_id(T.IMAGE_END_TAG)
    k = ids.index(boi)
    assert ids[k:k + 6] == [boi, img, img, img, img, eoi] and sum(slots) == 4 and slots[k + 1:k + 5] == [True] * 4
    assert c.decode(ids[:k]).endswith("<|User|>: ") and c.decode(ids[k + 6:]).startswith("\nwhat is there?")


def test_cli_shards_whole_batches_so_file_names_cannot_collide(monkeypatch):
    """64 rows over 3 ranks with test_batch_size 8 -- every rank starts on a batch boundary, the global batch indices
    partition 0..7, and the names {gidx*bs+i} of all ranks are exactly 0..63."""
    from project.plangen import plangen_base as P
    names = []
    for rank in range(3):
        s = object.__new__(P.System)
        s.cli = SimpleNamespace(test_batch_size=8)
        s._rows = lambda: [{"k": i} for i in range(64)]
        s.collate = lambda rows: rows
        monkeypatch.setattr(P, "world", lambda r=rank: (r, 3))
        dl = s.setup_data()
        assert s.row_offset == s.batch_offset * 8
        for idx, b in enumerate(dl):
            names += [(s.batch_offset + idx) * 8 + i for i in range(len(b))]
            assert [r["k"] for r in b] == list(range((s.batch_offset + idx) * 8, (s.batch_offset + idx) * 8 + len(b)))
    assert sorted(names) == list(range(64)) and len(set(names)) == 64


def test_cli_integer_resume_maps_to_checkpoint_dir(tmp_path):
    """base_system.py:132-134: resume=<int> -> out_path/checkpoint-<int>."""
    from project.plangen.plangen_base import System
    s = object.__new__(System)
    s.engine = None
    s.synthetic = False
    s.cli = SimpleNamespace(out_path=str(tmp_path), resume=7, janus_path=str(tmp_path / "nope"))
    with pytest.raises(FileNotFoundError, match="checkpoint-7"):
        s.resume()
    os.makedirs(tmp_path / "checkpoint-7")
    with pytest.raises(FileNotFoundError, match="janus_path"):          # the overlay resolves; the base weights are what is missing now
        s.resume()


def _tiny_hf_tokenizer_dir(tmp_path):
    """A small byte-level BPE tokenizer saved in the HF layout (tokenizer.json + tokenizer_config.json + special_tokens_map.json) with the
    Janus special tokens -- what VLChatProcessor.from_pretrained(janus_path).tokenizer loads, minus the real vocabulary (no network here)."""
    from tokenizers import AddedToken, Tokenizer, decoders, models, pre_tokenizers, processors, trainers
    from transformers import PreTrainedTokenizerFast
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    specials = ["<｜begin▁of▁sentence｜>", T.SEP2, T.PAD_TAG, T.IMAGE_TAG, T.IMAGE_START_TAG, T.IMAGE_END_TAG, "<|User|>", "<|Assistant|>",
                "<grounding>", "</grounding>", "<ref>", "</ref>", "<box>", "</box>"]
    corpus = ["a red cat on the table", "two dogs playing in a field", "what is on the table?", "0 1 2 3 4 5 6 7 8 9 , [ ] : \n\n",
              T.MMU_SYSTEM_PROMPT, "Describe the layout of the image."]
    tok.train_from_iterator(corpus * 4, trainers.BpeTrainer(vocab_size=400, special_tokens=specials, initial_alphabet=pre_tokenizers.ByteLevel.alphabet()))
    bos = tok.token_to_id(specials[0])
    tok.post_processor = processors.TemplateProcessing(single=f"{specials[0]} $A", special_tokens=[(specials[0], bos)])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token=specials[0], eos_token=T.SEP2, pad_token=T.PAD_TAG,
                                   additional_special_tokens=[AddedToken(s, special=True) for s in specials[3:]])
    d = tmp_path / "janus_tok"
    fast.save_pretrained(str(d))
    return str(d)


def test_hf_codec_on_tokenizer_files(tmp_path):
    """HFCodec (the path real Janus-Pro tokenizer files take): BOS prepended like tokenizer.encode, special tags are single ids,
    pad / eos ids come from the files, decode keeps specials (plangen_base.py:294), and the mmu prompt expands its placeholder."""
    d = _tiny_hf_tokenizer_dir(tmp_path)
    assert os.path.exists(os.path.join(d, "tokenizer.json"))
    c = T.HFCodec(d)
    assert c.eos_token_id == c.token_id(T.SEP2) and c.pad_id == c.token_id(T.PAD_TAG)
    prompt, ids = T.wrap_uni_prompt_ids(c, "a red cat on the table", "<grounding><ref>a cat</ref><box>[1,2,300,400]</box></grounding>")
    assert ids[0] == c.bos_token_id and ids[-1] == c.token_id(T.IMAGE_START_TAG)
    assert ids.count(c.token_id("<ref>")) == 1 and ids.count(c.token_id("<|User|>")) == 1 and ids.count(c.eos_token_id) == 1
    assert c.decode(ids[1:]) == prompt
    p1, ids1 = T.wrap_uni_prompt_ids(c, "two dogs", "<grounding>", in_stage1=True)
    assert p1.endswith("<grounding>" + T.SEP2) and ids1[-1] == c.token_id("<grounding>")            # trailing EOS-tag token dropped (:259-260)
    _, mids, slots = T.wrap_mmu_prompt_ids(c, "what is on the table?", 5)
    k = mids.index(c.token_id(T.IMAGE_START_TAG))
    assert mids[k + 1:k + 6] == [c.token_id(T.IMAGE_TAG)] * 5 and mids[k + 6] == c.token_id(T.IMAGE_END_TAG) and sum(slots) == 5
    assert c.decode(mids[1:k]).startswith(T.MMU_SYSTEM_PROMPT) and c.decode(mids[k + 7:]).endswith("<|Assistant|>:")
    with pytest.raises(KeyError):
        c.token_id("<no_such_tag>")
    # answers cut at the first EOS and parsed (decode_mmu_text_batch / trans_gr_to_creati on real tokenizer output)
    ans = c.encode("<ref>a cat</ref><box>[12,40,500,620]</box>")[1:] + [c.eos_token_id] + c.encode("junk")[1:]
    text = c.decode(T.cut_at_eos(ans, c.eos_token_id))
    assert T.trans_gr_to_creati(text) == ([[0.012, 0.04, 0.5, 0.62]], ["a cat"])
