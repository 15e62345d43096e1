"""Host-side text plumbing either side of the hot path: the chat template
the prompts are wrapped in, the post-processing of greedy layout / answer tokens, and the stage-1 -> stage-2
hand-off of ``task_type='uni_2stage'``.  Pure Python on ids and strings; the tokenizer itself is pluggable:

* :class:`HFCodec` -- the HF ``LlamaTokenizerFast`` files of ``deepseek-ai/Janus-Pro-1B`` when they are on disk
  (what ``VLChatProcessor.from_pretrained(janus_path).tokenizer`` loads, plangen_base.py:97-100);
* :class:`TagWordCodec` -- a small reversible tag/word/number vocabulary for offline runs and tests (no files).

Reference lines (relative to /root/reference):
  apply_sft_template_for_multi_turn_prompts  three_party/Janus/janus/models/processing_vlm.py:137-177
  Conversation.get_prompt, DeepSeek style     three_party/Janus/janus/utils/conversation.py:80-91, :293-308
  wrap_t2i_prompt / wrap_uni_prompt           project/plangen/plangen_base.py:210-261
  decode_plan_text_batch                      project/plangen/plangen_base.py:296-306
  get_pr_grounding_part                       project/plangen/plangen_base.py:308-312
  decode_mmu_text_batch                       project/plangen/plangen_base.py:314-325
  trans_gr_to_creati                          project/plangen/plangen_base.py:460-473
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Sequence, Tuple

ROLES = ("<|User|>", "<|Assistant|>")                 # conversation.py:300
SEP, SEP2 = "\n\n", "<｜end▁of▁sentence｜>"            # conversation.py:304-305
IMAGE_START_TAG = "<begin_of_image>"                  # processing_vlm.py:89
IMAGE_TAG = "<image_placeholder>"                     # processing_vlm.py:88
IMAGE_END_TAG = "<end_of_image>"                      # processing_vlm.py:90
# VLChatProcessor.system_prompt (processing_vlm.py:78-82): what process_one / __call__ pass to the sft template (:292-296), i.e.
# what wrap_mmu_prompt's prompts start with (wrap_uni_prompt passes system_prompt="" instead, plangen_base.py:245-249)
MMU_SYSTEM_PROMPT = ("You are a helpful language and vision assistant. "
                     "You are able to understand the visual content that the user provides, "
                     "and assist the user with a variety of tasks using natural language.")
PAD_TAG = "<｜▁pad▁｜>"                                # processing_vlm.py:91
GROUNDING_OPEN, GROUNDING_CLOSE = "<grounding>", "</grounding>"


# ------------------------------------------------------------------------------------------ chat template
def apply_sft_template(conversations: Sequence[Dict[str, str]], system_prompt: str = "") -> str:
    """processing_vlm.py:171-177 with the 'deepseek' template (conversation.py:80-91): every message's content is
    stripped; a non-empty message renders as ``role: message sep`` (sep alternates ``\\n\\n`` / EOS tag), an empty one
    as ``role:``; the whole prompt is stripped."""
    seps = (SEP, SEP2)
    ret = "" if not system_prompt else system_prompt + seps[0]
    for i, m in enumerate(conversations):
        msg = m["content"].strip()
        ret += (m["role"] + ": " + msg + seps[i % 2]) if msg else (m["role"] + ":")
    return ret.strip()


def wrap_t2i_prompt_text(caption: str) -> str:
    """System.wrap_t2i_prompt (plangen_base.py:210-230), text part."""
    conv = [{"role": ROLES[0], "content": caption}, {"role": ROLES[1], "content": ""}]
    return apply_sft_template(conv) + IMAGE_START_TAG


def wrap_uni_prompt_text(caption: str, grounding: Optional[str], in_stage1: bool = False) -> str:
    """System.wrap_uni_prompt (plangen_base.py:232-261), text part: the layout string is the assistant turn; stage 2
    appends ``<begin_of_image>``, stage 1 does not (and drops the last TOKEN after encoding, see
    :func:`wrap_uni_prompt_ids`).  ``grounding`` goes through an f-string in the reference, so None renders 'None'."""
    conv = [{"role": ROLES[0], "content": caption}, {"role": ROLES[1], "content": f"{grounding}"}]
    sft = apply_sft_template(conv)
    return sft if in_stage1 else sft + IMAGE_START_TAG


def wrap_mmu_prompt_text(question: str, answer: str = "") -> str:
    """System.wrap_mmu_prompt (plangen_base.py:263-290), text part: the user turn is ``<image_placeholder>\\n{question}``, the
    assistant turn the (usually empty) answer, rendered by VLChatProcessor.process_one with ITS system prompt
    (processing_vlm.py:290-296)."""
    conv = [{"role": ROLES[0], "content": f"{IMAGE_TAG}\n{question}"}, {"role": ROLES[1], "content": f"{answer}"}]
    return apply_sft_template(conv, MMU_SYSTEM_PROMPT)


def expand_image_tokens(ids: Sequence[int], image_id: int, image_start_id: int, image_end_id: int, num_image_tokens: int) -> List[int]:
    """VLChatProcessor.add_image_token (processing_vlm.py:210-262, add_special_token=False): every ``<image_placeholder>`` id is
    REPLACED by ``<begin_of_image>`` + num_image_tokens x ``<image_placeholder>`` + ``<end_of_image>``; the placeholder slots are
    what images_seq_mask marks (:379-380) and prepare_inputs_embeds overwrites with the aligner's output."""
    out: List[int] = []
    for t in ids:
        if int(t) == image_id:
            out += [image_start_id] + [image_id] * num_image_tokens + [image_end_id]
        else:
            out.append(int(t))
    return out


def wrap_mmu_prompt_ids(codec, question: str, num_image_tokens: int, answer: str = "") -> Tuple[str, List[int], List[bool]]:
    """-> (prompt, input_ids with the image slots expanded, images_seq_mask)."""
    prompt = wrap_mmu_prompt_text(question, answer)
    img = codec.token_id(IMAGE_TAG)
    ids = expand_image_tokens(codec.encode(prompt), img, codec.token_id(IMAGE_START_TAG), codec.token_id(IMAGE_END_TAG), num_image_tokens)
    return prompt, ids, [t == img for t in ids]


def wrap_uni_prompt_ids(codec, caption: str, grounding: Optional[str], in_stage1: bool = False) -> Tuple[str, List[int]]:
    prompt = wrap_uni_prompt_text(caption, grounding, in_stage1)
    ids = list(codec.encode(prompt))
    if in_stage1:
        ids = ids[:-1]                                   # plangen_base.py:259-260: drop the trailing EOS-tag token
    return prompt, ids


# ------------------------------------------------------------------------------------------ a12
def cut_plan_text(text: str) -> str:
    """One row of decode_plan_text_batch (plangen_base.py:296-306): the decoded new tokens get ``<grounding>``
    prepended; everything after the first ``</grounding>`` is dropped; no closing tag -> the empty layout."""
    text = GROUNDING_OPEN + text
    end = text.find(GROUNDING_CLOSE)
    return text[:end + len(GROUNDING_CLOSE)] if end != -1 else GROUNDING_OPEN + GROUNDING_CLOSE


def get_pr_grounding_part(text: str) -> str:
    """plangen_base.py:308-312."""
    pos = text.find(GROUNDING_OPEN)
    return text[pos:] if pos != -1 else text


def cut_at_eos(ids: Sequence[int], eos_id: int) -> List[int]:
    """One row of decode_mmu_text_batch (plangen_base.py:316-322): ids before the first EOS (all of them if none)."""
    ids = list(ids)
    return ids[:ids.index(eos_id)] if eos_id in ids else ids


_BOX_RE = re.compile(r"<ref>(.*?)</ref><box>\[(.*?)\]</box>")


def trans_gr_to_creati(prompt: str) -> Tuple[List[List[float]], List[str]]:
    """plangen_base.py:460-473: ``<ref>desc</ref><box>[x1,y1,x2,y2]</box>`` items -> (boxes / 1000 as [x1,y1,x2,y2], descriptions)."""
    boxes, prompts = [], []
    for desc, box in _BOX_RE.findall(prompt):
        x1, y1, x2, y2 = map(int, box.split(","))
        prompts.append(desc)
        boxes.append([x1 / 1000, y1 / 1000, x2 / 1000, y2 / 1000])
    return boxes, prompts


# ------------------------------------------------------------------------------------------ tokenizers
class HFCodec:
    """The Janus-Pro tokenizer files (tokenizer.json / tokenizer_config.json / special_tokens_map.json) through
    ``transformers.AutoTokenizer`` -- the object ``VLChatProcessor.tokenizer`` is (processing_vlm.py:97-125)."""

    def __init__(self, path: str):
        from transformers import AutoTokenizer
        self.tok = AutoTokenizer.from_pretrained(path)
        self.eos_token_id = self.tok.eos_token_id
        self.bos_token_id = self.tok.bos_token_id
        pad = self.tok.convert_tokens_to_ids(PAD_TAG)
        self.pad_id = pad if pad is not None and pad >= 0 else self.tok.pad_token_id

    def encode(self, text: str) -> List[int]:
        return self.tok.encode(text)                      # adds BOS like the reference's tokenizer.encode(prompt)

    def token_id(self, tag: str) -> int:
        """tokenizer.vocab.get(tag) (processing_vlm.py:102, :181-207 image_id / image_start_id / image_end_id)."""
        i = self.tok.convert_tokens_to_ids(tag)
        if i is None or i < 0 or i == self.tok.unk_token_id:
            raise KeyError(f"tokenizer has no token {tag!r}")
        return int(i)

    def decode(self, ids: Sequence[int]) -> str:
        return self.tok.decode(list(ids), skip_special_tokens=False)     # plangen_base.py:294


class TagWordCodec:
    """Reversible offline vocabulary: special tags, ``<...>`` tags, digits, punctuation, newline runs and words
    (words are added on first sight).  ``decode(encode(s)) == s`` for text built from those pieces
    with single spaces -- enough to carry layouts (``<ref>a cat</ref><box>[12,40,500,620]</box>``) through the
    stage-1 -> stage-2 hand-off without tokenizer files.  Not the model's tokenizer: with real weights use HFCodec."""

    _PIECE = re.compile(r"<｜[^｜]*｜>|<\|[^|]*\|>|</?[a-z_]+>|\d|\n+| |[A-Za-z']+|[^\sA-Za-z\d]")

    def __init__(self, vocab_size: int, eos_id: int = 7, pad_id: int = 3, bos_id: int = 1, first_id: int = 8):
        self.vocab_size, self.eos_token_id, self.pad_id, self.bos_token_id = vocab_size, eos_id, pad_id, bos_id
        self._t2i: Dict[str, int] = {SEP2: eos_id, PAD_TAG: pad_id}
        self._i2t: Dict[int, str] = {eos_id: SEP2, pad_id: PAD_TAG, bos_id: ""}
        self._next = first_id
        for t in (ROLES[0], ROLES[1], IMAGE_START_TAG, IMAGE_TAG, IMAGE_END_TAG, GROUNDING_OPEN, GROUNDING_CLOSE, "<ref>", "</ref>", "<box>", "</box>",
                  " ", "\n\n", "\n", ":", ",", "[", "]", "."):
            self._add(t)
        for n in range(10):
            self._add(str(n))

    def _add(self, t: str) -> int:
        if t not in self._t2i:
            while self._next in self._i2t:
                self._next += 1
            if self._next >= self.vocab_size:
                raise ValueError(f"TagWordCodec: vocabulary of {self.vocab_size} ids exhausted at {t!r}")
            self._t2i[t] = self._next
            self._i2t[self._next] = t
        return self._t2i[t]

    def encode(self, text: str) -> List[int]:
        return [self.bos_token_id] + [self._add(p) for p in self._PIECE.findall(text)]

    def token_id(self, tag: str) -> int:
        return self._add(tag)

    def decode(self, ids: Sequence[int]) -> str:
        return "".join(self._i2t.get(int(i), "") for i in ids)
