"""CPU: host-side logic of the product package (no GPU compute): collate semantics vs the
oracle's restatement, config, mask handling, and that the C-ABI library loads and exports
every symbol include/plangen_hip.h declares."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from oracle import ref_cpu as R
from plangen_amd import _lib
from plangen_amd.config import PlanGenConfig
from plangen_amd.system import pad_input_ids, t2i_infer_collate_batch


def test_config_shapes():
    c = PlanGenConfig.janus_pro_1b()
    assert (c.img_tokens, c.img_size, c.hidden, c.inter, c.n_layers) == (576, 384, 2048, 5632, 24)
    t = PlanGenConfig.tiny()
    assert t.img_tokens == 64 and t.img_size == 32
    assert set(t.model_dict()) == set(R.OracleCfg().__dataclass_fields__)


def test_pad_input_ids_left_pads_like_reference():
    prompts = [[5, 6, 7], [9], [1, 2, 3, 4, 5]]
    ids, mask = pad_input_ids(prompts, pad_id=3)
    rid, rmask = R.pad_input_ids(prompts, 3)
    assert torch.equal(ids, rid) and torch.equal(mask, rmask)
    ids2, _ = pad_input_ids(prompts, pad_id=3, debug_max_seq_len=8)      # cfg debug_max_seq_len
    assert ids2.shape == (3, 8) and (ids2[:, :3] == 3).all()
    with pytest.raises(Exception):
        pad_input_ids(prompts, pad_id=3, max_length=2)


@pytest.mark.parametrize("neg_len", [2, 6, 9])
def test_cfg_collate_matches_oracle(neg_len):
    g = torch.Generator().manual_seed(neg_len)
    cond = [torch.randint(8, 500, (n,), generator=g).tolist() for n in (7, 3, 5)]
    neg = torch.randint(8, 500, (neg_len,), generator=g).tolist()
    ids, mask = t2i_infer_collate_batch(cond, neg, 3, 64)
    rid, rmask = R.t2i_infer_collate_batch(cond, neg, 3, 64)
    assert torch.equal(ids, rid) and torch.equal(mask, rmask)
    assert ids.shape[0] == 6 and mask.shape[1] == ids.shape[1] + 64


def test_collate_reproduces_golden_fixture():
    g = load_golden("sample_image_tiny.npz")
    ids, mask = t2i_infer_collate_batch([list(c) for c in g["cond"]], g["neg"].tolist(), 3, 64)
    assert np.array_equal(ids.numpy(), g["ids"]) and np.array_equal(mask.numpy(), g["mask"])


def test_per_sample_negative_prompts():
    cond = [[5, 6, 7, 8], [9, 10]]
    negs = [[1, 2], [3, 4, 5, 6, 7]]
    ids, mask = t2i_infer_collate_batch(cond, negs, 0, 4)
    assert ids.shape == (4, 5)
    assert ids[1].tolist() == [0, 0, 0, 1, 2] and ids[3].tolist() == [3, 4, 5, 6, 7]
    assert mask[0].tolist() == [0, 1, 1, 1, 1, 1, 1, 1, 1]
    rid, rmask = R.t2i_infer_collate_batch(cond, negs, 0, 4)               # the oracle's use_neg_box branch (plangen_base.py:652-670)
    assert torch.equal(ids, rid) and torch.equal(mask, rmask)
    cond2 = [[5, 6, 7, 8, 9, 10, 11], [9, 10]]                             # cond longer than every negative prompt
    a, b = t2i_infer_collate_batch(cond2, negs, 0, 4), R.t2i_infer_collate_batch(cond2, negs, 0, 4)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[0].shape == (4, 7)


def test_pad_len_from_mask():
    from plangen_amd.engine import Engine, PlanGenError
    m = torch.tensor([[0, 0, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1]])
    assert Engine.pad_len_from_mask(m, 4) == [2, 0]
    with pytest.raises(PlanGenError):
        Engine.pad_len_from_mask(torch.tensor([[1, 0, 1, 1]]), 4)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "plangen_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pg_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    lib = _lib.load()                       # raises if the .so is missing or a symbol is absent
    for name in declared:
        assert hasattr(lib, name)


def _dynamic_exports(path):
    import subprocess
    nm = "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm if os.path.exists(nm) else "nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {l.split()[-1] for l in out.splitlines() if l.strip()}


def test_product_library_exports_nothing_but_the_header():
    """libplangen_hip.so exports EXACTLY what include/plangen_hip.h declares (version script plangen_hip.map) -- no
    microbenchmark / forensics entry points, no C++ launchers; those live in libplangen_diag.so."""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "plangen_hip.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(pg_[a-z0-9_]+)\s*\(", hdr))
    exported = _dynamic_exports(_lib.LIB_PATH)
    assert exported == declared, exported ^ declared
    diag = _dynamic_exports(_lib.DIAG_LIB_PATH)
    assert declared < diag and "pg_diag_set_option" in diag and any(n.startswith("pg_bench_") for n in diag)
    assert all(n.startswith("pg_") for n in diag), [n for n in diag if not n.startswith("pg_")][:5]


def test_product_option_table_is_small_and_has_no_result_changing_switch():
    """pg_set_option of the product: <= 21 keys (round 6 added vq_tail_fused, a bit-identical A/B fallback), none of the measurement-only or
    rounding-changing switches (skip_attn, attn_variant, defer_norm); those exist only behind pg_diag_set_option."""
    api = open(os.path.join(ROOT, "plangen_amd", "csrc", "engine_api.hip")).read()
    body = api[api.index("int pg_set_option("):api.index("int64_t pg_device_bytes(")]
    keys = re.findall(r'strcmp\(key, "([a-z0-9_]+)"\)', body)
    assert len(keys) == len(set(keys)) and 10 <= len(keys) <= 21, keys
    diag_src = open(os.path.join(ROOT, "plangen_amd", "csrc", "diag_api.hip")).read()
    diag_keys = set(re.findall(r'strcmp\(key, "([a-z0-9_]+)"\)', diag_src))
    for k in ("skip_attn", "attn_variant", "defer_norm"):
        assert k not in keys and k in diag_keys
    for gone in ("cu_split", "mall_prefetch", "pf_blocks", "attn_pair", "lpt_snake", "gn_fuse"):
        assert gone not in keys and gone not in diag_keys
    hdr = open(os.path.join(ROOT, "include", "plangen_hip.h")).read()
    for k in keys:
        assert k in hdr, f"option {k} is not documented in include/plangen_hip.h"


def test_no_cpu_fallback_without_gpu():
    from plangen_amd.engine import Engine, PlanGenError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(PlanGenError):
        Engine(PlanGenConfig.tiny())


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "plangen_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_checkpoint_discovery_and_safetensors_iteration(tmp_path):
    """HF safetensors directory + PlanGen 'latest' checkpoint discovery (no GPU)."""
    from safetensors.torch import save_file
    from plangen_amd.weights import iter_safetensors, latest_checkpoint
    a = {"gen_embed.weight": torch.randn(4, 8), "language_model.model.norm.weight": torch.ones(6)}
    b = {"gen_head.vision_head.bias": torch.zeros(5)}
    save_file(a, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file(b, str(tmp_path / "model-00002-of-00002.safetensors"))
    got = dict(iter_safetensors(str(tmp_path)))
    assert set(got) == set(a) | set(b) and torch.equal(got["gen_embed.weight"], a["gen_embed.weight"])
    for step in (10, 200, 30):
        (tmp_path / f"checkpoint-{step}").mkdir()
    assert latest_checkpoint(str(tmp_path)).endswith("checkpoint-200")
    assert latest_checkpoint(str(tmp_path / "checkpoint-10")) is None


def test_cli_config_loader_matches_reference_shape():
    """train.py --cfg <python cfg with _base_> --opt dotted=overrides (train.py:23-49)."""
    import train
    a = train.parse_args(["--cfg", os.path.join(ROOT, "project/plangen/cfg/uni/h_text_ump+oimsam.py"), "--opt", "test=True",
                          "test_data.task_type='uni_2stage'", "cfg_weight=3.5", "resume=None"])
    assert a.test is True and a.cfg_weight == 3.5 and a.resume is None
    assert a.test_data["task_type"] == "uni_2stage" and a.test_data["data_name"] == "synthetic"
    assert a.out_path.endswith("h_text_ump+oimsam") and a.system_cls_path == "project.plangen.plangen_base"


def test_package_pins_device_kernargs_for_the_stream_launched_loop():
    """The decode loop is ~100 k stream launches per call; with HIP_FORCE_DEV_KERNARG=0 it measures 6-17 % slower.
    The package pins the ROCm default before the HIP runtime can load, without overriding an explicit choice."""
    import subprocess
    import sys
    code = "import os, plangen_amd; print(os.environ.get('HIP_FORCE_DEV_KERNARG'))"
    env = {k: v for k, v in os.environ.items() if k != "HIP_FORCE_DEV_KERNARG"}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.check_output([sys.executable, "-c", code], env=env, cwd=root).decode().strip() == "1"
    env["HIP_FORCE_DEV_KERNARG"] = "0"
    assert subprocess.check_output([sys.executable, "-c", code], env=env, cwd=root).decode().strip() == "0"


def test_uncond_rows_shared_host_check():
    """Engine.uncond_rows_shared: the host-side form of pg_prefill's device probe (every odd row carries row 1's padding and ids)."""
    from plangen_amd.engine import Engine
    cond = [[5, 6, 7, 8], [9, 10], [4, 4, 4]]
    ids, mask = t2i_infer_collate_batch(cond, [1, 2, 3], 0, 4)
    pad = Engine.pad_len_from_mask(mask, ids.shape[1])
    assert Engine.uncond_rows_shared(ids, pad)
    ids2 = ids.clone(); ids2[5, -1] = 99
    assert not Engine.uncond_rows_shared(ids2, pad)
    ids3, mask3 = t2i_infer_collate_batch(cond, [[1, 2], [1, 2, 3], [1, 2]], 0, 4)           # per-sample negatives of different lengths
    assert not Engine.uncond_rows_shared(ids3, Engine.pad_len_from_mask(mask3, ids3.shape[1]))
    assert not Engine.uncond_rows_shared(ids[:2], pad[:2])                                    # a single pair: nothing to share
