import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=True)


@pytest.fixture(scope="session")
def tiny_cfg():
    from plangen_amd.config import PlanGenConfig
    return PlanGenConfig.tiny()


@pytest.fixture(scope="session")
def ocfg(tiny_cfg):
    from oracle.ref_cpu import OracleCfg
    return OracleCfg(**tiny_cfg.model_dict())


@pytest.fixture(scope="session")
def tiny_weights(ocfg):
    """Seeded weights, identical to the ones the golden fixtures were generated with."""
    from oracle.ref_cpu import make_weights
    return make_weights(ocfg, seed=1, with_encoder=True, with_vision=True)


def wsum(W):
    return float(sum(v.double().abs().sum() for v in W.values()))


_ENGINES = {}


def get_engine(tiny_cfg, tiny_weights, dtype, **kw):
    """One engine per (dtype, options) for the whole session (creation uploads weights)."""
    from plangen_amd.engine import Engine
    key = (dtype, tuple(sorted(kw.items())))
    if key not in _ENGINES:
        args = dict(max_rows=8, max_prompt=96, max_new=tiny_cfg.img_tokens, max_images=4, with_lm_head=True,
                    with_vq_encoder=True, with_vision=True)
        args.update(kw)
        e = Engine(tiny_cfg, dtype=dtype, **args)
        e.load_state_dict(tiny_weights)
        _ENGINES[key] = e
    return _ENGINES[key]
