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


@pytest.fixture(scope="session", autouse=True)
def _background_memory_load():
    """PG_BG_LOAD=<mode> (0 LDS-DMA, 1 LDS-DMA nt, 2 register loads): run the whole GPU suite while a background kernel on another
    stream streams a private 768 MB buffer (round 4: the decode GEMM's LDS-DMA staging lost pieces under concurrent memory load;
    every LDS-DMA kernel is screened this way, tools/gpu_under_load.sh).  Off by default."""
    mode = os.environ.get("PG_BG_LOAD")
    if mode is None or not torch.cuda.is_available():
        yield
        return
    import ctypes
    import threading
    import time
    from plangen_amd import _lib
    lib = _lib.load_diag()          # the stressor kernel lives in the diagnostics library; the engines under test stay in libplangen_hip.so
    stop = threading.Event()
    blocks, depth = int(os.environ.get("PG_BG_BLOCKS", "256")), int(os.environ.get("PG_BG_DEPTH", "16"))

    def pump():
        while not stop.is_set():
            if lib.pg_bench_background_done():
                lib.pg_bench_background(768, 400, blocks, depth, int(mode))
            time.sleep(0.01)
    t = threading.Thread(target=pump, daemon=True)
    t.start()
    yield
    stop.set()
    t.join(timeout=10)
    lib.pg_bench_background_join()
