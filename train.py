#!/usr/bin/env python3
"""CLI with the reference's shape (train.py:23-49, README.md:58-64):

    python train.py --cfg project/plangen/cfg/uni/h_text_ump+oimsam.py \\
        --opt test=True resume=<ckpt> test_data.data_name='creati' test_data.task_type='uni'

Only the inference branch (``test=True`` -> ``System.validation``, train.py:132-136) exists here:
training is outside the MI355X path.  The config loader understands the mmengine-style python files
the reference uses (flat globals, ``_base_`` inheritance, dotted ``--opt`` overrides)."""
import argparse
import ast
import importlib
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_cfg(path):
    ns = {}
    src = open(path).read()
    exec(compile(src, path, "exec"), ns)
    cfg = {}
    for base in ns.get("_base_", []):
        cfg.update(load_cfg(os.path.normpath(os.path.join(os.path.dirname(path), base))))
    for k, v in ns.items():
        if not k.startswith("_") and not callable(v) and not isinstance(v, type(os)):
            if isinstance(v, dict) and isinstance(cfg.get(k), dict):
                cfg[k] = {**cfg[k], **v}
            else:
                cfg[k] = v
    return cfg


def apply_opts(cfg, opts):
    for kv in opts:
        key, val = kv.split("=", 1)
        try:
            val = ast.literal_eval(val)
        except Exception:
            pass
        d = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = val
    return cfg


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", required=True)
    ap.add_argument("--opt", nargs="*", default=[])
    a = ap.parse_args(argv)
    return SimpleNamespace(**apply_opts(load_cfg(a.cfg), a.opt))


def main(argv=None):
    args = parse_args(argv)
    if not getattr(args, "test", False):
        raise SystemExit("only test=True (System.validation) is implemented: training is out of the MI355X path")
    import torch.distributed as dist
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime
        import torch
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        # same arguments as bench.py: the communicator is bound to this rank's device up front (no lazy init on the first
        # collective) and a dead peer fails the collective after a FINITE time instead of hanging the node
        dist.init_process_group("nccl", device_id=torch.device("cuda", local),
                                timeout=datetime.timedelta(seconds=float(os.environ.get("PG_DIST_TIMEOUT_S", "900"))))
    system_cls = getattr(importlib.import_module(args.system_cls_path), "System")     # train.py:85-86
    model = system_cls(args, None)
    model.setup_data(None)
    model.resume(None)
    print(model.validation(0))


if __name__ == "__main__":
    main()
