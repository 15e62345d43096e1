"""plangen_amd -- MI355X-native layout->image generation path for PlanGen (Janus-Pro-1B).

Only what the hot path needs: ``csrc/`` (hand-written HIP kernels + the C ABI of
``include/plangen_hip.h``), a ctypes binding, and the host-side mirror of the reference's
call surface (``janus.MultiModalityCausalLM`` facade, ``system.System``).  PyTorch is used
for device memory, streams and torch.distributed only.  There is no CPU fallback: importing
the binding without the built library raises.
"""
import os as _os

# The decode loop is ~100 k plain stream launches per call: kernel arguments must go straight to device memory.  This is
# ROCm 7.2's default on gfx950; with HIP_FORCE_DEV_KERNARG=0 the loop measures 6 % (bs=64) to 17 % (bs=8) slower
# (DESIGN 4.1).  Only effective when set before the HIP runtime loads, hence here and at the top of bench.py.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from .config import PlanGenConfig  # noqa: E402,F401

__all__ = ["PlanGenConfig"]
