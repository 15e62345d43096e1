"""plangen_amd -- MI355X-native layout->image generation path for PlanGen (Janus-Pro-1B).

Only what the hot path needs: ``csrc/`` (hand-written HIP kernels + the C ABI of
``include/plangen_hip.h``), a ctypes binding, and the host-side mirror of the reference's
call surface (``janus.MultiModalityCausalLM`` facade, ``system.System``).  PyTorch is used
for device memory, streams and torch.distributed only.  There is no CPU fallback: importing
the binding without the built library raises.
"""
from .config import PlanGenConfig  # noqa: F401

__all__ = ["PlanGenConfig"]
