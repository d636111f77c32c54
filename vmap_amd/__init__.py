"""vmap_amd - MI355X-native vectorised per-object training step (drop-in for kxhit/vMAP's hot path).

Host code is Python on PyTorch-ROCm; the arithmetic lives in ``libvmapstep.so`` (hand-written HIP for gfx950,
C ABI in ``include/vmapstep.h``).  Importing the package does not load the library; ``vmap_amd.step.VmapStep``
does, and raises if it is missing - there is no CPU or eager-PyTorch fallback for the training step.
"""
from . import layout  # noqa: F401

__all__ = ["layout", "synth", "fields", "trainer", "ensemble", "step"]
