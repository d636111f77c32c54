import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver's GPU tier)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than a few seconds")


def relerr(a, b):
    """max-norm relative error: max|a-b| / max|b| (b = reference)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def tensor_err_q(a, b, q=0.999, floor=1e-3):
    """Per-tensor check that a systematically wrong LOW-magnitude block cannot hide under one large element: the
    q-quantile over the tensor's elements of |a - b| / (|b| + floor * max|b|)."""
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.quantile(np.abs(a - b) / (np.abs(b) + floor * (np.abs(b).max() + 1e-30)), q))


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))


GRAD_KEYS = [f"g_fc{t}" for t in range(14)] + ["g_B"]
RENDER_KEYS = ["render_depth", "render_color", "opacity"]


def round_bf16(a):
    """float32 array -> nearest bfloat16 (ties to even), returned as float32 (the rounding vmapstep applies to the
    parameter image when weight_dtype = bf16)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(a))
