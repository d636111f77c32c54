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


def kink_aware(got, ref, n_obj, signed=False):
    """ReLU-kink accounting (oracle.vmap_oracle.kink_deltas): ``ref`` = an oracle result computed with ``kinks=True``, ``got`` = another
    float32 implementation's result.  Per object, the derivative bit of every kink-adjacent hidden unit is solved for by least
    squares (one unknown per ambiguous entry against the object's ~10^4..10^5 gradient elements), REQUIRED to round to 0 or 1,
    and the rounded combination is added to the oracle's gradients.  Returns ({key: corrected oracle gradient}, flipped bits,
    ambiguous entries, worst effect of rounding a beta, in units of the affected tensor's max).

    ``signed``: ``ref``'s gradients come from a THIRD implementation (a reference fixture) while its ``kink_deltas`` are the
    oracle's: a bit may then differ from the oracle's state in ``ref``, in ``got`` or in both, so each coefficient is the
    difference of two bits, in {-1, 0, +1}."""
    shapes = [np.shape(ref[k]) for k in GRAD_KEYS]
    flat = lambda d, k: np.concatenate([np.asarray(d[key], np.float64)[k].ravel() for key in GRAD_KEYS])
    # per-element scale of an object's flat gradient vector: the max of the tensor the element belongs to (the tolerance is per tensor)
    corr = {key: np.array(ref[key], dtype=np.float64) for key in GRAD_KEYS}
    flipped, worst = 0, 0.0
    for k in range(n_obj):
        scale = np.concatenate([np.full(int(np.prod(shp[1:])), np.abs(np.asarray(ref[key])[k]).max() + 1e-30)
                                for key, shp in zip(GRAD_KEYS, shapes)])
        # an entry whose flip moves no tensor by 1e-7 of its max (dead downstream path, masked ray) is not ambiguous in effect
        D = [d for (ko, d) in ref["kink_deltas"] if ko == k and np.abs(d / scale).max() > 1e-7]
        if not D:
            continue
        A = np.stack(D, axis=1)
        beta, *_ = np.linalg.lstsq(A / scale[:, None], (flat(got, k) - flat(ref, k)) / scale, rcond=None)
        rb = np.clip(np.round(beta), -1 if signed else 0, 1)
        worst = max(worst, float((np.abs(beta - rb) * np.abs(A / scale[:, None]).max(axis=0)).max()))
        flipped += int(np.abs(rb).sum())
        add = A @ rb
        o = 0
        for key, shp in zip(GRAD_KEYS, shapes):
            sz = int(np.prod(shp[1:]))
            corr[key][k] += add[o:o + sz].reshape(shp[1:])
            o += sz
    return corr, flipped, len(ref["kink_deltas"]), worst
