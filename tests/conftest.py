import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The measurement build of the library (tests/tools/libvmapstep_ab.so = the same sources with -DVMAPSTEP_AB, built by build()): phase
# stamps and the A/B kernel forms no automatic plan launches.  TEST INFRASTRUCTURE: the product package knows nothing about it.
AB_LIBRARY = os.path.join(ROOT, "tests", "tools", "libvmapstep_ab.so")

# plan overrides merged UNDER the ``tuning`` of every operator a test builds through make_op (the "f32" leg of tests/test_gpu_parity.py)
TEST_TUNING = {"default": None}


def make_op(*args, tuning=None, **kw):
    """A ``vmap_amd.step.VmapStep`` for a test: on the PRODUCT library whenever the product carries the requested kernel form
    (every automatic plan, the exact-fp32 kernels step_main_h32 / step_main_gen, workgroups_per_object, generic_finalize, ws_flags
    1 / 4), on the measurement build only when the product's own plan refuses the form ("measurement build only")."""
    from vmap_amd import _lib, step
    tuning = {**(TEST_TUNING["default"] or {}), **(tuning or {})} or None
    try:
        return step.VmapStep(*args, tuning=tuning, **kw)
    except _lib.VmapStepError as e:
        if "measurement build only" not in str(e) or not os.path.exists(AB_LIBRARY):
            raise
    return step.VmapStep(*args, tuning=tuning, library=AB_LIBRARY, **kw)


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


def kink_aware(got, ref, n_obj, signed=False, tol=1e-4):
    """ReLU-kink accounting (oracle.vmap_oracle.kink_deltas): ``ref`` = an oracle result computed with ``kinks=True``, ``got`` = another
    float32 implementation's result.  Per object, while the difference exceeds ``tol`` / 4 of a tensor's max: the ONE candidate bit
    whose flip explains most of the remaining difference is taken, provided its coefficient (a one-unknown least squares against
    the object's ~10^4..10^5 gradient elements) is within 0.25 of 1 (``signed``: of +-1) - matching pursuit, because the flips are
    few and the deltas of different sample points nearly orthogonal, while a joint least squares over hundreds of candidates fits
    rounding noise with near-collinear columns.  Returns ({key: corrected oracle gradient}, flipped bits, ambiguous entries, worst
    distance of an applied coefficient from +-1 weighted by its effect, in units of the affected tensor's max).

    ``signed``: ``ref``'s gradients come from a THIRD implementation (a reference fixture) while its ``kink_deltas`` are the
    oracle's: a bit may then differ from the oracle's state in ``ref``, in ``got`` or in both, so each coefficient is the
    difference of two bits, in {-1, 0, +1}."""
    shapes = [np.shape(ref[k]) for k in GRAD_KEYS]
    flat = lambda d, k: np.concatenate([np.asarray(d[key], np.float64)[k].ravel() for key in GRAD_KEYS])
    corr = {key: np.array(ref[key], dtype=np.float64) for key in GRAD_KEYS}
    flipped, worst = 0, 0.0
    for k in range(n_obj):
        # per-element scale of an object's flat gradient vector: the max of the tensor the element belongs to (the tolerance is per tensor)
        scale = np.concatenate([np.full(int(np.prod(shp[1:])), np.abs(np.asarray(ref[key])[k]).max() + 1e-30)
                                for key, shp in zip(GRAD_KEYS, shapes)])
        r = (flat(got, k) - flat(ref, k)) / scale
        # an entry whose flip moves no tensor by tol / 4 of its max cannot be what separates the two results
        D = [d / scale for (ko, d) in ref["kink_deltas"] if ko == k and np.abs(d / scale).max() > tol / 4]
        if not D or np.abs(r).max() < tol / 4:
            continue
        A = np.stack(D, axis=0)                                     # [candidates, elements]: one BLAS product per pursuit step
        nrm = np.einsum("ce,ce->c", A, A)
        used = np.zeros(A.shape[0], dtype=bool)
        add = np.zeros_like(r)
        for _ in range(A.shape[0]):
            if np.abs(r).max() < tol / 4:
                break
            beta = (A @ r) / nrm
            rb = np.clip(np.round(beta), -1 if signed else 0, 1)
            gain = np.where((rb != 0) & ~used & (np.abs(beta - rb) < 0.25), (2 * beta * rb - rb * rb) * nrm, -np.inf)
            j = int(np.argmax(gain))
            if not np.isfinite(gain[j]) or gain[j] <= 0:
                break
            used[j] = True
            r = r - rb[j] * A[j]
            add += rb[j] * A[j]
            flipped += 1
            worst = max(worst, float(abs(beta[j] - rb[j]) * np.abs(A[j]).max()))
        add = add * scale
        o = 0
        for key, shp in zip(GRAD_KEYS, shapes):
            sz = int(np.prod(shp[1:]))
            corr[key][k] += add[o:o + sz].reshape(shp[1:])
            o += sz
    return corr, flipped, len(ref["kink_deltas"]), worst
