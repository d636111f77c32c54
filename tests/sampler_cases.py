"""Seeded synthetic keyframe buffers + per-ray random numbers for the sampler parity tests (shared by the golden
generator, the CPU tests and the GPU tests)."""
from __future__ import annotations

import hashlib

import numpy as np

CASES = {
    # name: (K keyframes, W, H, F frames, P samples/frame, n1, n2, seed)
    "obj":   (4, 64, 48, 10, 6, 1, 9, 1),      # object shapes: 1 camera-to-surface bin + 9 surface bins
    "bg":    (3, 48, 40, 8, 5, 5, 9, 2),       # background shapes: 5 + 9
    "twokf": (2, 40, 32, 6, 4, 1, 9, 3),       # n_keyframes <= 2: no forced 'latest two' (vmap.py:322-341)
}


def _rot(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def build_scene(name):
    K, W, H, F, P, n1, n2, seed = CASES[name]
    rng = np.random.default_rng(seed)
    rgbs = rng.integers(0, 256, (K, W, H, 4)).astype(np.uint8)
    state = rng.choice(np.array([0, 1, 2], dtype=np.uint8), size=(K, W, H), p=[0.35, 0.55, 0.10])
    rgbs[..., 3] = state
    depth = rng.uniform(0.4, 4.0, (K, W, H)).astype(np.float32)
    depth[rng.uniform(0, 1, (K, W, H)) < 0.08] = 0.0                        # invalid depth
    t_wc = np.zeros((K, 4, 4), dtype=np.float32)
    bbox = np.zeros((K, 4), dtype=np.float32)
    for k in range(K):
        t_wc[k, :3, :3] = _rot(rng)
        t_wc[k, :3, 3] = rng.uniform(-1, 1, 3)
        t_wc[k, 3, 3] = 1.0
        u0, v0 = rng.integers(0, W // 3), rng.integers(0, H // 3)
        bbox[k] = [u0, rng.integers(W // 2, W - 1), v0, rng.integers(H // 2, H - 1)]
    return dict(name=name, K=K, W=W, H=H, F=F, P=P, n1=n1, n2=n2, rgbs=rgbs, depth=depth, t_wc=t_wc, bbox=bbox,
                intr=(60.0, 55.0, (W - 1) / 2.0, (H - 1) / 2.0), center=rng.uniform(-0.5, 0.5, 3).astype(np.float32),
                last2=(K - 2, K - 1), min_bound=0.0, seed=seed)


def draw_randoms(sc):
    rng = np.random.default_rng(1000 + sc["seed"])
    F, P, S, n2, K = sc["F"], sc["P"], sc["n1"] + sc["n2"], sc["n2"], sc["K"]
    kf = rng.integers(0, K, F).astype(np.int64)
    if K > 2:
        kf[F - 2:] = sc["last2"]                                            # vmap.py:329-331
    return dict(kf_ids=kf, u_w=rng.uniform(0, 1, (F, P)).astype(np.float32), u_h=rng.uniform(0, 1, (F, P)).astype(np.float32),
                u_z=rng.uniform(0, 1, (F * P, S)).astype(np.float32), g_z=rng.standard_normal((F * P, n2)).astype(np.float32))


def digest(sc, rnd):
    h = hashlib.sha256()
    for k in ("rgbs", "depth", "t_wc", "bbox", "center"):
        h.update(np.ascontiguousarray(sc[k]).tobytes())
    for k in ("kf_ids", "u_w", "u_h", "u_z", "g_z"):
        h.update(np.ascontiguousarray(rnd[k]).tobytes())
    return h.hexdigest()
