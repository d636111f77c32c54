"""Seeded synthetic inputs for the per-object training step (numpy only, no torch).

There is no dataset in the build/bench environment, so parameters and ray batches are drawn
to match the distributions the reference produces:

* parameters: ``xavier_normal_`` weights (reference ``model.py:4-6`` applied by ``trainer.py:32``),
  PyTorch ``nn.Linear`` default biases U(-1/sqrt(fan_in), 1/sqrt(fan_in)), ``B`` = the 21
  icosahedron directions of ``embedding.py:51-73``, ``scale`` = obj_scale (``trainer.py:33``);
* ray batches: the shapes/semantics of ``vmap.py:366-459`` (``sample_3d_points``): one
  camera-to-surface stratified sample followed by sorted clipped-normal (this object) or
  stratified (other object) samples around the measured depth, full-range stratified samples
  for invalid-depth rays; labels 0 = other object, 1 = this object, 2 = unknown (``vmap.py:154-156``).

The same generator feeds the golden-vector script, the parity tests and ``bench.py`` so that all
of them see bit-identical inputs for a given seed.
"""
from __future__ import annotations

import numpy as np

from . import layout

# embedding.py:51-73 (direction table; data, not code)
ICOSA_DIRS = np.array([
    0.8506508, 0, 0.5257311,
    0.809017, 0.5, 0.309017,
    0.5257311, 0.8506508, 0,
    1, 0, 0,
    0.809017, 0.5, -0.309017,
    0.8506508, 0, -0.5257311,
    0.309017, 0.809017, -0.5,
    0, 0.5257311, -0.8506508,
    0.5, 0.309017, -0.809017,
    0, 1, 0,
    -0.5257311, 0.8506508, 0,
    -0.309017, 0.809017, -0.5,
    0, 0.5257311, 0.8506508,
    -0.309017, 0.809017, 0.5,
    0.309017, 0.809017, 0.5,
    0.5, 0.309017, 0.809017,
    0.5, -0.309017, 0.809017,
    0, 0, 1,
    -0.5, 0.309017, 0.809017,
    -0.809017, 0.5, 0.309017,
    -0.809017, 0.5, -0.309017,
], dtype=np.float32).reshape(layout.N_DIRS, 3)


def make_params(n_obj: int, H: int, scale: float = 2.0, seed: int = 0, gain: float = 1.0):
    """Random-init parameters of ``n_obj`` object fields.

    Returns ``(fc, B, pe_scale)``: ``fc`` is a list of 14 float32 arrays with a leading object
    dimension, ``B`` is [n_obj, 21, 3], ``pe_scale`` is [n_obj].  ``gain`` multiplies the weight
    matrices (gain > 1 drives the occupancy logits into saturation for the edge-case tests).
    """
    rng = np.random.default_rng(seed)
    fc = []
    shapes = layout.fc_shapes(H)
    for t, shp in enumerate(shapes):
        if len(shp) == 2:
            fan_out, fan_in = shp
            std = np.sqrt(2.0 / (fan_in + fan_out))
            w = rng.standard_normal((n_obj,) + shp).astype(np.float32) * np.float32(std * gain)
            fc.append(w.astype(np.float32))
        else:
            fan_in = shapes[t - 1][1]
            bound = 1.0 / np.sqrt(fan_in)
            fc.append(rng.uniform(-bound, bound, (n_obj,) + shp).astype(np.float32))
    B = np.broadcast_to(ICOSA_DIRS, (n_obj, layout.N_DIRS, 3)).copy()
    # perturb the directions a little so that gradient tests are not run at a symmetric point
    B = (B + rng.standard_normal(B.shape).astype(np.float32) * np.float32(1e-3)).astype(np.float32)
    pe_scale = np.full((n_obj,), scale, dtype=np.float32)
    return fc, B, pe_scale


def make_batch(n_obj: int, R: int, S: int, seed: int = 1, n_cam2surf: int | None = None,
               invalid_frac: float = 0.05, eps: float = 0.1, other_eps: float = 0.05,
               max_depth: float = 4.5):
    """One step worth of ray samples for ``n_obj`` objects: the six tensors of ``train.py:271-277``.

    Returns dict with
      pcs [n,R,S,3] f32, z [n,R,S] f32, gt_depth [n,R] f32, gt_rgb [n,R,3] f32 in [0,1],
      sem [n,R] u8 in {0,1,2}, depth_mask [n,R] u8 in {0,1}.
    """
    rng = np.random.default_rng(seed)
    if n_cam2surf is None:
        n_cam2surf = 1 if S <= 10 else 5          # config: n_bins_cam2surface 1 (objects) / 5 (background)
    n_surf = S - n_cam2surf
    f32 = np.float32

    depth = rng.uniform(0.5, max_depth, (n_obj, R)).astype(f32)
    invalid = rng.uniform(0, 1, (n_obj, R)) < invalid_frac
    depth[invalid] = 0.0
    sem = rng.choice(np.array([0, 1, 2], dtype=np.uint8), size=(n_obj, R), p=[0.35, 0.6, 0.05])
    depth_mask = (~invalid).astype(np.uint8)

    def stratified(lo, hi, nb):
        # vmap.py:45-72: nb bins between lo and hi, one uniform sample per bin
        lo = lo[..., None].astype(f32)
        hi = hi[..., None].astype(f32)
        edges = np.linspace(0.0, 1.0, nb + 1, dtype=f32)[:-1]
        width = (hi - lo) / f32(nb)
        return (lo + (hi - lo) * edges + rng.uniform(0, 1, lo.shape[:-1] + (nb,)).astype(f32) * width).astype(f32)

    z = np.zeros((n_obj, R, S), dtype=f32)
    zeros = np.zeros((n_obj, R), dtype=f32)
    z_invalid = stratified(zeros, np.full((n_obj, R), depth.max(), dtype=f32), S)
    z_c2s = stratified(zeros, np.maximum(depth - f32(eps), f32(0.01)), n_cam2surf)
    normal = np.sort(np.clip(rng.standard_normal((n_obj, R, n_surf)).astype(f32) * f32(eps / 3.0),
                             -eps, eps), axis=-1).astype(f32)
    z_this = depth[..., None] + normal
    z_other = stratified(depth - f32(eps), depth + f32(other_eps), n_surf)
    z[..., :n_cam2surf] = z_c2s
    z[..., n_cam2surf:] = np.where((sem == 1)[..., None], z_this, z_other)
    z = np.where(invalid[..., None], z_invalid, z).astype(f32)

    u = rng.uniform(-1.0, 1.0, (n_obj, R)).astype(f32)
    v = rng.uniform(-0.57, 0.57, (n_obj, R)).astype(f32)
    dirs = np.stack([u, v, np.ones_like(u)], axis=-1)                     # camera-frame pixel rays
    origin = rng.uniform(-1.0, 1.0, (n_obj, R, 3)).astype(f32)
    centre = (origin + dirs * np.maximum(depth, f32(1.0))[..., None]).mean(axis=1, keepdims=True)
    pcs = (origin[:, :, None, :] + dirs[:, :, None, :] * z[..., None] - centre[:, :, None, :]).astype(f32)

    gt_rgb = (rng.integers(0, 256, (n_obj, R, 3)).astype(f32) / f32(255.0)).astype(f32)
    return {
        "pcs": np.ascontiguousarray(pcs), "z": np.ascontiguousarray(z),
        "gt_depth": depth, "gt_rgb": gt_rgb,
        "sem": sem.astype(np.uint8), "depth_mask": depth_mask,
    }


# BASELINE.json configs -> (n_obj, R, S, H, scale)
CONFIGS = {
    "imap_plumbing": dict(n_obj=1, R=100, S=14, H=256, scale=10.0),     # configs[0] as worded
    "imap_full": dict(n_obj=1, R=4800, S=14, H=256, scale=10.0),        # the reference's own iMAP batch (config_replica_room0_iMAP.json:
                                                                        # n_per_optim 4800, 9 + 5 bins) - not a BASELINE config; measurement
    "replica_room0_vmap": dict(n_obj=20, R=120, S=10, H=32, scale=2.0),  # configs[1] (headline)
    "scannet0024_vmap": dict(n_obj=50, R=120, S=10, H=32, scale=3.0),    # configs[3] shapes
    "stress_256x64": dict(n_obj=256, R=256, S=10, H=64, scale=2.0),      # configs[4] shapes
    "stress_rank8": dict(n_obj=32, R=256, S=10, H=64, scale=2.0),        # ... one rank's object shard of configs[4] on its 8 GPUs (measurement)
    "background": dict(n_obj=1, R=1200, S=14, H=128, scale=5.0),         # train.py:308-316
    "background_rank4": dict(n_obj=1, R=300, S=14, H=128, scale=5.0),    # ... one rank's ray shard of it at 4 / 8 GPUs (measurement:
    "background_rank8": dict(n_obj=1, R=150, S=14, H=128, scale=5.0),    # the latency of the ray-sharded step, parallel.SharedBackgroundHip)
}
