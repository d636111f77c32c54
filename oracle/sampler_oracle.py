"""TEST INFRASTRUCTURE ONLY - numpy restatement of vMAP's depth-guided ray / sample-point sampler for ONE object.

Follows reference ``vmap.py:319-364`` (``sceneObject.get_training_samples``) and ``vmap.py:366-459``
(``sample_3d_points``) with helpers ``stratified_bins`` (``:45-72``), ``normal_bins_sampling`` (``:75-87``),
``origin_dirs_W`` (``:31-41``) and the pixel-ray table of ``cameraInfo.get_rays_dirs`` (``:507-524``).

The reference draws its random numbers from torch's global generator in a data-dependent order (masked subsets).  To
make the deterministic part comparable bit-for-bit, this restatement takes the random numbers as PER-RAY arrays:

  kf_ids  int   [F]        keyframe slot of every sampled frame (already including the forced latest two)
  u_w,u_h f32   [F, P]     uniforms for the pixel position inside the keyframe's 2-D box
  u_z     f32   [F*P, S]   uniforms of the stratified bins (first n1 columns: camera-to-surface bins)
  g_z     f32   [F*P, n2]  standard normals of the surface samples of rays that hit this object

``tests/golden/make_sampler_goldens.py`` runs the real reference with torch's RNG entry points replaced by functions
that replay exactly these arrays (compacted the way the reference indexes them) and stores its outputs; the HIP
sampler consumes the same per-ray arrays in its test mode.  Imported only by tests/.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def linspace01(nb: int) -> np.ndarray:
    """torch.linspace(0, 1, nb + 1, dtype=float32) (ATen: step in float32, symmetric evaluation from both ends)."""
    steps = nb + 1
    step = f32(1.0) / f32(steps - 1)
    out = np.empty(steps, dtype=f32)
    half = steps // 2
    for i in range(steps):
        out[i] = f32(0.0) + step * f32(i) if i < half else f32(1.0) - step * f32(steps - 1 - i)
    return out


def stratified_bins(lo, hi, nb, u):
    """vmap.py:45-72: one sample per bin; lo/hi [n] (or scalars), u [n, nb] uniforms."""
    n = u.shape[0]
    lo = np.broadcast_to(np.asarray(lo, dtype=f32), (n,)).astype(f32)
    hi = np.broadcast_to(np.asarray(hi, dtype=f32), (n,)).astype(f32)
    limits = linspace01(nb)
    rng = (hi - lo).astype(f32)
    lower = (rng[:, None] * limits[None, :] + lo[:, None]).astype(f32)[:, :-1]
    length = (rng / f32(nb)).astype(f32)
    return (lower + (u.astype(f32) * length[:, None]).astype(f32)).astype(f32)


def sample_object(rgbs, depth, t_wc, bbox, kf_ids, u_w, u_h, u_z, g_z, intr, center, n1, n2,
                  min_bound=0.0, eps=0.1, stop_eps=0.05, this_obj=1):
    """One object, one frame worth of samples.  rgbs u8 [K,W,H,4], depth f32 [K,W,H], t_wc f32 [K,4,4],
    bbox f32 [K,4] (u lo, u hi, v lo, v hi), intr = (fx, fy, cx, cy), center f32 [3].
    Returns dict(rgb u8 [F*P,3], depth [F*P], valid [F*P] bool, labels u8 [F*P], pcs [F*P,S,3], z [F*P,S])."""
    F, P = u_w.shape
    S = n1 + n2
    fx, fy, cx, cy = (f32(v) for v in intr)
    kf = np.asarray(kf_ids, dtype=np.int64)
    b = bbox[kf].astype(f32)                                                  # [F,4]
    iw = (u_w.astype(f32) * (b[:, 1] - b[:, 0])[:, None] + b[:, 0][:, None]).astype(f32)     # vmap.py:347
    ih = (u_h.astype(f32) * (b[:, 3] - b[:, 2])[:, None] + b[:, 2][:, None]).astype(f32)     # :348
    iw = iw.astype(np.int64)                                                  # :350 .long() truncates
    ih = ih.astype(np.int64)
    srgb = rgbs[kf[:, None], iw, ih]                                          # :353  [F,P,4]
    sdep = depth[kf[:, None], iw, ih].astype(f32)                             # :354
    dirs = np.stack([(iw.astype(f32) - cx) / fx, (ih.astype(f32) - cy) / fy, np.ones_like(iw, dtype=f32)], -1).astype(f32)  # :512-516
    T = t_wc[kf].astype(f32)                                                  # :360
    dirs_w = np.einsum("fij,fpj->fpi", T[:, :3, :3], dirs).astype(f32)        # :38  R @ dir
    origins = T[:, :3, 3]                                                     # :40

    d = sdep.reshape(-1)
    state = srgb[..., 3].reshape(-1)
    invalid = d <= f32(min_bound)                                             # :389
    valid = ~invalid
    z = np.zeros((F * P, S), dtype=f32)
    max_bound = d.max()                                                       # :391
    if invalid.any():
        z[invalid] = stratified_bins(f32(min_bound), max_bound, S, u_z[invalid])               # :395-399
    if valid.any():
        z[valid, :n1] = stratified_bins(f32(min_bound), d[valid] - f32(eps), n1, u_z[valid][:, :n1])   # :408-410
        obj = (state == this_obj) & valid                                     # :413
        if obj.any():
            bins = np.sort((g_z[obj].astype(f32) * f32(eps / 3.0)).astype(f32), axis=-1)       # :81 normal_, sort
            bins = np.clip(bins, f32(-eps), f32(eps))                          # :82
            z[obj, n1:] = (d[obj][:, None] + bins).astype(f32)                # :83
        other = (state != this_obj) & valid                                   # :438
        if other.any():
            z[other, n1:] = stratified_bins(d[other] - f32(eps), d[other] + f32(stop_eps), n2, u_z[other][:, n1:])   # :441-445
    zz = z.reshape(F, P, S)
    pcs = (origins[:, None, None, :] + dirs_w[:, :, None, :] * zz[..., None]).astype(f32)      # :452-453
    pcs = (pcs - np.asarray(center, dtype=f32)).astype(f32)                   # :454
    return dict(rgb=srgb[..., :3].reshape(F * P, 3), depth=d, valid=valid, labels=state.astype(np.uint8),
                pcs=pcs.reshape(F * P, S, 3), z=z)
