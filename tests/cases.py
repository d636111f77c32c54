"""Named, seeded parity cases shared by the golden generator and the tests.

Each case is (parameters, batch) built by ``vmap_amd.synth`` plus, for the quirk cases, a
deterministic edit that forces one of the reference's data-dependent branches
(SURVEY.md section 8(c) known-answer list).
"""
from __future__ import annotations

import hashlib

import numpy as np

from vmap_amd import synth

CASES = {
    # name: (n_obj, R, S, H, scale, param_seed, batch_seed, gain, edit)
    "tiny":          (3, 12, 10, 32, 2.0, 10, 11, 1.0, None),
    "ragged":        (5, 17, 10, 32, 2.0, 12, 13, 1.0, None),        # R*S not a multiple of 32
    "cfg2":          (20, 120, 10, 32, 2.0, 20, 21, 1.0, None),      # BASELINE configs[1]
    "drop_depth":    (4, 24, 10, 32, 2.0, 30, 31, 1.0, "no_depth"),  # one object without valid depth
    "drop_opacity":  (4, 24, 10, 32, 2.0, 32, 33, 1.0, "all_unknown"),
    "drop_colour":   (4, 24, 10, 32, 2.0, 34, 35, 1.0, "all_other"),
    "saturated":     (3, 24, 10, 32, 2.0, 40, 41, 4.0, None),        # |alpha| >> 17: occupancy == 1.0f
    "exact_hit":     (2, 12, 10, 32, 2.0, 42, 43, 1.0, "no_valid_surface"),
    "explode":       (3, 24, 10, 32, 2.0, 44, 45, 4.0, "far_depth"),   # render_rays.py:88-90 fires: saturated occupancy (var -> 0: weight 1e4) x a 50 m depth error
    "h64":           (4, 32, 10, 64, 2.0, 50, 51, 1.0, None),
    "bg_h128_s14":   (1, 48, 14, 128, 5.0, 60, 61, 1.0, None),       # train.py:308-316 shapes (fewer rays)
    "imap_h256":     (1, 100, 14, 256, 10.0, 70, 71, 1.0, None),     # BASELINE configs[0]
    "scannet_scale": (6, 120, 10, 32, 3.0, 80, 81, 1.0, None),       # configs[3] obj_scale
    "imap_full":     (1, 4800, 14, 256, 10.0, 72, 73, 1.0, None),    # the reference's OWN iMAP batch (config_replica_room0_iMAP.json:31 n_per_optim 4800):
                                                                     # 2400 single-tile rounds on 240 workgroups - the multi-round step_main_ws<8> form
}


# step fixtures that also exist on bfloat16-rounded parameters (<name>_bf16.npz): BASELINE configs[3] / [4] weight mode
BF16_CASES = ("scannet_scale", "h64", "bg_h128_s14", "imap_full")


def build_case(name):
    n, R, S, H, scale, ps, bs, gain, edit = CASES[name]
    fc, B, pe_scale = synth.make_params(n, H, scale=scale, seed=ps, gain=gain)
    batch = synth.make_batch(n, R, S, seed=bs)
    if edit == "no_depth":
        batch["depth_mask"][2, :] = 0
    elif edit == "all_unknown":
        batch["sem"][1, :] = 2
    elif edit == "all_other":
        batch["sem"][3, :] = 0
    elif edit == "far_depth":
        # object 1's measured depths replaced by 50 m (its samples stay where they were): |D - d| / (sqrt(var) + 1e-4) ~ 5e5 per ray
        batch["gt_depth"][1, :] = np.where(batch["gt_depth"][1, :] > 0, np.float32(50.0), batch["gt_depth"][1, :]).astype(np.float32)
    elif edit == "no_valid_surface":
        # every ray of object 0 is 'other object' with invalid depth: depth term has an empty mask
        batch["sem"][0, :] = 0
    return dict(name=name, n=n, R=R, S=S, H=H, fc=fc, B=B, scale=pe_scale, batch=batch)


def input_digest(case) -> str:
    h = hashlib.sha256()
    for a in case["fc"]:
        h.update(np.ascontiguousarray(a).tobytes())
    h.update(case["B"].tobytes())
    h.update(case["scale"].tobytes())
    for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask"):
        h.update(np.ascontiguousarray(case["batch"][k]).tobytes())
    return h.hexdigest()


# Whole-frame trajectory cases: the reference's own step loop (train.py:270-326) over [n, n_steps * R, ...] frame tensors.
FRAME_CASES = {
    # name: (n_obj, R, S, H, scale, param_seed, frame_seed, n_steps, objects whose final parameters are stored)
    "cfg2_frame20":   (20, 120, 10, 32, 2.0, 20, 121, 20, None),            # BASELINE configs[1], the frame bench.py times
    "scannet50_frame": (50, 120, 10, 32, 3.0, 90, 191, 4, (0, 7, 23, 49)),  # configs[3]: 50 objects -> multi-pass step_main at real n
    "h64_r256_frame": (32, 256, 10, 64, 2.0, 92, 193, 3, (0, 13, 31)),      # configs[4] per-GPU shape (256 objects / 8 GPUs)
    "bg128_frame":    (1, 240, 14, 128, 5.0, 94, 195, 4, None),             # the background model (train.py:308-316): hidden 128, 14 samples; 60 rounds per step
}


# The same frames with "bf16 weights + fp32 accumulate" (BASELINE configs[3] / [4]): fp32 masters, every step computed from
# the masters rounded to bfloat16 (oracle/ref_runner.reference_frame(weights_bf16=True)); fixture <name>_bf16.npz.
BF16_FRAME_CASES = ("scannet50_frame", "h64_r256_frame", "bg128_frame")


def build_frame_case(name):
    n, R, S, H, scale, ps, bs, n_steps, keep = FRAME_CASES[name]
    fc, B, pe_scale = synth.make_params(n, H, scale=scale, seed=ps)
    frame = synth.make_batch(n, R * n_steps, S, seed=bs)
    return dict(name=name, n=n, R=R, S=S, H=H, n_steps=n_steps, fc=fc, B=B, scale=pe_scale, frame=frame,
                keep=tuple(range(n)) if keep is None else tuple(keep))
