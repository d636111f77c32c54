"""Measurement tool: the dominant kernel of a configuration launched back to back on FIXED parameters (no optimiser, so a measurement
build whose gradients are wrong on purpose - tests/tools/build_variant.py ... -DVS_ABL=... - still runs the same forward every time):
    VMAPSTEP_LIBRARY=tests/tools/libvmapstep_<tag>.so python tests/tools/abl_probe.py [config] [weights]
prints one JSON line {library, config, kernel, kernel_us}."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vmap_amd import _lib, step, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "replica_room0_vmap"
weights = sys.argv[2] if len(sys.argv) > 2 else "f32"
kern = sys.argv[3] if len(sys.argv) > 3 else "auto"          # auto | bwd6 | f32
cfg = synth.CONFIGS[name]
n, R, S, H = cfg["n_obj"], cfg["R"], cfg["S"], cfg["H"]
fc, B, sc = synth.make_params(n, H, scale=cfg["scale"], seed=0)
b = synth.make_batch(n, R, S, seed=1)
dev = torch.device("cuda:0")
tfc = [torch.from_numpy(a).to(dev) for a in fc]
tB, tsc = torch.from_numpy(B).to(dev), torch.from_numpy(sc).to(dev)
tb = [torch.from_numpy(b[k]).to(dev) for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask")]
tuning = {"bwd6": {"kernel": _lib.KERNEL_S32_BWD6}, "f32": {"kernel": _lib.KERNEL_H32_F32}}.get(kern)
op = step.VmapStep(n, R, S, H, device=dev, max_steps=20, weights=weights, tuning=tuning)
for _ in range(3):
    ms = op.profile_main_kernel(tfc, tB, tsc, *tb, reps=300)
print(json.dumps({"library": os.path.relpath(_lib.LIB_PATH, ROOT), "config": name, "weights": weights, "kernel": op.plan()["kernel"], "kernel_us": ms * 1e3}))
