"""Generate the golden fixtures by running the REAL reference (kxhit/vMAP) on CPU.

Run in the authoring container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

For every case in ``tests/cases.py`` it runs ``oracle.ref_runner.reference_step`` (the reference's own
model.py / embedding.py / render_rays.py / loss.py driven through functorch exactly like
utils.py:30-34 + train.py:293-326) in float32, and again in float64 (the tie-breaker), and stores
loss, rendered depth/colour/opacity/variance and all 15 gradient tensors.  Inputs are not stored:
they are re-derived from the seeds, and ``input_sha256`` guards against generator drift.
``<name>_bf16`` (``cases.BF16_CASES``): the same case with every parameter tensor rounded to bfloat16 first
(the weight_dtype = bf16 mode of the library computes from exactly those values).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_runner  # noqa: E402
import cases  # noqa: E402


def main(names=None):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    for name in (names or (list(cases.CASES) + [f"{n}_bf16" for n in cases.BF16_CASES])):
        bf16 = name.endswith("_bf16")
        c = cases.build_case(name[:-5] if bf16 else name)
        if bf16:      # "bf16 weights + fp32 accumulate": the unmodified reference evaluated on bfloat16-rounded parameters
            rnd = lambda a: torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()
            c = dict(c, fc=[rnd(a) for a in c["fc"]], B=rnd(c["B"]))
        out = {}
        if name == "explode":
            # the reference ENDS THE PROCESS on this input (render_rays.py:88-90): certify that, then run it again with the exit call
            # recorded instead of obeyed (ref_runner.ExitTrap) to get the values a surviving caller must reproduce
            try:
                ref_runner.reference_step(c["fc"], c["B"], c["scale"], c["batch"], c["H"], torch.float32)
                raise AssertionError("the reference did not exit on the explode case")
            except SystemExit as e:
                out["exit_code"] = np.array(int(e.code))
            with ref_runner.ExitTrap() as trap:
                r32 = ref_runner.reference_step(c["fc"], c["B"], c["scale"], c["batch"], c["H"], torch.float32)
                out["explode_calls"] = np.array(len(trap.codes))
                r64 = ref_runner.reference_step(c["fc"], c["B"], c["scale"], c["batch"], c["H"], torch.float64)
        else:
            r32 = ref_runner.reference_step(c["fc"], c["B"], c["scale"], c["batch"], c["H"], torch.float32)
            r64 = ref_runner.reference_step(c["fc"], c["B"], c["scale"], c["batch"], c["H"], torch.float64)
        keep = ["loss", "render_depth", "render_color", "opacity", "var", "g_B"] + [f"g_fc{t}" for t in range(14)]
        for k in keep:
            out[k] = np.asarray(r32[k], dtype=np.float64 if k == "loss" else np.float32)
            out["f64_" + k] = np.asarray(r64[k], dtype=np.float64 if k == "loss" else np.float32)
        if name in ("tiny", "drop_depth"):
            # forloop strategy (train.py:278-290) must agree with vmap: pins the reference's own noise floor
            rf = ref_runner.reference_step(c["fc"], c["B"], c["scale"], c["batch"], c["H"], torch.float32,
                                           strategy="forloop")
            out["forloop_loss"] = np.asarray(rf["loss"], dtype=np.float64)
            out["forloop_g_fc4"] = rf["g_fc4"]
        if name in ("tiny", "scannet_scale"):
            ra = ref_runner.reference_step(c["fc"], c["B"], c["scale"], c["batch"], c["H"], torch.float32,
                                           adamw_steps=3)
            out["adamw_losses"] = ra["adamw_losses"]
            for t in range(14):
                out[f"adamw_p_fc{t}"] = ra[f"p_fc{t}"]
            out["adamw_p_B"] = ra["p_B"]
        out["input_sha256"] = np.array(cases.input_digest(c))
        out["torch_version"] = np.array(torch.__version__)
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name:14s} loss={float(out['loss']):.6f}  f64={float(out['f64_loss']):.6f}  "
              f"-> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main(sys.argv[1:] or None)
