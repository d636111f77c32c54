"""Generate the whole-frame trajectory fixtures by running the REAL reference's step loop on CPU.

Run in the authoring container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_frame_goldens.py [names...]

For every case in ``tests/cases.py:FRAME_CASES`` it runs ``oracle.ref_runner.reference_frame`` (train.py:270-326: strided
slices of the frame tensors, functorch vmap, loss.step_batch_loss, backward, AdamW(lr 1e-3, wd 0.013), zero_grad) in
float32 and stores the per-step losses, the gradients of the first step and the final parameters of the kept objects
(all objects for the headline frame), plus per-object / per-tensor L2 norms of the final parameters of every object; the
float64 twin contributes its losses (the tie-breaker for how fast the two precisions drift apart).
``<name>_bf16``: the same frame with bfloat16-rounded run-time weights over full-precision masters
(``reference_frame(weights_bf16=True)``), for the cases listed in ``cases.BF16_FRAME_CASES``.
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
    todo = names or (list(cases.FRAME_CASES) + [f"{n}_bf16" for n in cases.BF16_FRAME_CASES])
    for name in todo:
        bf16 = name.endswith("_bf16")
        c = cases.build_frame_case(name[:-5] if bf16 else name)
        keep = list(c["keep"])
        r32 = ref_runner.reference_frame(c["fc"], c["B"], c["scale"], c["frame"], c["H"], c["R"], c["n_steps"], torch.float32,
                                         weights_bf16=bf16)
        r64 = ref_runner.reference_frame(c["fc"], c["B"], c["scale"], c["frame"], c["H"], c["R"], c["n_steps"], torch.float64,
                                         weights_bf16=bf16)
        out = {"losses": r32["losses"], "f64_losses": r64["losses"], "keep": np.array(keep)}
        if name == "cfg2_frame20":
            # the reference's OTHER float32 path over the same frame (training_strategy "forloop", train.py:278-290): how far two
            # float32 evaluations of the reference itself drift apart over 20 steps - the yardstick for the kernel's own drift
            out["forloop_losses"] = ref_runner.reference_frame(c["fc"], c["B"], c["scale"], c["frame"], c["H"], c["R"], c["n_steps"],
                                                               torch.float32, strategy="forloop")["losses"]
        for t in list(range(14)) + ["B"]:
            k = f"fc{t}" if t != "B" else "B"
            out[f"p_{k}"] = r32[f"p_{k}"][keep].astype(np.float32)
            out[f"g0_{k}"] = r32[f"g0_{k}"][keep].astype(np.float32)
            n = r32[f"p_{k}"].shape[0]
            out[f"pnorm_{k}"] = np.sqrt((r32[f"p_{k}"].astype(np.float64).reshape(n, -1) ** 2).sum(-1))
            out[f"f64_pnorm_{k}"] = np.sqrt((r64[f"p_{k}"].astype(np.float64).reshape(n, -1) ** 2).sum(-1))
        # ---- the other side of a ReLU kink (round 4) ----
        # Where the reference's own float32 run of step 0 and the numpy oracle disagree on the derivative bit of a kink-adjacent hidden
        # unit, a second float32 implementation may follow either branch: find the bits (conftest.kink_aware on the first-step
        # gradients), re-run the SAME unmodified reference loop with the first step's gradients moved to the oracle's side of those
        # bits, and store that branch as alt_* next to the primary one.
        from conftest import GRAD_KEYS, kink_aware, round_bf16
        from oracle import vmap_oracle as vo
        rnd = round_bf16 if bf16 else (lambda a: a)
        sub = {k: np.ascontiguousarray(v[:, :c["R"]]) for k, v in c["frame"].items()}
        o = vo.training_step([rnd(a) for a in c["fc"]], rnd(c["B"]), c["scale"], sub, dtype=np.float32, kinks=True)
        ref0 = {(f"g_fc{t}" if t < 14 else "g_B"): r32[f"g0_fc{t}" if t < 14 else "g0_B"] for t in range(15)}
        corr, flipped, cand, worst = kink_aware(ref0, o, c["n"], signed=False, tol=1e-4)
        out["alt_flipped_bits"] = np.array(flipped)
        if flipped:
            delta = [np.asarray(o[k], np.float64) - corr[k] for k in GRAD_KEYS]           # reference side -> oracle side
            alt = ref_runner.reference_frame(c["fc"], c["B"], c["scale"], c["frame"], c["H"], c["R"], c["n_steps"], torch.float32,
                                             weights_bf16=bf16, first_step_grad_delta=[d.astype(np.float32) for d in delta])
            out["alt_losses"] = alt["losses"]
            for t in list(range(14)) + ["B"]:
                k = f"fc{t}" if t != "B" else "B"
                out[f"alt_p_{k}"] = alt[f"p_{k}"][keep].astype(np.float32)
                out[f"alt_pnorm_{k}"] = np.sqrt((alt[f"p_{k}"].astype(np.float64).reshape(alt[f"p_{k}"].shape[0], -1) ** 2).sum(-1))
            print(f"{name:16s} {flipped} kink bit(s) between the reference's run and the oracle: alternate branch stored, losses "
                  f"{alt['losses'][0]:.4f} .. {alt['losses'][-1]:.4f}; max rel separation {np.abs(alt['losses'] / r32['losses'] - 1).max():.2e}")
        out["torch_version"] = np.array(torch.__version__)
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name:16s} losses {r32['losses'][0]:.4f} .. {r32['losses'][-1]:.4f}  (f64 {r64['losses'][0]:.4f} .. {r64['losses'][-1]:.4f})"
              f"  -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main(sys.argv[1:] or None)
