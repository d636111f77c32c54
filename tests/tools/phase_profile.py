"""Measurement tool: in-kernel phase breakdown of step_main_h32 (shader clocks) for a BASELINE config."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vmap_amd import step, synth  # noqa: E402

NAMES = ["start", "staged+cb zeroed", "encoding", "mlp fwd (5 layers)", "heads+cb write", "barrier B", "composite+barrier C",
         "bwd heads + dW colour", "bwd d4 + d e2", "bwd mid2", "bwd cat", "bwd mid1", "bwd in + enc", "bwd dB",
         "final barrier", "partials written"]
NAMES_WS = ["start", "encoding + barrier", "in_layer", "mid1", "cat_layer", "mid2", "color_linear", "heads + composite",
            "bwd: enc F images, heads delta", "bwd: heads dW, delta 0", "bwd color_linear", "bwd mid2", "bwd cat_layer", "bwd mid1",
            "bwd in_layer + d(proj)", "bwd dB"]
name = sys.argv[1] if len(sys.argv) > 1 else "replica_room0_vmap"
cfg = synth.CONFIGS[name]
n, R, S, H = cfg["n_obj"], cfg["R"], cfg["S"], cfg["H"]
if H in (64, 128):
    NAMES = NAMES_WS
fc, B, sc = synth.make_params(n, H, scale=cfg["scale"], seed=0)
batch = synth.make_batch(n, R, S, seed=1)
dev = "cuda:0"
tfc = [torch.from_numpy(a).to(dev) for a in fc]
tB, tsc = torch.from_numpy(B).to(dev), torch.from_numpy(sc).to(dev)
tb = {k: torch.from_numpy(v).to(dev) for k, v in batch.items()}
from vmap_amd import _lib  # noqa: E402
# the phase-stamp instantiations live in the measurement build of the library (tests/tools/libvmapstep_ab.so, built by build())
AB_LIBRARY = os.path.join(ROOT, "tests", "tools", "libvmapstep_ab.so")
kern = sys.argv[2] if len(sys.argv) > 2 else "split"       # split (default kernel at hidden 32) | f32
ws_flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # hidden 128: tuning.ws_flags (4 = never three-tile rounds)
op = step.VmapStep(n, R, S, H, device=dev, tuning={"kernel": _lib.KERNEL_H32_F32} if kern == "f32" else {"ws_flags": ws_flags}, library=AB_LIBRARY)
args = (tfc, tB, tsc, tb["pcs"], tb["z"], tb["gt_depth"], tb["gt_rgb"], tb["sem"], tb["depth_mask"])
for _ in range(3):
    t = op.profile_phases(*args)
t = op.profile_phases(*args).astype(np.float64)          # [WG, 4 waves, 16]
d = np.diff(t, axis=-1)
print(f"config {name} kernel {kern} ws_flags {ws_flags}: {t.shape[0]} workgroups; kernel span {t.max():.0f} clocks; per-phase clocks (median / p90 / max over waves)")
tot = 0.0
for i in range(15):
    x = d[:, :, i].ravel()
    print(f"  {i:2d}->{i+1:2d} {NAMES[i+1]:26s} {np.median(x):9.0f} {np.quantile(x, 0.9):9.0f} {x.max():9.0f}")
    tot += np.median(x)
print(f"  sum of medians {tot:.0f}; wave-0 first stamp spread across WGs {np.ptp(t[:, 0, 0]):.0f}; "
      f"last stamp median {np.median(t[:, :, 15]):.0f} max {t[:, :, 15].max():.0f}")
json.dump({"config": name, "median": np.median(d.reshape(-1, 15), axis=0).tolist(), "names": NAMES[1:]},
          open(os.path.join(ROOT, "gpurun_out", f"phases_{name}_{kern}_{ws_flags}.json"), "w"))
if H == 128:
    print("  per wave (median over workgroups):")
    for w in range(4):
        print(f"    wave {w}: " + " ".join(f"{np.median(d[:, w, i]):7.0f}" for i in range(15)))
