"""Measurement tool: N training steps of a BASELINE config (used under rocprofv3 --pmc)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vmap_amd import step, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "replica_room0_vmap"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
weights = sys.argv[3] if len(sys.argv) > 3 else "f32"
cfg = synth.CONFIGS[name]
n, R, S, H = cfg["n_obj"], cfg["R"], cfg["S"], cfg["H"]
fc, B, sc = synth.make_params(n, H, scale=cfg["scale"], seed=0)
frame = synth.make_batch(n, R * 20, S, seed=1)
dev = "cuda:0"
tfc = [torch.from_numpy(a).to(dev) for a in fc]
tB, tsc = torch.from_numpy(B).to(dev), torch.from_numpy(sc).to(dev)
fr = {k: torch.from_numpy(v).to(dev) for k, v in frame.items()}
op = step.VmapStep(n, R, S, H, device=dev, max_steps=20, weights=weights)
opt = step.FusedAdamWState(n, H, dev)
done = 0
while done < steps:
    k = min(20, steps - done)
    op.train_steps(tfc, tB, tsc, fr["pcs"], fr["z"], fr["gt_depth"], fr["gt_rgb"], fr["sem"], fr["depth_mask"], opt=opt, n_steps=k)
    done += k
torch.cuda.synchronize()
print("done", done)
