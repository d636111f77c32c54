"""Diagnostics: shader-clock stamps of the carried-finalize prologue (step_main_h32_carry), per workgroup."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vmap_amd import _lib, step, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "replica_room0_vmap"
cfg = synth.CONFIGS[name]
n, R, S, H = cfg["n_obj"], cfg["R"], cfg["S"], cfg["H"]
fc, B, sc = synth.make_params(n, H, scale=cfg["scale"], seed=0)
SLAB = os.environ.get('SLAB', '1') == '1'
frame = synth.make_batch(n, R * 20, S, seed=1)
dev = "cuda:0"
t = lambda a: torch.from_numpy(a).to(dev)
tfc, tB, tsc = [t(a) for a in fc], t(B), t(sc)
if SLAB:
    from vmap_amd import layout
    _, tfc, tB = layout.stack_in_slab(tfc, tB)
fr = {k: t(v) for k, v in frame.items()}
buf = torch.zeros(512 * 8, dtype=torch.int32, device=dev)
# two operators over the same state: the second one carries the finalize and stamps its prologue (per-operator tuning)
op = step.VmapStep(n, R, S, H, device=dev, max_steps=20, tuning={"carried_finalize": 1})
op_st = step.VmapStep(n, R, S, H, device=dev, max_steps=20, tuning={"carried_finalize": 1, "carry_stamps": buf.data_ptr()})
opt = step.FusedAdamWState(n, H, dev)
lib = _lib.load()
args = (tfc, tB, tsc, fr["pcs"], fr["z"], fr["gt_depth"], fr["gt_rgb"], fr["sem"], fr["depth_mask"])
for _ in range(3):
    op.train_steps(*args, opt=opt, n_steps=20)
op_st.train_steps(*args, opt=opt, n_steps=20)
torch.cuda.synchronize()
s = (buf.cpu().numpy().astype(np.int64) & 0xFFFFFFFF).reshape(512, 8)
s = s[s[:, 0] != 0]
names = ["start", "trip 0: first loads issued", "trip 0: partials summed", "trip 0 done", "trip 1 done", "all waves drained + barrier", "before wait-all", "after wait-all"]
d = s - s[:, :1]                      # per workgroup: clocks since its own start (the XCDs' counters are not aligned)
print(f"{len(s)} workgroups, shader clocks since each workgroup's own first stamp")
for i, nm in enumerate(names):
    col = d[:, i]
    print(f"{nm:30s} min {col.min():7d} median {int(np.median(col)):7d} p90 {int(np.quantile(col, 0.9)):7d} max {col.max():7d}")
