"""Experiment: one frame's step loop (20 x {step_main, step_finalize} + step_prep) replayed as ONE hipGraph vs
launched kernel by kernel.  Timing only: the captured AdamW bias-correction scalars are those of the captured frame."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vmap_amd import step, synth  # noqa: E402

dev = "cuda:0"
ITERS = 20
name = sys.argv[1] if len(sys.argv) > 1 else "replica_room0_vmap"
cfg = synth.CONFIGS[name]
n, R, S, H = cfg["n_obj"], cfg["R"], cfg["S"], cfg["H"]
fc, B, sc = synth.make_params(n, H, scale=cfg["scale"], seed=0)
fr = synth.make_batch(n, R * ITERS, S, seed=1)
t = lambda a: torch.from_numpy(a).to(dev)
tfc, tB, tsc = [t(a) for a in fc], t(B), t(sc)
fargs = tuple(t(fr[k]) for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask"))
op = step.VmapStep(n, R, S, H, device=dev, max_steps=ITERS)
opt = step.FusedAdamWState(n, H, dev)


def frame():
    op.train_steps(tfc, tB, tsc, *fargs, opt=opt, n_steps=ITERS)


def timeit(fn, reps=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


res = {"config": name}
ms = timeit(frame)
res["launches"] = {"ms_per_frame": ms, "us_per_step": ms / ITERS * 1e3, "rays_per_s": n * R * ITERS / ms * 1e3}
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    frame()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    frame()
ms = timeit(g.replay)
res["graph"] = {"ms_per_frame": ms, "us_per_step": ms / ITERS * 1e3, "rays_per_s": n * R * ITERS / ms * 1e3}
ms = timeit(frame)
res["launches_again"] = {"ms_per_frame": ms, "us_per_step": ms / ITERS * 1e3, "rays_per_s": n * R * ITERS / ms * 1e3}
print(json.dumps(res))
