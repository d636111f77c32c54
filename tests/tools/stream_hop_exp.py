"""Experiment: what does the per-frame hop to another stream and back (what torch.distributed's NCCL all_reduce does around
the flag reduction of the N-GPU path: record event -> collective on its own stream -> wait event) cost on the step loop?"""
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
cfg = synth.CONFIGS["replica_room0_vmap"]
n, R, S, H = cfg["n_obj"], cfg["R"], cfg["S"], cfg["H"]
fc, B, sc = synth.make_params(n, H, scale=cfg["scale"], seed=0)
fr = synth.make_batch(n, R * ITERS, S, seed=1)
t = lambda a: torch.from_numpy(a).to(dev)
tfc, tB, tsc = [t(a) for a in fc], t(B), t(sc)
fargs = tuple(t(fr[k]) for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask"))
op = step.VmapStep(n, R, S, H, device=dev, max_steps=ITERS)
opt = step.FusedAdamWState(n, H, dev)
side = torch.cuda.Stream()


def hop(flags):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        flags.add_(0)                      # stands in for the collective
    cur.wait_stream(side)


def same_stream(flags):
    flags.add_(0)


def timeit(fr_, reps=60):
    def frame():
        op.train_steps(tfc, tB, tsc, *fargs, opt=opt, n_steps=ITERS, flag_reduce=fr_)
    for _ in range(5):
        frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        frame()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


res = {}
for name, f in (("no_reduce", None), ("prepared_split_same_stream_op", same_stream), ("prepared_split_stream_hop", hop),
                ("no_reduce_again", None)):
    ms = timeit(f)
    res[name] = {"ms_per_frame": ms, "us_per_step": ms / ITERS * 1e3}
print(json.dumps(res))
