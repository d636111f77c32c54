"""Experiment: the 20 objects as G independent groups on G streams (dependency chains main -> finalize -> main overlap
across groups) vs one chain."""
import json
import os
import sys

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
res = {}
for G in (1, 2, 4):
    per = n // G
    groups = []
    for g in range(G):
        sl = slice(g * per, (g + 1) * per)
        op = step.VmapStep(per, R, S, H, device=dev, max_steps=ITERS)
        opt = step.FusedAdamWState(per, H, dev)
        args = ([t(a[sl].copy()) for a in fc], t(B[sl].copy()), t(sc[sl].copy()),
                *[t(fr[k][sl].copy()) for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask")])
        groups.append((op, opt, args, torch.cuda.Stream()))

    def frame():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(cur)
        for op, opt, args, st in groups:
            st.wait_event(ev)
            with torch.cuda.stream(st):
                op.train_steps(*args, opt=opt, n_steps=ITERS)
        for _, _, _, st in groups:
            e = torch.cuda.Event(); e.record(st); cur.wait_event(e)

    for _ in range(3):
        frame()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 30
    for _ in range(reps):
        frame()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    res[G] = {"ms_per_frame": ms, "us_per_step": ms / ITERS * 1e3, "rays_per_s": n * R * ITERS / ms * 1e3}
print(json.dumps(res))
