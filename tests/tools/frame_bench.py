"""Measurement tool: one whole vMAP mapping frame on the HIP path at the Replica room0 vMAP shapes (train.py:195-338):
batched sampling of 20 objects, 20 optimisation steps of the 20 object fields (hidden 32) and of the background field
(hidden 128, 1200 rays x 14 samples), all device-resident; the per-frame times say where a real run spends its time."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vmap_amd import step, synth  # noqa: E402

dev = "cuda:0"
ITERS = 20


def setup(name):
    cfg = synth.CONFIGS[name]
    n, R, S, H = cfg["n_obj"], cfg["R"], cfg["S"], cfg["H"]
    fc, B, sc = synth.make_params(n, H, scale=cfg["scale"], seed=0)
    fr = synth.make_batch(n, R * ITERS, S, seed=1)
    t = lambda a: torch.from_numpy(a).to(dev)
    op = step.VmapStep(n, R, S, H, device=dev, max_steps=ITERS)
    opt = step.FusedAdamWState(n, H, dev)
    args = ([t(a) for a in fc], t(B), t(sc), *[t(fr[k]) for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask")])
    return lambda: op.train_steps(*args, opt=opt, n_steps=ITERS), n * R


obj, obj_rays = setup("replica_room0_vmap")
bg, bg_rays = setup("background")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t_obj, t_bg = timed(obj), timed(bg)
t_both = timed(lambda: (obj(), bg()))
s_obj, s_bg = torch.cuda.Stream(), torch.cuda.Stream()


def overlapped():
    cur = torch.cuda.current_stream()
    e = torch.cuda.Event(); e.record(cur)
    for st, fn in ((s_bg, bg), (s_obj, obj)):
        st.wait_event(e)
        with torch.cuda.stream(st):
            fn()
        d = torch.cuda.Event(); d.record(st); cur.wait_event(d)


t_overlap = timed(overlapped)
print(json.dumps({"tool": "frame_bench", "iters_per_frame": ITERS,
                  "objects_ms_per_frame": t_obj, "background_ms_per_frame": t_bg, "objects_plus_background_ms_per_frame": t_both, "objects_and_background_on_two_streams_ms_per_frame": t_overlap,
                  "object_rays_per_s": obj_rays * ITERS / t_obj * 1e3, "background_rays_per_s": bg_rays * ITERS / t_bg * 1e3,
                  "note": "sampler 0.10 ms/frame (profiles/r01_sampler_bench.json) not included"}))
