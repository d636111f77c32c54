"""Measurement tool (round 4): how the stream topology of a frame (20 object steps + 20 background steps) decides its time.  One configuration
per PROCESS (hardware queues are handed to streams in creation order).  usage: frame_stream_topology_probe.py <case>"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vmap_amd import step, synth  # noqa: E402

case = sys.argv[1]
dev = torch.device("cuda:0")
ipf = 20


def setup(name):
    cfg = synth.CONFIGS[name]
    n, R, S, H = cfg["n_obj"], cfg["R"], cfg["S"], cfg["H"]
    fc, B, sc = synth.make_params(n, H, scale=cfg["scale"], seed=0)
    fr = synth.make_batch(n, R * ipf, S, seed=1)
    t = lambda a: torch.from_numpy(a).to(dev)
    op = step.VmapStep(n, R, S, H, device=dev, max_steps=ipf)
    opt = step.FusedAdamWState(n, H, dev)
    b = op.bind([t(a) for a in fc], t(B), t(sc), *[t(fr[k]) for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask")], opt=opt)
    return lambda: b.train_steps(ipf)


obj, bg = setup("replica_room0_vmap"), setup("background")
mk = lambda p: torch.cuda.Stream(device=dev, priority=p)
null = torch.cuda.default_stream(dev)
# case -> (caller stream, objects' stream, background stream, background issued first)
if case == "A":   s_call = mk(0); s_obj = s_call; s_bg = mk(-1); first = "bg"
elif case == "A2": s_call = mk(0); s_obj = s_call; s_bg = mk(-1); first = "obj"
elif case == "A0": s_call = mk(0); s_obj = s_call; s_bg = mk(0); first = "bg"
elif case == "B": s_call = null; s_obj = mk(0); s_bg = mk(-1); first = "bg"
elif case == "B2": s_call = null; s_bg = mk(-1); s_obj = mk(0); first = "bg"
elif case == "C": s_call = null; s_obj = null; s_bg = mk(-1); first = "bg"
elif case == "D": s_call = null; s_obj = mk(0); s_bg = mk(0); first = "bg"
elif case == "D0": s_call = null; s_obj = null; s_bg = mk(0); first = "bg"
elif case == "E": s_call = mk(0); s_obj = mk(0); s_bg = mk(-1); first = "bg"
elif case == "G": s_call = null; s_obj = mk(-1); s_bg = mk(0); first = "bg"
else: raise SystemExit("unknown case")


def frame():
    fork = torch.cuda.Event(); fork.record(s_call)
    order = [(s_bg, bg), (s_obj, obj)] if first == "bg" else [(s_obj, obj), (s_bg, bg)]
    joins = []
    for st, fn in order:
        if st is not s_call:
            st.wait_event(fork)
        with torch.cuda.stream(st):
            fn()
        if st is not s_call:
            j = torch.cuda.Event(); j.record(st); joins.append(j)
    for j in joins:
        s_call.wait_event(j)


ms = []
for rep in range(3):
    for _ in range(4):
        frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        frame()
    torch.cuda.synchronize()
    ms.append((time.perf_counter() - t0) / 30 * 1e3)
print(json.dumps({"case": case, "caller": "null" if s_call is null else f"created p{s_call.priority}", "objects": "null" if s_obj is null else ("caller's" if s_obj is s_call else f"created p{s_obj.priority}"),
                  "background": f"created p{s_bg.priority}", "issued_first": first, "ms_per_frame": ms}))
