"""Measurement tool (round 4): a frame (20 object steps + 20 background steps) with the two chains on streams restricted to DISJOINT sets
of compute units (hipExtStreamCreateWithCUMask): the background kernel runs 200 workgroups, one per compute unit; the objects' kernels get
the 56 units it leaves idle, so that neither chain's workgroups wait for the other's.  One mask layout per process: frame_cumask_probe.py <layout>"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vmap_amd import step, synth  # noqa: E402

layout = sys.argv[1] if len(sys.argv) > 1 else "none"
dev = torch.device("cuda:0")
ipf = 20
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(cus):
    """stream whose kernels may only run on the listed compute units (bit i of the 256-bit mask = unit i)"""
    words = (ctypes.c_uint32 * 8)()
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


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
allcu = list(range(256))
if layout == "none":
    s_obj, s_bg = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
elif layout == "xcd_major":            # units 32 x .. 32 x + 31 = XCD x: the last 7 of each XCD for the objects
    o = [32 * x + i for x in range(8) for i in range(25, 32)]
    s_obj, s_bg = masked_stream(o), masked_stream([c for c in allcu if c not in o])
elif layout == "interleaved":          # unit c on XCD c % 8: the last 56 indices = 7 per XCD
    o = list(range(200, 256))
    s_obj, s_bg = masked_stream(o), masked_stream([c for c in allcu if c not in o])
elif layout == "obj_only_xcd_major":
    o = [32 * x + i for x in range(8) for i in range(25, 32)]
    s_obj, s_bg = masked_stream(o), torch.cuda.Stream(device=dev)
elif layout == "obj_only_interleaved":
    s_obj, s_bg = masked_stream(list(range(200, 256))), torch.cuda.Stream(device=dev)
elif layout == "obj64":                # 64 / 192: the background's 200 workgroups then need two waves
    o = [32 * x + i for x in range(8) for i in range(24, 32)]
    s_obj, s_bg = masked_stream(o), masked_stream([c for c in allcu if c not in o])
else:
    raise SystemExit("unknown layout")
cur = torch.cuda.current_stream(dev)


def run(fn, st, reps):
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def frame():
    fork = torch.cuda.Event(); fork.record(cur)
    joins = []
    for st, fn in ((s_bg, bg), (s_obj, obj)):
        st.wait_event(fork)
        with torch.cuda.stream(st):
            fn()
        j = torch.cuda.Event(); j.record(st); joins.append(j)
    for j in joins:
        cur.wait_event(j)


t_obj, t_bg = run(obj, s_obj, 20), run(bg, s_bg, 20)
for _ in range(4):
    frame()
torch.cuda.synchronize()
ms = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(30):
        frame()
    torch.cuda.synchronize()
    ms.append((time.perf_counter() - t0) / 30 * 1e3)
print(json.dumps({"layout": layout, "objects_alone_ms_per_frame": t_obj, "background_alone_ms_per_frame": t_bg, "both_ms_per_frame": ms}))
