"""Measurement tool (round 4): one mapping frame (20 object steps + 20 background steps on two streams, driver.HipMapper) with the
background stream at default / high priority and the objects' at default / low - does the dispatcher let the object workgroups fill the
56 compute units the background kernel (200 workgroups) leaves idle instead of delaying it?"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vmap_amd import synth  # noqa: E402
from vmap_amd.driver import HipMapper  # noqa: E402
from vmap_amd.trainer import SimpleConfig, Trainer  # noqa: E402

dev = torch.device("cuda:0")
ipf = 20
cfg, bcfg = synth.CONFIGS["replica_room0_vmap"], synth.CONFIGS["background"]
keys = ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask")
of = synth.make_batch(cfg["n_obj"], cfg["R"] * ipf, cfg["S"], seed=1)
bf = synth.make_batch(1, bcfg["R"] * ipf, bcfg["S"], seed=77)
ob = tuple(torch.from_numpy(of[k]).to(dev) for k in keys)
bb = tuple(torch.from_numpy(bf[k]).to(dev) for k in keys)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
out = []
streams = {p: (torch.cuda.Stream(device=dev, priority=p), torch.cuda.Stream(device=dev, priority=p)) for p in (0, -1)}
m = HipMapper(SimpleConfig(training_device=str(dev), n_iter_per_frame=ipf), device=dev)
torch.manual_seed(3)
for _ in range(cfg["n_obj"]):
    m.add_object(Trainer(SimpleConfig(training_device=str(dev), hidden_feature_size=cfg["H"], obj_scale=cfg["scale"])))
m.attach_background(Trainer(SimpleConfig(training_device=str(dev), hidden_feature_size=bcfg["H"], obj_scale=bcfg["scale"])), bcfg["R"], bcfg["S"])
for rep in range(3):
    for bg_prio, obj_prio in [(0, 0), (0, -1), (-1, 0), (-1, -1), (-1, None), (0, None)]:
        m._bg_stream = streams[bg_prio][0]
        obj_stream = streams[obj_prio][1] if obj_prio is not None else torch.cuda.default_stream(dev)     # None: the device's NULL stream
        with torch.cuda.stream(obj_stream):                   # the mapper runs the objects on the caller's current stream
            for _ in range(4):
                m.train_frame_with_background(ob, bb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                m.train_frame_with_background(ob, bb)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 30 * 1e3
        rec = {"rep": rep, "background_stream_priority": bg_prio, "object_stream_priority": obj_prio, "two_streams_ms_per_frame": ms}
        print(json.dumps(rec), flush=True)
        out.append(rec)
# the driver's own default (background stream high priority, objects on its own normal-priority stream, caller on the NULL stream)
m2 = HipMapper(SimpleConfig(training_device=str(dev), n_iter_per_frame=ipf), device=dev)
torch.manual_seed(3)
for _ in range(cfg["n_obj"]):
    m2.add_object(Trainer(SimpleConfig(training_device=str(dev), hidden_feature_size=cfg["H"], obj_scale=cfg["scale"])))
m2.attach_background(Trainer(SimpleConfig(training_device=str(dev), hidden_feature_size=bcfg["H"], obj_scale=bcfg["scale"])), bcfg["R"], bcfg["S"])
for rep in range(3):
    for _ in range(4):
        m2.train_frame_with_background(ob, bb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        m2.train_frame_with_background(ob, bb)
    torch.cuda.synchronize()
    rec = {"rep": rep, "driver_default": "HipMapper.train_frame_with_background called on the NULL stream", "two_streams_ms_per_frame": (time.perf_counter() - t0) / 30 * 1e3}
    print(json.dumps(rec), flush=True)
    out.append(rec)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "frame_priority_probe.json"), "w"), indent=1)
