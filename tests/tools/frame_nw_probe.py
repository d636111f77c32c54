"""Round 5 probe: one mapping frame (20 object steps + 20 background steps on two streams) by the object stack's workgroups per object -
fewer workgroups keep the objects on the compute units the background's 200 one-per-CU workgroups leave free."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vmap_amd import step, synth
from vmap_amd.driver import HipMapper
from vmap_amd.trainer import SimpleConfig, Trainer
dev = torch.device("cuda:0")
cfg, bcfg, ipf = synth.CONFIGS["replica_room0_vmap"], synth.CONFIGS["background"], 20
frame = synth.make_batch(cfg["n_obj"], cfg["R"] * ipf, cfg["S"], seed=1)
obj_batch = tuple(torch.from_numpy(frame[k]).to(dev) for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask"))
bframe = synth.make_batch(1, bcfg["R"] * ipf, bcfg["S"], seed=77)
bg_batch = tuple(torch.from_numpy(bframe[k]).to(dev) for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask"))
out = []
for wpo in (0, 5, 4, 3, 2, 1):
    m = HipMapper(SimpleConfig(training_device=str(dev), n_iter_per_frame=ipf), device=dev)
    torch.manual_seed(3)
    for _ in range(cfg["n_obj"]):
        m.add_object(Trainer(SimpleConfig(training_device=str(dev), hidden_feature_size=cfg["H"], obj_scale=cfg["scale"])))
    m.attach_background(Trainer(SimpleConfig(training_device=str(dev), hidden_feature_size=bcfg["H"], obj_scale=bcfg["scale"])), bcfg["R"], bcfg["S"])
    m.restack(cfg["R"], cfg["S"])
    if wpo:
        m.op = step.VmapStep(cfg["n_obj"], cfg["R"], cfg["S"], cfg["H"], device=dev, max_steps=ipf, tuning={"workgroups_per_object": wpo})
    def timed(fn, reps=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    t_obj = timed(lambda: m.train_frame(*obj_batch))
    t_two = timed(lambda: m.train_frame_with_background(obj_batch, bg_batch))
    r = {"workgroups_per_object": wpo or "auto (10)", "plan": m.op.plan()["workgroups_per_object"], "objects_alone_ms_per_frame": round(t_obj, 4), "two_streams_ms_per_frame": round(t_two, 4)}
    print(json.dumps(r)); out.append(r)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "frame_nw_probe.json"), "w"), indent=1)
