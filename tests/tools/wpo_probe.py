"""Round 5 probe: configs[3] (50 objects x 120 rays, hidden 32) by workgroups per object - the automatic plan (5: 250 workgroups, no XCD-affine
map because 7 object groups x 5 = 35 workgroups per XCD exceed its 32 CUs) against 4 (200 workgroups, XCD-affine) and others."""
import sys, time, json, numpy as np, torch
sys.path.insert(0, '/root/repo')
from vmap_amd import step, synth
dev = "cuda:0"; ipf = 20
cfg = synth.CONFIGS["scannet0024_vmap"]; n, R, S, H = cfg["n_obj"], cfg["R"], cfg["S"], cfg["H"]
fc, B, sc = synth.make_params(n, H, scale=cfg["scale"], seed=5)
frame = synth.make_batch(n, R * ipf, S, seed=6)
for weights in ("bf16", "f32"):
    for wpo in (0, 5, 4, 3, 6):
        tfc = [torch.from_numpy(a).to(dev) for a in fc]; tB, tsc = torch.from_numpy(B).to(dev), torch.from_numpy(sc).to(dev)
        fr = tuple(torch.from_numpy(frame[k]).to(dev) for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask"))
        op = step.VmapStep(n, R, S, H, device=dev, max_steps=ipf, weights=weights, tuning={"workgroups_per_object": wpo} if wpo else None)
        opt = step.FusedAdamWState(n, H, dev, lr=1e-3, weight_decay=0.013)
        bound = op.bind(tfc, tB, tsc, *fr, opt=opt)
        for _ in range(3): bound.train_steps(ipf)
        torch.cuda.synchronize(); ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(20): bound.train_steps(ipf)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 400 * 1e3)
        p = op.plan()
        print(json.dumps({"weights": weights, "workgroups_per_object_asked": wpo, "plan": {k: p[k] for k in ("kernel", "workgroups_per_object", "rounds_per_object") if k in p},
                          "ms_per_step_median": sorted(ts)[2], "rays_per_s": n * R / (sorted(ts)[2] * 1e-3)}), flush=True)
