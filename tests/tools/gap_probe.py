"""Measurement tool (not product): where does the time of ONE short timed frame call go?  Replays bench.py's
`--steps 20 --warmup 5` sequence in one process and reports, per variant, the host time spent inside the frame call
(enqueue) and the wall time until the device is idle - for the marshalled-per-call and the bound call, cold and pre-heated.
Usage (GPU box): python tests/tools/gap_probe.py > gpurun_out/gap_probe.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vmap_amd import step, synth  # noqa: E402

cfg = synth.CONFIGS["replica_room0_vmap"]
n, R, S, H = cfg["n_obj"], cfg["R"], cfg["S"], cfg["H"]
ipf = 20
dev = torch.device("cuda:0")
fc, B, sc = synth.make_params(n, H, scale=cfg["scale"], seed=0)
frame = synth.make_batch(n, R * ipf, S, seed=1)
tfc = [torch.from_numpy(a).to(dev) for a in fc]
tB, tsc = torch.from_numpy(B).to(dev), torch.from_numpy(sc).to(dev)
fr = {k: torch.from_numpy(v).to(dev) for k, v in frame.items()}
fargs = (fr["pcs"], fr["z"], fr["gt_depth"], fr["gt_rgb"], fr["sem"], fr["depth_mask"])
op = step.VmapStep(n, R, S, H, device=dev, max_steps=ipf)
opt = step.FusedAdamWState(n, H, dev, lr=1e-3, weight_decay=0.013)
bound = op.bind(tfc, tB, tsc, *fargs, opt=opt)


def call(mode, k):
    if mode == "bound":
        bound.train_steps(k)
    else:
        op.train_steps(tfc, tB, tsc, *fargs, opt=opt, n_steps=k)


out = []
for mode in ("per_call", "bound"):
    for preheat_ms in (0.0, 50.0, 300.0):
        trials = []
        for trial in range(6):
            time.sleep(0.2)                       # the chip goes idle between trials, as it is in front of a fresh bench run
            t_ph = time.perf_counter()
            while (time.perf_counter() - t_ph) * 1e3 < preheat_ms:
                call(mode, ipf)
                torch.cuda.synchronize()
            call(mode, 5)                         # the driver's warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            call(mode, 20)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            trials.append({"enqueue_us": (t1 - t0) * 1e6, "total_us": (t2 - t0) * 1e6})
        tot = sorted(t["total_us"] for t in trials)
        out.append({"mode": mode, "preheat_ms": preheat_ms, "total_us_min": tot[0], "total_us_median": tot[len(tot) // 2],
                    "us_per_step_median": tot[len(tot) // 2] / 20, "enqueue_us_median": sorted(t["enqueue_us"] for t in trials)[3],
                    "trials": trials})
# steady state for reference: 20 frame calls back to back
for mode in ("per_call", "bound"):
    call(mode, 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        call(mode, 20)
    torch.cuda.synchronize()
    out.append({"mode": mode, "steady_state_us_per_step": (time.perf_counter() - t0) * 1e6 / 400})
print(json.dumps(out, indent=1))
