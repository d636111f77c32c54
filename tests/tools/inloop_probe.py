"""Measurement tool: the dominant kernel's dispatch time INSIDE the step loop (prep / main / finalize sequence with the fused AdamW),
once with the frame's steps walking through the frame's rays (ray_step = R: every step reads samples nobody touched since the last
frame) and once with every step on the SAME rays (ray_step = 0: the samples stay in L2) - what the cold samples cost the prologue.
    python tests/tools/inloop_probe.py [config]"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vmap_amd import _lib, step, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "replica_room0_vmap"
cfg = synth.CONFIGS[name]
n, R, S, H = cfg["n_obj"], cfg["R"], cfg["S"], cfg["H"]
ipf = 20
fc, B, sc = synth.make_params(n, H, scale=cfg["scale"], seed=0)
frame = synth.make_batch(n, R * ipf, S, seed=1)
dev = torch.device("cuda:0")
tfc = [torch.from_numpy(a).to(dev) for a in fc]
tB, tsc = torch.from_numpy(B).to(dev), torch.from_numpy(sc).to(dev)
fr = [torch.from_numpy(frame[k]).to(dev) for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask")]
op = step.VmapStep(n, R, S, H, device=dev, max_steps=ipf)
opt = step.FusedAdamWState(n, H, dev, lr=1e-3, weight_decay=0.013)
pp = op._params(tfc, tB)
scs = _lib.Tensor(tsc.data_ptr(), tsc.stride(0))
bt = op._batch(*fr, rays_total=fr[0].shape[1])
res, out = op._outputs(ipf, False)
rows = {}
for label, ray_step in (("walking (ray_step = R)", R), ("same rays every step (ray_step = 0)", 0), ("walking (ray_step = R)", R), ("same rays every step (ray_step = 0)", 0)):
    tot = 0.0
    for rep in range(12):
        oc = opt.c_struct()
        ms = (ctypes.c_float * 2)(0.0, 0.0)
        _lib.check(lib=op.lib, rc=op.lib.vmapstep_profile_train_steps(ctypes.byref(op.shape), ctypes.byref(pp), ctypes.byref(scs), ctypes.byref(bt), ray_step, ipf,
                                                                       op.color_scaling, op.opacity_scaling, ctypes.byref(oc), ctypes.byref(out), op._ws_ptr, op._ws_bytes, op._stream(), ms))
        opt.step += ipf
        opt.note_host_steps(ipf)
        if rep >= 2:
            tot += float(ms[0])
    rows.setdefault(label, []).append(tot / 10 * 1e3)
print(json.dumps({"config": name, "kernel": op.plan()["kernel"], "in_loop_kernel_us": rows}))
