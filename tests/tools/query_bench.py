"""Measurement tool: HIP inference query (vmapstep_query_points) vs the module's own eager PyTorch-ROCm forward in the
reference's chunking (trainer.py:77-95, chunk 100 000) on a mesh-extraction sized grid (render_rays.py:98-122)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vmap_amd import layout  # noqa: E402
from vmap_amd.trainer import SimpleConfig, Trainer  # noqa: E402

torch.manual_seed(0)
res = {"tool": "query_bench", "device": torch.cuda.get_device_properties(0).gcnArchName, "grids": []}
for H, dim in ((32, 100), (32, 256), (128, 100), (128, 256), (256, 100)):
    tr = Trainer(SimpleConfig(training_device="cuda:0", hidden_feature_size=H))
    flop_pt = 2 * (layout.EMB1 * H + H * H + (H + layout.EMB1) * H + H * H + (H + layout.EMB2) * H + H + 3 * H)
    n = dim ** 3
    pts = torch.rand(n, 3, device="cuda") * 2 - 1
    tr.eval_points(pts[:4096]); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        occ, col = tr._eval_points_hip(pts)
    e1.record(); torch.cuda.synchronize()
    hip_ms = e0.elapsed_time(e1) / reps
    with torch.no_grad():
        def eager():
            al, co = [], []
            for k in range(0, n, 100000):
                a, c = tr.fc_occ_map(tr.pe(pts[k:k + 100000]))
                al.append(a.squeeze(-1)); co.append(c)
            return torch.sigmoid(torch.cat(al)), torch.cat(co)
        eager(); torch.cuda.synchronize()
        e0.record(); ro, rc = eager(); e1.record(); torch.cuda.synchronize()
    eager_ms = e0.elapsed_time(e1)
    res["grids"].append({"hidden": H, "grid_dim": dim, "points": n, "hip_ms": hip_ms, "eager_torch_ms": eager_ms,
                         "points_per_s": n / hip_ms * 1e3, "tflops_fp32": flop_pt * n / hip_ms * 1e-9,
                         "frac_of_fp32_mfma_peak": flop_pt * n / hip_ms * 1e-9 / 157.3,
                         "max_abs_diff_occ": (occ - ro).abs().max().item(), "max_abs_diff_rgb": (col - rc).abs().max().item()})
print(json.dumps(res))
