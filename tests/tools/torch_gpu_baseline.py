"""Measurement tool (not product): the eager PyTorch port of the step (oracle/vmap_oracle_torch.py) timed on the
GPU - the stand-in for 'the reference single-GPU PyTorch path' whose rays/s the north star asks to beat by >= 5x.
Usage (GPU box): python tests/tools/torch_gpu_baseline.py [config] > gpurun_out/torch_gpu_baseline.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import vmap_oracle_torch as vt  # noqa: E402
from vmap_amd import synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "replica_room0_vmap"
cfg = synth.CONFIGS[name]
fc, B, sc = synth.make_params(cfg["n_obj"], cfg["H"], scale=cfg["scale"], seed=0)
batch = synth.make_batch(cfg["n_obj"], cfg["R"], cfg["S"], seed=1)
dev = "cuda:0"
tr = vt.CpuTrainer(fc, B, sc, device=dev)
tb = {k: torch.from_numpy(v).to(dev) for k, v in batch.items()}
for _ in range(10):
    tr.step(tb)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 100
for _ in range(N):
    tr.step(tb)
torch.cuda.synchronize()
el = time.perf_counter() - t0
rays = cfg["n_obj"] * cfg["R"]
print(json.dumps({"what": "eager PyTorch-ROCm port of the step (fwd+loss+bwd+AdamW), device-resident batch",
                  "config": name, "ms_per_step": el / N * 1e3, "rays_per_s": rays * N / el,
                  "device": torch.cuda.get_device_name(0), "torch": torch.__version__}))
