"""Measurement tool: batched HIP sampler vs an eager PyTorch-ROCm port of the reference's per-object sampler loop
(train.py:208-218 + vmap.py:319-459 + the torch.stack of train.py:255-260) at the Replica room0 vMAP shapes."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vmap_amd import sampler  # noqa: E402

dev = "cuda:0"
n, K, W, H, F, P, n1, n2 = 20, 20, 1200, 680, 100, 24, 1, 9
fx = fy = 600.0
cx, cy = 599.5, 339.5
eps, stop_eps, min_b = 0.1, 0.05, 0.0
torch.manual_seed(0)
objs = []
for k in range(n):
    rgbs = torch.randint(0, 256, (K, W, H, 4), dtype=torch.uint8, device=dev)
    rgbs[..., 3] = torch.randint(0, 3, (K, W, H), dtype=torch.uint8, device=dev)
    depth = torch.rand(K, W, H, device=dev) * 3.5 + 0.5
    depth[torch.rand(K, W, H, device=dev) < 0.05] = 0.0
    t_wc = torch.eye(4, device=dev).repeat(K, 1, 1).contiguous()
    t_wc[:, :3, 3] = torch.rand(K, 3, device=dev)
    bbox = torch.tensor([[100.0, 900.0, 50.0, 600.0]], device=dev).repeat(K, 1).contiguous()
    objs.append(dict(rgbs=rgbs, depth=depth, t_wc=t_wc, bbox=bbox, n_keyframes=K, last2=(K - 2, K - 1), center=(0.0, 0.0, 0.0)))

def time_sampler(split):
    smp = sampler.FrameSampler(W, H, F, P, n1, n2, fx, fy, cx, cy, min_depth=min_b, surface_eps=eps, stop_eps=stop_eps, device=dev,
                               split=split, reuse_outputs=True)
    smp.set_objects(objs)
    for _ in range(3):
        smp.sample()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    N = 50
    for _ in range(N):
        smp.sample()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N


one_wg_ms = time_sampler(False)      # one workgroup per object (rounds 1-2)
hip_ms = time_sampler(True)          # split form: many workgroups per object (round 3, the default)

# eager PyTorch port of the reference loop (same ops, per object)
idx_w_c = torch.arange(W, device=dev)
idx_h_c = torch.arange(H, device=dev)
dirs_cache = torch.ones(W, H, 3, device=dev)
dirs_cache[:, :, 0] = ((idx_w_c - cx) / fx)[:, None]
dirs_cache[:, :, 1] = (idx_h_c - cy) / fy


def strat(lo, hi, nb, nr):
    lim = torch.linspace(0, 1, nb + 1, device=dev)
    if not torch.is_tensor(lo):
        lo = torch.ones(nr, device=dev) * lo
    if not torch.is_tensor(hi):
        hi = torch.ones(nr, device=dev) * hi
    rng = hi - lo
    lower = (rng[..., None] * lim + lo[..., None])[:, :-1]
    return lower + torch.rand(nr, nb, device=dev) * (rng / nb)[..., None]


def ref_sample(o):
    kf = torch.cat([torch.randint(0, K, (F - 2,), device=dev), torch.tensor(o["last2"], device=dev)]).unsqueeze(-1)
    iw = torch.rand(F, P, device=dev) * (o["bbox"][kf, 1] - o["bbox"][kf, 0]) + o["bbox"][kf, 0]
    ih = torch.rand(F, P, device=dev) * (o["bbox"][kf, 3] - o["bbox"][kf, 2]) + o["bbox"][kf, 2]
    iw, ih = iw.long(), ih.long()
    srgb = o["rgbs"][kf, iw, ih]
    sdep = o["depth"][kf, iw, ih]
    sdir = dirs_cache[iw, ih]
    T = o["t_wc"][kf[:, 0]]
    dw = (T[:, None, :3, :3] @ sdir[..., None]).squeeze()
    org = T[:, :3, -1]
    z = torch.zeros(F * P, n1 + n2, device=dev)
    inv = (sdep <= min_b).view(-1)
    mx = sdep.max()
    ic = inv.count_nonzero()
    if ic:
        z[inv] = strat(min_b, mx, n1 + n2, ic)
    v = ~inv
    vc = v.count_nonzero()
    if vc:
        z[v, :n1] = strat(min_b, sdep.view(-1)[v] - eps, n1, vc)
        om = (srgb[..., -1] == 1).view(-1) & v
        oc = om.count_nonzero()
        if oc:
            bins = torch.empty(oc, n2, device=dev).normal_(0.0, eps / 3.0).sort().values.clip(-eps, eps)
            z[om, n1:] = sdep.view(-1)[om][:, None] + bins
        tm = (srgb[..., -1] != 1).view(-1) & v
        tc = tm.count_nonzero()
        if tc:
            z[tm, n1:] = strat(sdep.view(-1)[tm] - eps, sdep.view(-1)[tm] + stop_eps, n2, tc)
    zz = z.view(F, P, -1)
    pcs = org[..., None, None, :] + dw[:, :, None, :] * zz[..., None]
    return srgb[..., :3], sdep, v, srgb[..., -1].view(-1), pcs, zz


def ref_frame():
    outs = [ref_sample(o) for o in objs]
    pcs = torch.stack([x[4].reshape(F * P, n1 + n2, 3) for x in outs])
    gd = torch.stack([x[1].reshape(F * P) for x in outs])
    rgb = torch.stack([x[0].reshape(F * P, 3) for x in outs]) / 255.0
    dm = torch.stack([x[2] for x in outs])
    sm = torch.stack([x[3] for x in outs])
    zz = torch.stack([x[5].reshape(F * P, n1 + n2) for x in outs])
    return pcs, gd, rgb, dm, sm, zz


for _ in range(2):
    ref_frame()
torch.cuda.synchronize()
t0 = time.perf_counter()
M = 10
for _ in range(M):
    ref_frame()
torch.cuda.synchronize()
ref_ms = (time.perf_counter() - t0) / M * 1e3
rays = n * F * P
written = rays * ((n1 + n2) * 16 + 4 + 12 + 2)
print(json.dumps({"what": "one frame of ray samples for all objects (20 objects x 100 frames x 24 px, 10 samples/ray)",
                  "hip_ms": hip_ms, "hip_ms_one_workgroup_per_object": one_wg_ms, "hip_rays_per_s": rays / (hip_ms * 1e-3), "hip_write_GBs": written / (hip_ms * 1e-3) / 1e9,
                  "eager_pytorch_rocm_ms": ref_ms, "speedup": ref_ms / hip_ms,
                  "bytes_written_per_frame": written, "device": torch.cuda.get_device_name(0)}))
