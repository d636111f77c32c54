"""End-to-end GPU test of the pieces in the order train.py uses them (train.py:104-338): frames -> shared keyframe
store + per-object keyframe policy -> batched sampler -> re-stack -> 20-step training frames on the fused kernel ->
field query.  A synthetic scene with known geometry (two spheres in front of a wall, analytic depth): the loss must fall,
the rendered depth of training rays must approach the ground truth and the trained field must separate free space from
the inside of its object."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

W, H = 96, 72
FX = FY = 80.0
CX, CY = (W - 1) / 2.0, (H - 1) / 2.0
SPHERES = {1: (np.array([-0.45, 0.0, 2.2], np.float32), 0.45, (220, 40, 40)),
           2: (np.array([0.55, 0.1, 2.6], np.float32), 0.40, (40, 60, 230))}
WALL_Z = 4.0


def render_frame(t_wc):
    """Analytic RGB-D + instance image of the scene from camera pose t_wc (camera-to-world); images are [W, H] like the
    reference's (vmap.py transposes to width-major)."""
    iw, ih = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="ij")
    d_c = np.stack([(iw - CX) / FX, (ih - CY) / FY, np.ones_like(iw)], -1)          # z-depth parametrisation (vmap.py:507-516)
    R, o = t_wc[:3, :3], t_wc[:3, 3]
    d_w = d_c @ R.T
    depth = np.full((W, H), np.inf, np.float32)
    inst = np.zeros((W, H), np.int32)
    rgb = np.zeros((W, H, 3), np.uint8)
    rgb[:] = (120, 120, 120)
    tw = (WALL_Z - o[2]) / d_w[..., 2]                                           # wall: plane z = WALL_Z in the world
    depth = np.where(tw > 0, tw, depth).astype(np.float32)
    for oid, (c, r, col) in SPHERES.items():
        oc = o - c
        a = (d_w * d_w).sum(-1)
        b = 2.0 * (d_w * oc).sum(-1)
        cc = (oc * oc).sum() - r * r
        disc = b * b - 4 * a * cc
        t = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
        hit = (t > 0) & (t < depth)
        depth = np.where(hit, t, depth).astype(np.float32)
        inst = np.where(hit, oid, inst)
        rgb[hit] = col
    return rgb, depth.astype(np.float32), inst


def bbox_of(inst, oid):
    ws, hs = np.nonzero(inst == oid)
    return np.array([ws.min(), ws.max(), hs.min(), hs.max()], np.float32)


def test_mapping_pipeline_learns_a_synthetic_scene():
    from vmap_amd import sampler
    from vmap_amd.driver import HipMapper
    from vmap_amd.keyframes import FrameStore, ObjectKeyframes
    from vmap_amd.trainer import SimpleConfig, Trainer
    dev = "cuda:0"
    torch.manual_seed(0)
    cfg = SimpleConfig(training_device=dev, hidden_feature_size=32, n_iter_per_frame=20, n_per_optim=120, win_size=5)
    ITERS, F, P, n1, n2 = cfg.n_iter_per_frame, 100, 24, 1, 9                     # 100 frames x 24 pixels = 20 x 120 rays
    store = FrameStore(12, W, H, device=dev)
    oks = {}
    mapper = HipMapper(cfg, device=dev)
    trainers = {}
    smp = sampler.FrameSampler(W, H, F, P, n1, n2, FX, FY, CX, CY, min_depth=0.0, surface_eps=0.1, stop_eps=0.05, device=dev, seed=3)
    losses = []
    last = None
    for fid in range(16):                                                         # 16 camera poses on a small arc
        ang = 0.06 * (fid - 7.5)
        t_wc = np.eye(4, dtype=np.float32)
        t_wc[:3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
        t_wc[:3, 3] = [0.6 * np.sin(ang) * 2.4, 0.0, 2.4 - 2.4 * np.cos(ang)]
        rgb, depth, inst = render_frame(t_wc)
        slot = store.put(torch.from_numpy(rgb), torch.from_numpy(depth), torch.from_numpy(inst), torch.from_numpy(t_wc), fid)
        for oid, (c, r, _) in SPHERES.items():
            if not (inst == oid).any():
                continue
            if oid not in oks:                                                    # new object: train.py:146-164
                oks[oid] = ObjectKeyframes(store, oid, slot, bbox_of(inst, oid), keyframe_buffer_size=6, center=tuple(float(v) for v in c))
                trainers[oid] = Trainer(SimpleConfig(training_device=dev, hidden_feature_size=32, obj_scale=1.0))
                mapper.add_object(trainers[oid])
            else:
                oks[oid].append(slot, bbox_of(inst, oid))
        store.collect()
        smp.set_objects([oks[o].sampler_entry() for o in sorted(oks)])
        fr = smp.sample()
        res = mapper.train_frame(fr["pcs"], fr["z"], fr["gt_depth"], fr["gt_rgb"], fr["sem"], fr["depth_mask"], render=True)
        mapper.check_flags(res)
        losses.append(res.loss.cpu().numpy().copy())
        last = (fr, res)
    first, final = float(losses[0][0]), float(losses[-1][-1])
    print("loss per frame (first, last step):", [(round(float(l[0]), 3), round(float(l[-1]), 3)) for l in losses])
    assert np.isfinite(final) and final < 0.6 * first, (first, final)
    # rendered depth of the last step's rays on the object: close to the sampled ground truth
    fr, res = last
    R = fr["gt_depth"].shape[1] // ITERS
    sl = slice((ITERS - 1) * R, ITERS * R)
    on_obj = (fr["sem"][:, sl] == 1) & (fr["depth_mask"][:, sl] != 0)
    err = (res.render_depth - fr["gt_depth"][:, sl]).abs()[on_obj]
    print("median |rendered - gt| depth on object rays:", float(err.median()), "of", err.numel())
    assert err.numel() > 50 and float(err.median()) < 0.25, float(err.median())
    # the modules are live views of the trained slab: the field query sees the trained weights.  Object frame = world
    # minus the object's centre.  The samples concentrate on the surface shell (vmap.py:425-445), so that is what the field
    # knows: just behind the camera-facing surface it is occupied, the space in front of it is free
    for oid, (c, r, _) in SPHERES.items():
        tr = trainers[oid]
        shell = torch.tensor([[0.0, 0.0, -r + 0.04], [0.05, 0.0, -(r * r - 0.0025) ** 0.5 + 0.04]], device=dev)
        front = torch.tensor([[0.0, 0.0, -r - 0.5], [0.1, 0.0, -r - 0.7]], device=dev)
        occ_sh, _ = tr.eval_points(shell)
        occ_fr, _ = tr.eval_points(front)
        print("object", oid, "occupancy behind the surface", occ_sh.tolist(), "in front", occ_fr.tolist())
        assert float(occ_fr.max()) < 0.2 and float(occ_sh.min()) > 0.5, (oid, occ_sh.tolist(), occ_fr.tolist())
    # the tables ran into their limit (16 frames, 6 entries: the stand-in policy overwrote old entries) and the shared store holds
    # each kept frame once
    assert any(ok.n_keyframes == ok.keyframe_buffer_size for ok in oks.values())
    live = sum(r > 0 for r in store.refs)
    assert live <= store.capacity and live < 16, live          # 16 frames came in; only the kept ones are stored, once
