"""Generate the sampler fixtures by running the REAL reference sampler (vmap.py:319-459) on CPU.

Authoring container only (needs /root/reference).  ``vmap.py`` imports open3d / trimesh / bidict / cv2 / imgviz /
skimage at module level (none installed, none used by the sampler): they are stubbed in ``sys.modules``.  The reference
draws from torch's global RNG in a data-dependent order; here ``torch.randint`` / ``torch.rand`` / ``Tensor.normal_`` are
replaced, for the duration of the call, by functions that REPLAY pre-drawn per-ray numbers compacted exactly the way
the reference indexes them (the masks come from ``oracle/sampler_oracle.py``; a wrong mask would change shapes or
values and the comparison below would fail).  Stored: scene seeds + the per-ray random arrays + the reference outputs.
"""
from __future__ import annotations

import os
import sys
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True

import sampler_cases  # noqa: E402
from oracle import sampler_oracle as so  # noqa: E402

REF = "/root/reference"


def import_reference_vmap():
    for name in ("open3d", "trimesh", "cv2", "imgviz", "skimage", "skimage.measure", "scipy.spatial"):
        sys.modules.setdefault(name, mock.MagicMock())

    class bidict(dict):                       # the two features vmap.py uses: dict + .inv[value] = key
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.inv = {}
    sys.modules["bidict"] = types.SimpleNamespace(bidict=bidict)
    sys.path.insert(0, REF)
    import vmap as ref_vmap
    assert os.path.dirname(ref_vmap.__file__) == REF
    return ref_vmap


def main():
    ref_vmap = import_reference_vmap()
    from cfg import Config
    for name in sampler_cases.CASES:
        sc = sampler_cases.build_scene(name)
        cfg = Config(os.path.join(REF, "configs/Replica/config_replica_room0_vMAP.json"))
        cfg.data_device = cfg.training_device = "cpu"
        cfg.n_bins_cam2surface, cfg.n_bins = sc["n1"], sc["n2"]
        cfg.W, cfg.H = sc["W"], sc["H"]
        cfg.fx, cfg.fy, cfg.cx, cfg.cy = sc["intr"]
        cfg.min_depth = sc["min_bound"]
        cfg.do_bg = False
        K = sc["K"]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        obj = ref_vmap.sceneObject(cfg, 1, t(sc["rgbs"][0, :, :, :3]), t(sc["depth"][0]), t(sc["rgbs"][0, :, :, 3]),
                                   t(sc["bbox"][0]), t(sc["t_wc"][0]), 0)
        obj.rgbs_batch[:K] = t(sc["rgbs"])               # fill the keyframe ring directly (append_keyframe bookkeeping
        obj.depth_batch[:K] = t(sc["depth"])             # is out of scope; the sampler only reads these fields)
        obj.t_wc_batch[:K] = t(sc["t_wc"])
        obj.bbox[:K] = t(sc["bbox"])
        obj.n_keyframes = K
        obj.lastest_kf_queue = list(sc["last2"])
        obj.obj_center = t(sc["center"])
        cam = ref_vmap.cameraInfo(cfg)

        rnd = sampler_cases.draw_randoms(sc)
        o = so.sample_object(sc["rgbs"], sc["depth"], sc["t_wc"], sc["bbox"], rnd["kf_ids"], rnd["u_w"], rnd["u_h"],
                             rnd["u_z"], rnd["g_z"], sc["intr"], sc["center"], sc["n1"], sc["n2"],
                             min_bound=sc["min_bound"], eps=cfg.surface_eps, stop_eps=cfg.stop_eps)
        F, P, n1, n2 = sc["F"], sc["P"], sc["n1"], sc["n2"]
        invalid = ~o["valid"]
        objm = (o["labels"] == 1) & o["valid"]
        other = (o["labels"] != 1) & o["valid"]
        queue = [("randint", rnd["kf_ids"][:F - 2] if K > 2 else rnd["kf_ids"]),
                 ("rand", rnd["u_w"]), ("rand", rnd["u_h"])]
        if invalid.any():
            queue.append(("rand", rnd["u_z"][invalid]))
        if o["valid"].any():
            queue.append(("rand", rnd["u_z"][o["valid"]][:, :n1]))
            if objm.any():
                queue.append(("normal", rnd["g_z"][objm]))
            if other.any():
                queue.append(("rand", rnd["u_z"][other][:, n1:]))
        calls = []

        def take(kind, shape):
            k, arr = queue.pop(0)
            assert k == kind, (k, kind, shape)
            assert tuple(arr.shape) == tuple(shape), (kind, arr.shape, shape)
            calls.append(kind)
            return torch.from_numpy(np.ascontiguousarray(arr))

        def fake_randint(low=0, high=None, size=None, **kw):
            return take("randint", size).to(torch.long)

        def fake_rand(*size, **kw):
            return take("rand", size).to(torch.float32)

        def fake_normal_(self, mean=0.0, std=1.0):
            g = take("normal", self.shape).to(torch.float32)
            self.copy_(g * std + mean)
            return self

        with mock.patch.object(torch, "randint", fake_randint), mock.patch.object(torch, "rand", fake_rand), \
                mock.patch.object(torch.Tensor, "normal_", fake_normal_):
            rgb, depth, valid, labels, pcs, z = obj.get_training_samples(F, P, cam.rays_dir_cache)
        assert not queue, queue
        ref = dict(rgb=rgb.numpy().reshape(F * P, 3), depth=depth.numpy().reshape(-1), valid=valid.numpy(),
                   labels=labels.numpy(), pcs=pcs.numpy().reshape(F * P, n1 + n2, 3), z=z.numpy().reshape(F * P, n1 + n2))
        # the oracle must already agree (this is the pin); store the reference's outputs
        for k in ("rgb", "labels", "valid"):
            assert np.array_equal(ref[k], o[k]), k
        for k in ("depth", "z", "pcs"):
            err = np.abs(ref[k].astype(np.float64) - o[k]).max()
            assert err < 2e-6, (k, err)
        out = {("ref_" + k): v for k, v in ref.items()}
        out["scene_sha256"] = np.array(sampler_cases.digest(sc, rnd))
        path = os.path.join(HERE, f"sampler_{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name:10s} F={F} P={P} S={n1 + n2} invalid={int(invalid.sum())} obj={int(objm.sum())} other={int(other.sum())} "
              f"calls={calls} -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
