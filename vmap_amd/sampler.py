"""Host side of the batched HIP sampler: keyframe buffers of all objects in, the six per-frame tensors out.

Mirrors the call the reference makes per object and frame - ``obj_k.get_training_samples(n_iter_per_frame * win_size,
n_samples_per_frame, rays_dir_cache)`` (train.py:208-218, vmap.py:319-364) followed by the reshapes and ``torch.stack``
of train.py:213-218,255-260 - but for every object in ONE launch (``vmapstep_sample_frame``).  Keyframe management
(which frames are kept, pruning) stays with the caller; this class only needs, per object, the tensors the reference's
``sceneObject`` already owns: ``rgbs_batch``, ``depth_batch``, ``t_wc_batch``, ``bbox``, ``n_keyframes``,
``lastest_kf_queue[-2:]`` and ``obj_center``.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib


class FrameSampler:
    def __init__(self, width, height, frames, samples_per_frame, n_bins_cam2surface, n_bins, fx, fy, cx, cy,
                 min_depth=0.0, surface_eps=0.1, stop_eps=0.05, device="cuda:0", seed=0, reuse_outputs=False, split=True, rays=False):
        """``rays``: hand the frame over as RAYS (``vmapstep_sample_frame_rays``): ``sample()["pcs"]`` is then a ``step.RayPoints``
        (origins / dirs [n, F*P, 3] + the objects' centres) instead of the [n, F*P, S, 3] points tensor - 64 instead of 160 bytes per
        ray at S = 10; the step kernels rebuild the points bit-identically (second half; vmap.py:452-457).
        ``reuse_outputs``: ``sample()`` writes into the SAME six tensors every frame (a consumer that binds its frame buffers -
        ``step.BoundFrame``, ``driver.HipMapper`` - then marshals them once); ``split``: many workgroups per object (a small
        workspace holds the objects' maximum depths between the two launches) instead of one."""
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.cfg = _lib.SampleCfg(width, height, frames, samples_per_frame, n_bins_cam2surface, n_bins,
                                  fx, fy, cx, cy, min_depth, surface_eps, stop_eps)
        self.F, self.P, self.S = frames, samples_per_frame, n_bins_cam2surface + n_bins
        self.seed = int(seed)
        self.frame_counter = 0
        self._table = None
        self._keep = None
        self.n_obj = 0
        self.reuse_outputs, self.split, self.rays = bool(reuse_outputs), bool(split), bool(rays)
        self._out, self._ws = None, None

    def set_objects(self, objects: Sequence[dict]):
        """objects: per object EITHER the reference's own buffers - dict(rgbs u8 [K,W,H,4], depth f32 [K,W,H], t_wc f32
        [K,4,4], bbox f32 [K,4], n_keyframes int, last2 (int, int), center (3 floats)) - OR an entry over a shared frame
        store (``keyframes.ObjectKeyframes.sampler_entry()``: dict(store, slots i32 [K], bbox, n_keyframes, last2,
        center, obj_id)).  Tensors on this sampler's device, contiguous."""
        n = len(objects)
        host = (_lib.SampleObject * n)()
        keep = []

        def chk(i, name, t, dt):
            if t.dtype != dt or t.device != self.device or not t.is_contiguous():
                raise ValueError(f"object {i}: {name} must be a contiguous {dt} tensor on {self.device}")

        for i, o in enumerate(objects):
            c = [float(v) for v in (o["center"].reshape(-1).tolist() if torch.is_tensor(o["center"]) else o["center"])]
            if len(c) == 1:
                c = c * 3                                   # the reference's default obj_center is the scalar 0.0
            last2 = (ctypes.c_int32 * 2)(*[int(v) for v in o["last2"]])
            chk(i, "bbox", o["bbox"], torch.float32)
            if "store" in o:
                st = o["store"]
                if st.W != self.cfg.width or st.H != self.cfg.height:
                    raise ValueError(f"object {i}: store is {st.W}x{st.H}, sampler is {self.cfg.width}x{self.cfg.height}")
                chk(i, "slots", o["slots"], torch.int32)
                for name, t, dt in (("store.rgbx", st.rgbx, torch.uint8), ("store.depth", st.depth, torch.float32),
                                    ("store.t_wc", st.t_wc, torch.float32), ("store.inst", st.inst, torch.int32)):
                    chk(i, name, t, dt)
                keep.append((st.rgbx, st.depth, st.t_wc, st.inst, o["slots"], o["bbox"]))
                host[i] = _lib.SampleObject(st.rgbx.data_ptr(), st.depth.data_ptr(), st.t_wc.data_ptr(), o["bbox"].data_ptr(),
                                            int(o["n_keyframes"]), last2, (ctypes.c_float * 3)(*c), int(o["obj_id"]),
                                            o["slots"].data_ptr(), st.inst.data_ptr())
            else:
                for k, dt in (("rgbs", torch.uint8), ("depth", torch.float32), ("t_wc", torch.float32)):
                    chk(i, k, o[k], dt)
                keep.append((o["rgbs"], o["depth"], o["t_wc"], o["bbox"]))
                host[i] = _lib.SampleObject(o["rgbs"].data_ptr(), o["depth"].data_ptr(), o["t_wc"].data_ptr(), o["bbox"].data_ptr(),
                                            int(o["n_keyframes"]), last2, (ctypes.c_float * 3)(*c), 0, None, None)
        raw = np.frombuffer(bytes(host), dtype=np.uint8).copy()
        self._table = torch.from_numpy(raw).to(self.device)
        self._keep = keep
        self.n_obj = n
        self._out, self._ws = None, None
        if self.split:
            nb = ctypes.c_size_t(0)
            _lib.check(self.lib.vmapstep_sample_workspace_bytes(n, ctypes.byref(nb)), self.lib)
            self._ws = torch.empty(nb.value, dtype=torch.uint8, device=self.device)

    def sample(self, test_randoms: Optional[dict] = None):
        """One frame of samples for all objects -> dict(pcs [n,F*P,S,3], z [n,F*P,S], gt_depth [n,F*P], gt_rgb [n,F*P,3],
        sem u8 [n,F*P], depth_mask u8 [n,F*P]) on the device.  ``test_randoms``: dict of device tensors kf_ids int32
        [n,F], u_w/u_h f32 [n,F*P], u_z [n,F*P,S], g_z [n,F*P,n_bins] (deterministic test mode)."""
        if self._table is None:
            raise RuntimeError("set_objects() first")
        n, FP, S, dev = self.n_obj, self.F * self.P, self.S, self.device
        out = self._out
        if out is None:
            from .step import RayPoints
            pts = RayPoints(torch.empty(n, FP, 3, device=dev), torch.empty(n, FP, 3, device=dev), torch.empty(n, 3, device=dev)) if self.rays \
                else torch.empty(n, FP, S, 3, device=dev)
            out = dict(pcs=pts, z=torch.empty(n, FP, S, device=dev),
                       gt_depth=torch.empty(n, FP, device=dev), gt_rgb=torch.empty(n, FP, 3, device=dev),
                       sem=torch.empty(n, FP, dtype=torch.uint8, device=dev), depth_mask=torch.empty(n, FP, dtype=torch.uint8, device=dev))
            if self.reuse_outputs:
                self._out = out
        rnd = None
        if test_randoms is not None:
            rnd = _lib.SampleRandoms(*[test_randoms[k].data_ptr() if test_randoms.get(k) is not None else None
                                       for k in ("kf_ids", "u_w", "u_h", "u_z", "g_z")])
        tail = (out["z"].data_ptr(), out["gt_depth"].data_ptr(), out["gt_rgb"].data_ptr(), out["sem"].data_ptr(), out["depth_mask"].data_ptr(),
                self.seed, self.frame_counter, ctypes.byref(rnd) if rnd is not None else None,
                self._ws.data_ptr() if self._ws is not None else None, self._ws.numel() if self._ws is not None else 0,
                torch.cuda.current_stream(self.device).cuda_stream)
        if self.rays:
            r = out["pcs"]
            _lib.check(self.lib.vmapstep_sample_frame_rays(ctypes.byref(self.cfg), self._table.data_ptr(), n, r.origins.data_ptr(),
                                                           r.dirs.data_ptr(), r.centers.data_ptr(), None, *tail), self.lib)
        else:
            _lib.check(self.lib.vmapstep_sample_frame(ctypes.byref(self.cfg), self._table.data_ptr(), n, out["pcs"].data_ptr(), *tail), self.lib)
        self.frame_counter += 1
        return out
