"""Headless, train.py-shaped driver of the object fields on the HIP path.

Keeps the reference's object-list semantics:

* objects arrive one by one, each with its own ``Trainer`` (``fc_occ_map`` + ``pe``)          train.py:123-164
* when the list changed, everything is re-stacked (``utils.update_vmap``) and the optimiser state of the new
  stack starts from zero - Adam moments restart, exactly like the new param group of utils.py:33     train.py:179-183
* per frame, ``n_iter_per_frame`` optimisation steps run over slices of the per-frame sample tensors   train.py:270-326
* the per-object modules always hold the trained weights (the write-back of train.py:331-338)

but replaces the ATen op stream by ``VmapStep.train_steps`` (one C call per frame).  The stacked parameters are views
of ONE ``[n, P]`` slab and every module parameter is re-pointed at its row of that slab, so the write-back is free:
``trainer.fc_occ_map`` / ``trainer.pe`` read the live weights (mesh queries, checkpoints) without any copy.
Keyframe management, sampling and visualisation stay with the caller (out of scope).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import layout, step


class HipMapper:
    def __init__(self, cfg, device=None, group_reduce=None, tuning=None):
        """``tuning``: passed to the OBJECT stack's ``step.VmapStep`` (e.g. ``{"kernel": _lib.KERNEL_S32_BWD6}`` for the six-product,
        float32-equivalent backward of the hidden-32 kernel, or ``KERNEL_H32_F32`` for the exact-fp32 matrix instruction); the
        background model's operator always takes the automatic plan."""
        self.cfg = cfg
        self.tuning = tuning
        self.device = torch.device(device or cfg.training_device)
        self.trainers: List = []
        self._dirty = False
        self.slab: Optional[torch.Tensor] = None
        self.views: List[torch.Tensor] = []
        self.scale: Optional[torch.Tensor] = None
        self.opt: Optional[step.FusedAdamWState] = None
        self.op: Optional[step.VmapStep] = None
        self.flag_reduce = group_reduce          # e.g. parallel.ObjectShard(...).reduce_flags for multi-GPU
        self.frames_trained = 0
        self.bg = None                           # the background field's own one-object stack (attach_background)
        self._bg_stream = None
        self._bound, self._seen = {}, {}         # frame buffers -> step.BoundFrame (see _frame_call)

    # ---- object list ---------------------------------------------------------------------------------------
    def add_object(self, trainer) -> int:
        """Append one object (its ``Trainer``); the stack is rebuilt lazily before the next frame (train.py:163-164)."""
        if self.trainers and trainer.hidden_feature_size != self.trainers[0].hidden_feature_size:
            raise ValueError("all stacked objects share one hidden width (the background model is a separate stack)")
        self.trainers.append(trainer)
        self._dirty = True
        return len(self.trainers) - 1

    def restack(self, rays: int, samples: int):
        """utils.update_vmap for the whole list: new slab, module parameters re-pointed at it, fresh optimiser state."""
        n = len(self.trainers)
        if n == 0:
            raise ValueError("no objects")
        H = self.trainers[0].hidden_feature_size
        P = layout.param_count(H)
        slab = torch.empty(n, P, dtype=torch.float32, device=self.device)
        offs = layout.flat_offsets(H)
        shapes = list(layout.fc_shapes(H)) + [layout.PE_B_SHAPE]
        views = []
        with torch.no_grad():
            for t, shp in enumerate(shapes):
                views.append(slab[:, offs[t]:offs[t] + layout.numel(shp)].view((n,) + tuple(shp)))
            for k, tr in enumerate(self.trainers):
                src = list(tr.fc_occ_map.parameters()) + [tr.pe.B_layer.weight]
                for t, p in enumerate(src):
                    views[t][k].copy_(p.detach())
                    p.data = views[t][k]                      # the module now IS a view of the slab (free write-back)
            self.scale = torch.stack([tr.pe.scale.detach().to(self.device).reshape(()) for tr in self.trainers]).contiguous()
        self.slab, self.views = slab, views
        self.opt = step.FusedAdamWState(n, H, self.device, lr=self.cfg.learning_rate, weight_decay=self.cfg.weight_decay)
        self.op = step.VmapStep(n, rays, samples, H, device=self.device, max_steps=self.cfg.n_iter_per_frame, tuning=self.tuning)
        self._dirty = False
        self._bound.pop("obj", None)
        self._seen.pop("obj", None)

    # ---- one frame -----------------------------------------------------------------------------------------
    def train_frame(self, pcs, z, gt_depth, gt_rgb, sem, depth_mask, render: bool = False) -> step.StepResult:
        """The step loop of one frame: inputs are the stacked per-frame tensors of train.py:255-260
        ([n, iters*R, S, 3], [n, iters*R, S], [n, iters*R], [n, iters*R, 3], u8 [n, iters*R], bool/u8 [n, iters*R]); ``pcs`` may be a
        ``step.RayPoints`` (the sampler's hand-off as rays, ``FrameSampler(rays=True)``)."""
        iters = self.cfg.n_iter_per_frame
        rays = pcs.shape[1] // iters
        samples = z.shape[2]                             # (pcs may be a step.RayPoints: the frame handed over as rays)
        if self._dirty or self.op is None:
            self.restack(rays, samples)                  # the object list changed: new stack, Adam restart (utils.py:33)
        elif (self.op.rays, self.op.samples) != (rays, samples):
            # only the batch shape changed: a new operator (workspace sized for the shape); slab, views and the optimiser state
            # stay - the reference restarts the moments only when update_vmap re-stacks the object list
            self.op = step.VmapStep(len(self.trainers), rays, samples, self.trainers[0].hidden_feature_size, device=self.device,
                                    max_steps=self.cfg.n_iter_per_frame, tuning=self.tuning)
            self._bound.pop("obj", None)
            self._seen.pop("obj", None)
        res = self._frame_call("obj", self.op, self.views, self.scale, (pcs, z, gt_depth, gt_rgb, sem, depth_mask), self.opt, iters, rays,
                               render, self.flag_reduce)
        self.frames_trained += 1
        return res

    def _frame_call(self, key, op, views, scale, batch, opt, iters, ray_step, render=False, flag_reduce=None):
        """``op.train_steps`` for a frame - through a ``step.BoundFrame`` when the caller hands over the SAME frame buffers as
        last time (a sampler that writes into fixed tensors, ``vmap_amd.sampler.FrameSampler``): arguments marshalled once, the
        kernels launched one by one on the device's current stream at every call (hipGraph replay is an opt-in of
        ``VmapStep.bind(graph=True)`` and is not used here: measured no gain).  New buffers (or a new stack / operator) re-bind."""
        if render:                                   # rendered outputs are per-call tensors: the plain path
            return op.train_steps(views[:14], views[14], scale, *batch, opt=opt, n_steps=iters, ray_step=ray_step, render=True,
                                  flag_reduce=flag_reduce)
        def tsig(t):
            if isinstance(t, step.RayPoints):
                return tuple(tsig(x) for x in (t.origins, t.dirs, t.centers) if x is not None)
            return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype)
        sig = (id(op), id(opt), ray_step) + tuple(tsig(t) for t in batch)
        cached = self._bound.get(key)
        if cached is None or cached[0] != sig:
            # the first frame on new buffers runs the plain path (the caller may never come back with them); the second binds
            seen = self._seen.get(key)
            self._seen[key] = sig
            if seen != sig:
                self._bound.pop(key, None)
                return op.train_steps(views[:14], views[14], scale, *batch, opt=opt, n_steps=iters, ray_step=ray_step,
                                      flag_reduce=flag_reduce)
            cached = (sig, op.bind(views[:14], views[14], scale, *batch, opt=opt, ray_step=ray_step, flag_reduce=flag_reduce))
            self._bound[key] = cached
        res = cached[1].train_steps(iters)
        # the bound outputs are reused by the next frame call: hand the caller its own copies (two tiny device copies per frame)
        return step.StepResult(res.loss[:iters].clone(), res.flags[:iters].clone())

    # ---- background field (train.py:146-152, 308-316) ----------------------------------------------------------
    def attach_background(self, trainer, rays: int, samples: int):
        """The background model (``cfg.do_bg``: one field of hidden_feature_size_bg over the whole scene, its own ray
        batch of n_per_optim_bg rays) as a second, one-object stack.  The reference adds its loss to the objects' and
        steps one optimiser (train.py:308-325); the two sets of parameters share nothing, so here each frame's object
        steps and background steps run side by side on two streams (measured: 6.7 -> 6.0 ms per frame at the Replica
        shapes, profiles/r01h_frame_bench.json)."""
        H = trainer.hidden_feature_size
        P = layout.param_count(H)
        slab = torch.empty(1, P, dtype=torch.float32, device=self.device)
        offs = layout.flat_offsets(H)
        shapes = list(layout.fc_shapes(H)) + [layout.PE_B_SHAPE]
        views = []
        with torch.no_grad():
            for t, shp in enumerate(shapes):
                views.append(slab[:, offs[t]:offs[t] + layout.numel(shp)].view((1,) + tuple(shp)))
            for t, p in enumerate(list(trainer.fc_occ_map.parameters()) + [trainer.pe.B_layer.weight]):
                views[t][0].copy_(p.detach())
                p.data = views[t][0]
        self.bg = dict(trainer=trainer, slab=slab, views=views,
                       scale=trainer.pe.scale.detach().to(self.device).reshape(1).clone(),
                       opt=step.FusedAdamWState(1, H, self.device, lr=self.cfg.learning_rate, weight_decay=self.cfg.weight_decay),
                       op=step.VmapStep(1, rays, samples, H, device=self.device, max_steps=self.cfg.n_iter_per_frame))
        # default priority: stream priorities were measured (tests/tools/frame_priority_probe.py, frame_stream_topology_probe.py,
        # profiles/round4g_*): 2.13 ms per frame with this stream at high priority in one process history, 3.9-4.2 ms in another,
        # 2.33-2.43 ms whatever the priorities in fresh processes - an artefact of which hardware queue a stream lands on, not a lever
        self._bg_stream = torch.cuda.Stream(device=self.device)

    def train_frame_with_background(self, obj_batch, bg_batch, render: bool = False):
        """One frame of both stacks: ``obj_batch`` / ``bg_batch`` = (pcs, z, gt_depth, gt_rgb, sem, depth_mask) with the
        leading dimensions [n, iters*R, ...] and [1, iters*R_bg, ...].  Returns (object StepResult, background StepResult);
        the caller's stream waits for both."""
        if self.bg is None:
            raise RuntimeError("attach_background() first")
        iters = self.cfg.n_iter_per_frame
        cur = torch.cuda.current_stream(self.device)
        fork = torch.cuda.Event()
        fork.record(cur)
        self._bg_stream.wait_event(fork)
        with torch.cuda.stream(self._bg_stream):
            b = self.bg
            for t in bg_batch:
                for x in ((t.origins, t.dirs, t.centers) if isinstance(t, step.RayPoints) else (t,)):
                    if x is not None:
                        x.record_stream(self._bg_stream)
            res_bg = self._frame_call("bg", b["op"], b["views"], b["scale"], tuple(bg_batch), b["opt"], iters, bg_batch[0].shape[1] // iters)
            join = torch.cuda.Event()
            join.record(self._bg_stream)
        res = self.train_frame(*obj_batch, render=render)
        cur.wait_event(join)
        # the background outputs were allocated from the background stream's pool: tell the allocator the caller's stream
        # reads them, so the blocks are not handed to the next frame's background work while they are still in use
        res_bg.loss.record_stream(cur)
        res_bg.flags.record_stream(cur)
        return res, res_bg

    def check_flags(self, res: step.StepResult):
        """Host-side look at the device flags of a frame (the reference exits on 'loss explode', render_rays.py:88-90)."""
        f = res.flags.cpu()
        if bool(f[:, 3].any()):
            raise RuntimeError("loss explode (a per-object loss term exceeded 1e5)")
        return f
