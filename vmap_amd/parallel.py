"""Multi-GPU layer: objects sharded over ranks, one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on MI355X, "gloo" in the CPU tests).

Objects are independent units - parameters, Adam state, ray samples and loss terms of one object touch no other
object - so the object list is partitioned and NOTHING is exchanged on the per-step data path.
The two cross-object couplings of the reference are handled outside it:

* the batch-wide "any object has an empty mask" switches (render_rays.py:68-73): max-reduced ONCE PER FRAME for
  all of its steps (``ObjectShard.reduce_flags``, a 4*n_steps int32 message) between ``vmapstep_prepare`` and
  ``vmapstep_train_steps_prepared``, which makes the N-GPU result identical to the 1-GPU result;
* the shared background/scene model (train.py:308-316): every rank holds a replica and trains it on its 1/N share
  of the background rays with the HIP step (``SharedBackgroundHip``): the mask counts that normalise its loss are summed
  ONCE PER FRAME for all of its steps, and per step the gradients + loss travel in ONE all-reduce of a single flat buffer
  (94 403 fp32 = 378 KB at H=128: latency-bound on 153 GB/s xGMI links, hence one fused message instead of 15) between two
  launches (forward/backward, fused AdamW + image rewrite).  The alternative SURVEY.md 8(e) names is here too:
  ``OwnerBackgroundHip`` - ONE rank trains the background on all of its rays with the plain frame call and broadcasts the
  trained parameter slab once per frame (no collective on the step path); ``bench.py --gpus N`` measures both.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


class ObjectShard:
    """Round-robin partition of ``n_total`` objects over the ranks of ``group`` (owner computes)."""

    def __init__(self, n_total: int, rank: Optional[int] = None, world_size: Optional[int] = None, group=None):
        self.group = group
        self.world_size = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        self.n_total = n_total
        self.owned: List[int] = list(range(self.rank, n_total, self.world_size))

    def take(self, seq: Sequence):
        """This rank's elements of a per-object sequence (modules, sceneObjects, ...)."""
        return [seq[i] for i in self.owned]

    def reduce_flags(self, flags: torch.Tensor) -> torch.Tensor:
        """In-place MAX over ranks of the int32 [n_steps, 4] empty-mask switches (once per frame)."""
        if self.world_size > 1:
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
        return flags

    def sum_losses(self, loss: torch.Tensor) -> torch.Tensor:
        """Total batch loss over all ranks (logging only; not needed for training)."""
        if self.world_size > 1:
            dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=self.group)
        return loss


class SharedBackgroundHip:
    """The shared background model (train.py:308-316) trained data-parallel over RAY shards with the HIP step.

    Every rank holds a replica whose 15 tensors are views of ONE ``[1, P]`` slab.  Per FRAME: ``vmapstep_prepare`` (parameter
    image + this rank's mask counts of every step) and ONE ``all_reduce(SUM)`` of the ``[n_steps, 4]`` counts, from which the
    empty-mask switches of all steps are rewritten.  Per STEP: launch (forward / loss / backward on this rank's rays with
    the GLOBAL normalisers: ``vmapstep_fwd_bwd_prepared``) -> ONE ``all_reduce(SUM)`` of ``[gradient slab | loss terms]`` (378 KB
    at hidden 128: latency-bound on xGMI, hence a single message) -> launch (``vmapstep_adamw_apply``: identical fused AdamW
    on every rank + rewrite of the parameter image + global loss and "loss explode" flag from the summed terms).  Everything is enqueued on the caller's current stream; nothing
    synchronises the host.
    """

    def __init__(self, fc_occ_map: torch.nn.Module, pe: torch.nn.Module, rays_local: int, samples: int, device,
                 lr=1e-3, weight_decay=0.013, group=None, max_steps: int = 32):
        from . import layout, step
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.modules = (fc_occ_map, pe)
        H = fc_occ_map.hidden_size
        P = layout.param_count(H)
        dev = torch.device(device)
        self.opt = step.FusedAdamWState(1, H, dev, lr=lr, weight_decay=weight_decay)
        PP = self.opt.padded
        self.slab = torch.zeros(1, P, dtype=torch.float32, device=dev)
        # ONE message per step: [gradient row (padded) | depth, colour, opacity terms, l_batch] - the kernels write both parts in place
        self.buf = torch.zeros(PP + 4, dtype=torch.float32, device=dev)
        self.gslab = self.buf[:PP].view(1, PP)
        self.terms = self.buf[PP:].view(1, 4)
        self.losses = torch.zeros(max_steps, dtype=torch.float32, device=dev)      # global loss per step of the current frame
        self.flags = torch.zeros(max_steps, 4, dtype=torch.int32, device=dev)      # global flags per step (empty masks, explode)
        offs = layout.flat_offsets(H)
        shapes = list(layout.fc_shapes(H)) + [layout.PE_B_SHAPE]
        src = list(fc_occ_map.parameters()) + [pe.B_layer.weight]
        self.views, self.gviews = [], []
        with torch.no_grad():
            for t, shp in enumerate(shapes):
                n = layout.numel(shp)
                v = self.slab[:, offs[t]:offs[t] + n].view((1,) + tuple(shp))
                v.copy_(src[t].detach().to(dev).unsqueeze(0))
                self.views.append(v)
                self.gviews.append(self.gslab[:, offs[t]:offs[t] + n].view((1,) + tuple(shp)))
        self.scale = pe.scale.detach().to(dev).reshape(1).clone()
        self.rays_local, self.max_steps = rays_local, max_steps
        self.op = step.VmapStep(1, rays_local, samples, H, device=dev, max_steps=max_steps)
        self._frame = None

    def ray_slice(self, n_rays: int) -> slice:
        return slice(self.rank, n_rays, self.world_size)

    @torch.no_grad()
    def prepare_frame(self, pcs, z, gt_depth, gt_rgb, sem, depth_mask, n_steps: int):
        """Inputs: THIS rank's rays of a whole frame, ``[n_steps * R_local, ...]``.  One collective: the mask counts of all steps.
        Also marshals the frame's C arguments ONCE (parameter / gradient blocks, one batch block per step, the output blocks): at
        8 ranks a step is ~0.1 ms of device time, which per-step Python marshalling (~0.2 ms) would exceed."""
        import ctypes
        from . import _lib
        u = lambda x: x.unsqueeze(0)
        self._frame = tuple(u(x) for x in (pcs, z, gt_depth, gt_rgb, sem, depth_mask))
        op = self.op
        counts, flags = op.prepare_frame(self.views[:14], self.views[14], *self._frame, n_steps=n_steps, ray_step=self.rays_local)
        if self.world_size > 1:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)        # [n_steps, 1, 4] float32, once per frame
        flags[:, :3] = (counts[:, :, :3] == 0).any(dim=1).to(torch.int32)
        self._n_steps = n_steps
        R = self.rays_local
        sig = (n_steps,) + tuple(x.signature() if hasattr(x, "signature") else (x.data_ptr(), tuple(x.shape), tuple(x.stride())) for x in self._frame)
        if getattr(self, "_sig", None) == sig:
            return                                   # the same frame buffers as last time (a sampler with fixed outputs): blocks still valid
        self._sig = sig
        self._pp = op._params(self.views[:14], self.views[14])
        self._gp = op._params(self.gviews[:14], self.gviews[14], "grads")
        self._sc = _lib.Tensor(self.scale.data_ptr(), 0)
        self._bt = [op._batch(*(x[:, i * R:(i + 1) * R] for x in self._frame)) for i in range(n_steps)]
        self._local = (torch.empty(1, dtype=torch.float32, device=self.buf.device), torch.empty(4, dtype=torch.int32, device=self.buf.device))
        self._out_fb = _lib.Outputs(self._local[0].data_ptr(), self._local[1].data_ptr(), None, None, None, None, self.terms.data_ptr())
        self._out_ap = [_lib.Outputs(self.losses[i:i + 1].data_ptr(), self.flags[i].data_ptr(), None, None, None, None, None)
                        for i in range(n_steps)]
        self._byref = ctypes.byref

    @torch.no_grad()
    def step_prepared(self, i: int) -> torch.Tensor:
        """Step i of the prepared frame: launch, ONE all_reduce, launch.  Returns the global loss (0-dim view of ``self.losses``).

        The forward/backward launch leaves this rank's gradients AND its three loss terms in ``self.buf``; after the all-reduce
        ``vmapstep_adamw_apply`` updates the replica and, in the same launch, evaluates the step's global loss and the
        "loss explode" test (render_rays.py:88-90) on the SUMMED terms - no launch besides the two and the collective.  Host
        side: two C calls on blocks marshalled by ``prepare_frame`` and one ``all_reduce``."""
        from . import _lib
        op, opt, br = self.op, self.opt, self._byref
        st = op._stream()
        _lib.check(lib=op.lib, rc=op.lib.vmapstep_fwd_bwd_prepared(br(op.shape), br(self._pp), br(self._sc), br(self._bt[i]), i, op.color_scaling,
                                                    op.opacity_scaling, br(self._gp), br(self._out_fb), op._ws_ptr, op._ws_bytes, st))
        if self.world_size > 1:
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)      # ONE message: gradients + loss terms
        oc = opt.c_struct()
        _lib.check(lib=op.lib, rc=op.lib.vmapstep_adamw_apply(br(op.shape), br(self._pp), self.gslab.data_ptr(), opt.padded, br(oc), self.terms.data_ptr(),
                                               i, op.color_scaling, op.opacity_scaling, br(self._out_ap[i]), op._ws_ptr, op._ws_bytes, st))
        opt.step += 1
        opt.note_host_steps(1)
        return self.losses[i]

    def train_frame(self, pcs, z, gt_depth, gt_rgb, sem, depth_mask, n_steps: int) -> torch.Tensor:
        """The background part of train.py:270-326 for one frame; returns the per-step global losses [n_steps]
        (``self.flags[:n_steps]`` holds the per-step flags; ``check_flags`` turns an explode into an exception)."""
        self.prepare_frame(pcs, z, gt_depth, gt_rgb, sem, depth_mask, n_steps)
        for i in range(n_steps):
            self.step_prepared(i)
        return self.losses[:n_steps].clone()

    def check_flags(self, n_steps: Optional[int] = None):
        """Host-side check of the explode flags of the last frame (synchronises): the reference calls exit(-1) at
        render_rays.py:88-90; here the decision travels as a device flag and the caller decides when to look."""
        fl = self.flags[:self._n_steps if n_steps is None else n_steps, 3]
        if bool((fl != 0).any()):
            raise RuntimeError("background model: loss explode (render_rays.py:88-90) in step(s) "
                               f"{torch.nonzero(fl).flatten().tolist()}")

    def step(self, pcs, z, gt_depth, gt_rgb, sem, depth_mask) -> torch.Tensor:
        """One optimisation step on a one-step frame ([R_local, ...] inputs); returns the global loss."""
        self.prepare_frame(pcs, z, gt_depth, gt_rgb, sem, depth_mask, 1)
        return self.step_prepared(0).clone()

    @torch.no_grad()
    def write_back(self):
        """Copy the trained slab into the modules (what train.py:331-338 does for the object fields)."""
        fc, pe = self.modules
        for p, v in zip(list(fc.parameters()) + [pe.B_layer.weight], self.views):
            p.copy_(v[0].to(p.device))


class OwnerBackgroundHip:
    """The shared background model (train.py:308-316), OWNER-COMPUTES: rank ``owner`` trains it on ALL of a step's rays with
    the plain frame call (``vmapstep_train_steps``: 1 + 2 n launches, nothing between the steps), every other rank does nothing
    for it during the frame; ONE ``broadcast`` of the ``[1, P]`` parameter slab per FRAME (378 KB at hidden 128) hands the
    trained weights to the other ranks' replicas (for rendering / meshing: no object's training reads the background field).

    Against ``SharedBackgroundHip`` (ray shards + one all-reduce per STEP): no collective on the step path and a twentieth of
    the messages, but the owner carries the whole 1200-ray step (0.10 ms on one MI355X against ~0.06 ms of launches + the
    all-reduce at 8 ranks) next to its object shard.  The owner's result is bit-identical to single-GPU
    training of the background model (same kernels, same plan); the replicas are bit-identical copies of it."""

    def __init__(self, fc_occ_map: torch.nn.Module, pe: torch.nn.Module, rays: int, samples: int, device, owner: int = 0,
                 lr=1e-3, weight_decay=0.013, group=None, max_steps: int = 32):
        from . import layout, step
        self.group, self.owner = group, owner
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.modules = (fc_occ_map, pe)
        H = fc_occ_map.hidden_size
        dev = torch.device(device)
        self.slab = torch.zeros(1, layout.param_count(H), dtype=torch.float32, device=dev)
        offs = layout.flat_offsets(H)
        shapes = list(layout.fc_shapes(H)) + [layout.PE_B_SHAPE]
        src = list(fc_occ_map.parameters()) + [pe.B_layer.weight]
        self.views = []
        with torch.no_grad():
            for t, shp in enumerate(shapes):
                v = self.slab[:, offs[t]:offs[t] + layout.numel(shp)].view((1,) + tuple(shp))
                v.copy_(src[t].detach().to(dev).unsqueeze(0))
                self.views.append(v)
        self.scale = pe.scale.detach().to(dev).reshape(1).clone()
        self.rays, self.max_steps = rays, max_steps
        self.is_owner = self.rank == owner
        self.op = self.opt = None
        if self.is_owner:                            # only the owner needs an operator, a workspace and optimiser state
            self.op = step.VmapStep(1, rays, samples, H, device=dev, max_steps=max_steps)
            self.opt = step.FusedAdamWState(1, H, dev, lr=lr, weight_decay=weight_decay)
        self._bound, self._sig = None, None
        self.result = None

    @torch.no_grad()
    def train_frame(self, pcs, z, gt_depth, gt_rgb, sem, depth_mask, n_steps: int, broadcast: bool = True):
        """The frame's ``n_steps`` background steps on the owner (inputs: ALL rays of the frame, ``[n_steps * R, ...]``; other
        ranks may pass None), then the slab broadcast.  Returns the owner's StepResult (loss [max_steps], flags) or None."""
        if self.is_owner:
            fr = tuple(x.unsqueeze(0) for x in (pcs, z, gt_depth, gt_rgb, sem, depth_mask))
            sig = tuple(x.signature() if hasattr(x, "signature") else (x.data_ptr(), tuple(x.shape), tuple(x.stride())) for x in fr)
            if self._sig != sig:                     # new frame buffers: marshal once
                self._bound = self.op.bind(self.views[:14], self.views[14], self.scale, *fr, opt=self.opt, ray_step=self.rays)
                self._sig = sig
            self.result = self._bound.train_steps(n_steps)
        if broadcast and self.world_size > 1:
            dist.broadcast(self.slab, src=self.owner if self.group is None else dist.get_global_rank(self.group, self.owner), group=self.group)
        return self.result if self.is_owner else None

    @torch.no_grad()
    def write_back(self):
        fc, pe = self.modules
        for p, v in zip(list(fc.parameters()) + [pe.B_layer.weight], self.views):
            p.copy_(v[0].to(p.device))
