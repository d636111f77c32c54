"""Multi-GPU layer: objects sharded over ranks, one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on MI355X, "gloo" in the CPU tests).

Objects are independent units - parameters, Adam state, ray samples and loss terms of one object touch no other
object - so the object list is partitioned and NOTHING is exchanged on the per-step data path.
The two cross-object couplings of the reference are handled outside it:

* the batch-wide "any object has an empty mask" switches (render_rays.py:68-73): max-reduced ONCE PER FRAME for
  all of its steps (``ObjectShard.reduce_flags``, a 4*n_steps int32 message) between ``vmapstep_prepare`` and
  ``vmapstep_train_steps_prepared``, which makes the N-GPU result identical to the 1-GPU result;
* the shared background/scene model (train.py:308-316): every rank holds a replica and trains it on its 1/N share
  of the background rays; gradients are summed with ONE all-reduce of a single flat buffer per step
  (``SharedBackground``; 94 403 fp32 = 378 KB at H=128: latency-bound on 153 GB/s xGMI links, hence one fused
  message instead of 15).  The mask counts that normalise its loss are reduced in the same way first.
  The background field itself still runs as PyTorch ops (hidden 128 does not fit the LDS-resident fused kernel;
  SURVEY.md 8(b) allows this until the K-tiled path exists).
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


def masked_losses(alpha, color, gt_depth, gt_rgb, sem, depth_mask, z, counts=None, color_scaling=5.0, opacity_scaling=10.0):
    """loss.py:5-62 for ONE field (the un-vmapped background call of train.py:311-315) with externally supplied mask
    counts, so that a ray-sharded evaluation normalises by the GLOBAL counts. alpha [R,S], color [R,S,3].
    Returns (loss, counts[3])."""
    m_o, m_s = sem != 0, sem != 2
    m_dd = depth_mask.bool() & m_o
    local = torch.stack([m_dd.sum(), m_o.sum(), m_s.sum()]).to(alpha.dtype)
    if counts is None:
        counts = local
    occ = torch.sigmoid(alpha)
    free = (1.0 - occ + 1e-10)[..., :-1]
    T = torch.cumprod(torch.cat((torch.ones_like(occ[..., :1]), free), -1), -1)
    w = occ * T
    D = (w * z).sum(-1)
    V = (w * (z - D.unsqueeze(-1)) ** 2).sum(-1).detach()
    C = (w.unsqueeze(-1) * color).sum(-2)
    O = w.sum(-1)
    zero = alpha.new_zeros(())
    l_d = zero if counts[0] == 0 else ((D - gt_depth).abs() * m_dd / (torch.sqrt(V) + 1e-4)).sum() / (counts[0] + 1e-10)
    l_c = zero if counts[1] == 0 else ((C - gt_rgb).abs().sum(-1) * m_o).sum() / (counts[1] + 1e-10)
    l_o = zero if counts[2] == 0 else ((O - m_o.to(alpha.dtype)).abs() * m_s).sum() / (counts[2] + 1e-10)
    return l_d + l_c * color_scaling + l_o * opacity_scaling, local


class SharedBackground:
    """Data-parallel training of the single shared background field (train.py:308-316) over ray shards."""

    def __init__(self, fc_occ_map: torch.nn.Module, pe: torch.nn.Module, lr=1e-3, weight_decay=0.013, group=None):
        self.fc, self.pe, self.group = fc_occ_map, pe, group
        self.params = list(fc_occ_map.parameters()) + list(pe.parameters())
        self.opt = torch.optim.AdamW(self.params, lr=lr, weight_decay=weight_decay)
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        n = sum(p.numel() for p in self.params)
        self._flat = torch.zeros(n + 1, dtype=self.params[0].dtype, device=self.params[0].device)   # grads + loss

    def ray_slice(self, n_rays: int) -> slice:
        return slice(self.rank, n_rays, self.world_size)

    def step(self, pcs, z, gt_depth, gt_rgb, sem, depth_mask) -> torch.Tensor:
        """One optimisation step on THIS rank's rays; returns the global loss. Inputs are the local ray shard."""
        _, local = masked_losses(*self._forward(pcs), gt_depth, gt_rgb, sem, depth_mask, z)    # counts only
        counts = local.clone()
        if self.world_size > 1:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)
        loss, _ = masked_losses(*self._forward(pcs), gt_depth, gt_rgb, sem, depth_mask, z, counts=counts)
        self.opt.zero_grad(set_to_none=True)
        if loss.requires_grad:
            loss.backward()
        flat = self._flat
        o = 0
        for p in self.params:
            n = p.numel()
            flat[o:o + n] = p.grad.reshape(-1) if p.grad is not None else 0.0
            o += n
        flat[o] = loss.detach()
        if self.world_size > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)      # ONE message: all gradients + loss
        o = 0
        for p in self.params:
            n = p.numel()
            p.grad = flat[o:o + n].view_as(p).clone()
            o += n
        self.opt.step()
        return flat[o].clone()

    def _forward(self, pcs):
        alpha, color = self.fc(self.pe(pcs))
        return alpha.squeeze(-1), color


class SharedBackgroundHip:
    """The shared background model (train.py:308-316) trained data-parallel over RAY shards with the HIP step.

    Every rank holds a replica whose 15 tensors are views of ONE ``[1, P]`` slab (so gradients are one contiguous
    buffer); per step:  vmapstep_prepare (local mask counts) -> all_reduce(SUM) of the 3 counts -> local forward /
    loss / backward on this rank's rays with the GLOBAL normalisers (step_main_gen for hidden 128) -> ONE
    all_reduce(SUM) of [gradient slab | loss] (378 KB at hidden 128) -> identical AdamW on every rank.
    """

    def __init__(self, fc_occ_map: torch.nn.Module, pe: torch.nn.Module, rays_local: int, samples: int, device,
                 lr=1e-3, weight_decay=0.013, group=None):
        from . import layout, step
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.modules = (fc_occ_map, pe)
        H = fc_occ_map.hidden_size
        P = layout.param_count(H)
        dev = torch.device(device)
        self.slab = torch.zeros(1, P, dtype=torch.float32, device=dev)
        self.buf = torch.zeros(P + 1, dtype=torch.float32, device=dev)             # [gradient slab | loss]
        self.gslab = self.buf[:P].view(1, P)
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
        self.slab.requires_grad_()
        self.opt = torch.optim.AdamW([self.slab], lr=lr, weight_decay=weight_decay)   # elementwise: == per-tensor AdamW
        self.op = step.VmapStep(1, rays_local, samples, H, device=dev)

    def ray_slice(self, n_rays: int) -> slice:
        return slice(self.rank, n_rays, self.world_size)

    def _reduce_counts(self, counts: torch.Tensor, flags: torch.Tensor):
        if self.world_size > 1:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)
        flags[:3] = (counts[:, :3] == 0).any(dim=0).to(torch.int32)

    def step(self, pcs, z, gt_depth, gt_rgb, sem, depth_mask) -> torch.Tensor:
        """One optimisation step; inputs are THIS rank's ray shard ([R_local, ...]); returns the global loss."""
        u = lambda x: x.unsqueeze(0)
        with torch.no_grad():
            views = [v.detach() for v in self.views]
            res = self.op.fwd_bwd(views[:14], views[14], self.scale, u(pcs), u(z), u(gt_depth), u(gt_rgb), u(sem),
                                  u(depth_mask), grads_fc=self.gviews[:14], grad_B=self.gviews[14],
                                  count_reduce=self._reduce_counts)
            self.buf[-1] = res.loss[0]
            if self.world_size > 1:
                dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)   # ONE message: gradients + loss
        self.slab.grad = self.gslab
        self.opt.step()
        return self.buf[-1].clone()

    @torch.no_grad()
    def write_back(self):
        """Copy the trained slab into the modules (what train.py:331-338 does for the object fields)."""
        fc, pe = self.modules
        for p, v in zip(list(fc.parameters()) + [pe.B_layer.weight], self.views):
            p.copy_(v[0].to(p.device))
