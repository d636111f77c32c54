"""TEST INFRASTRUCTURE: a PyTorch-ops twin of vmap_amd.parallel.SharedBackgroundHip for the CPU (gloo) tier - the same
protocol (mask counts of all steps of a frame summed once per frame; per step ONE all-reduce of [gradients | loss]) with
the per-shard arithmetic done by autograd, so that what is tested without a GPU is WHAT gets exchanged and when.  Never
imported by the product package."""
from __future__ import annotations

import torch
import torch.distributed as dist


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

    def prepare_frame(self, sem, depth_mask, n_steps: int):
        """Per FRAME: the mask counts of all steps ([n_steps * R_local] local rays), one all_reduce(SUM) of [n_steps, 3]."""
        R = sem.shape[0] // n_steps
        m_o = (sem != 0).view(n_steps, R)
        m_s = (sem != 2).view(n_steps, R)
        m_dd = depth_mask.bool().view(n_steps, R) & m_o
        counts = torch.stack([m_dd.sum(1), m_o.sum(1), m_s.sum(1)], dim=1).to(torch.float32)
        if self.world_size > 1:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)
        self.frame_counts = counts

    def step(self, pcs, z, gt_depth, gt_rgb, sem, depth_mask, step_index: int = 0) -> torch.Tensor:
        """One optimisation step on THIS rank's rays of step `step_index` of the prepared frame; returns the global loss."""
        counts = self.frame_counts[step_index].to(pcs.dtype)
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
