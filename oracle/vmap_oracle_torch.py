"""TEST INFRASTRUCTURE ONLY - PyTorch-CPU restatement of the vectorised training step, used as the timed
``cpu_baseline`` ("port") in bench.py and cross-checked against the golden fixtures in tests/.

The reference's arithmetic lives in PyTorch ATen + functorch (pinned pytorch=1.12.1 / functorch==0.2.0,
environment.yml:59,85), driven from train.py:293-326.  The reference tree itself cannot travel to the GPU
box, so this file restates that step with plain batched torch ops (bmm instead of vmap) + autograd +
``torch.optim.AdamW`` - the same third-party kernels the reference ends up calling, multi-threaded.
Never imported by the product package.
"""
from __future__ import annotations

import math

import torch


def _lin(x, W, b):                       # x [n,M,K], W [n,J,K], b [n,J]
    return torch.baddbmm(b.unsqueeze(1), x, W.transpose(1, 2))


def forward_loss(fc, B, scale, pcs, z, gt_depth, gt_rgb, sem, depth_mask, color_scaling=5.0, opacity_scaling=10.0):
    """embedding.py:82-91 -> model.py:54-85 -> loss.py:5-62, batched over the object dimension."""
    n, R, S, _ = pcs.shape
    t = (pcs / scale.view(n, 1, 1, 1)).reshape(n, R * S, 3)
    proj = torch.bmm(t, B.transpose(1, 2))                                   # [n,M,21]
    bands = 2.0 ** torch.arange(6, dtype=pcs.dtype, device=pcs.device)
    xb = (proj.unsqueeze(-2) * bands.view(1, 1, 6, 1)).reshape(n, R * S, 126)
    emb = torch.cat((t, torch.sin(xb * math.pi)), dim=-1)
    e1, e2 = emb[..., :87], emb[..., 87:]
    h = torch.relu(_lin(e1, fc[0], fc[1]))
    h = torch.relu(_lin(h, fc[2], fc[3]))
    h = torch.relu(_lin(torch.cat((h, e1), -1), fc[4], fc[5]))
    h = torch.relu(_lin(h, fc[6], fc[7]))
    alpha = (_lin(h, fc[8], fc[9]) * 10.0).reshape(n, R, S)
    hc = torch.relu(_lin(torch.cat((h, e2), -1), fc[10], fc[11]))
    color = torch.sigmoid(_lin(hc, fc[12], fc[13])).reshape(n, R, S, 3)
    occ = torch.sigmoid(alpha)
    free = (1.0 - occ + 1e-10)[..., :-1]
    T = torch.cumprod(torch.cat((torch.ones(n, R, 1, dtype=pcs.dtype, device=pcs.device), free), -1), -1)
    w = occ * T
    D = (w * z).sum(-1)
    V = (w * (z - D.unsqueeze(-1)) ** 2).sum(-1).detach()
    C = (w.unsqueeze(-1) * color).sum(-2)
    O = w.sum(-1)
    m_o, m_s = sem != 0, sem != 2
    m_dd = depth_mask.bool() & m_o

    def reduce(mat, mask, var=None):                                         # render_rays.py:67-96
        cnt = mask.sum(-1)
        if (cnt == 0).any():
            return torch.zeros(n, dtype=pcs.dtype, device=pcs.device)
        if var is not None:
            mat = mat * (1.0 / (torch.sqrt(var) + 1e-4))
        return mat.sum(-1) / (cnt + 1e-10)

    l_d = reduce((D - gt_depth).abs() * m_dd, m_dd, V)
    l_c = reduce((C - gt_rgb).abs().sum(-1) * m_o, m_o)
    l_o = reduce((O - m_o.to(pcs.dtype)).abs() * m_s, m_s)
    loss = (l_d + l_c * color_scaling + l_o * opacity_scaling).sum()
    return loss, dict(render_depth=D, render_color=C, opacity=O, var=V)


class CpuTrainer:
    """fwd + loss + backward + AdamW on CPU tensors (train.py:293-326 without the data plumbing)."""

    def __init__(self, fc_np, B_np, scale_np, lr=1e-3, weight_decay=0.013, device="cpu", weights_bf16=False):
        """``weights_bf16``: "bf16 weights + fp32 accumulate" (BASELINE configs[3]/[4]): the tensors the optimiser owns are fp32
        masters; every step is evaluated on a copy rounded to bfloat16 whose gradients become the masters' gradients
        (same semantics as oracle.ref_runner.reference_frame(weights_bf16=True))."""
        self.weights_bf16 = weights_bf16
        self.device = torch.device(device)
        self.fc = [torch.from_numpy(a.copy()).to(self.device).requires_grad_() for a in fc_np]
        self.B = torch.from_numpy(B_np.copy()).to(self.device).requires_grad_()
        self.scale = torch.from_numpy(scale_np.copy()).to(self.device)
        self.opt = torch.optim.AdamW(self.fc + [self.B], lr=lr, weight_decay=weight_decay)

    def step(self, batch, update=True):
        args = [batch[k] if torch.is_tensor(batch[k]) else torch.from_numpy(batch[k]).to(self.device)
                for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask")]
        run = self.fc + [self.B]
        if self.weights_bf16:
            run = [p.detach().to(torch.bfloat16).to(p.dtype).requires_grad_() for p in run]
        loss, rend = forward_loss(run[:14], run[14], self.scale, *args)
        if loss.requires_grad:
            loss.backward()
            if self.weights_bf16:
                for m, r in zip(self.fc + [self.B], run):
                    m.grad = r.grad
        grads = [(p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for p in self.fc + [self.B]]
        if update:
            self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        return loss.detach(), rend, grads
