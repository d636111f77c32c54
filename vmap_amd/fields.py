"""Per-object field modules with the reference's parameter names, shapes and order.

``OccupancyMap`` mirrors reference ``model.py:16-85`` and ``UniDirsEmbed`` mirrors ``embedding.py:43-91`` at
the state-dict level (same attribute paths -> checkpoints written by either side load in the other, the
stacking order of ``update_vmap`` is identical).  Their ``forward`` is ordinary PyTorch and exists for the
host-side uses of a single object (mesh queries, checkpoints, CPU tests of the host logic).  The training
step never calls it: that path is the HIP library (``vmap_amd.step``), with no PyTorch fallback.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import layout
from .synth import ICOSA_DIRS


def _block(n_in: int, n_out: int) -> nn.Sequential:
    # index 0 = Linear, index 1 = ReLU: gives the reference's "<name>.0.weight" parameter paths
    return nn.Sequential(nn.Linear(n_in, n_out), nn.ReLU())


class OccupancyMap(nn.Module):
    """Occupancy + colour MLP of one object (4 hidden blocks, skip connection, two heads)."""

    def __init__(self, emb_size1: int = layout.EMB1, emb_size2: int = layout.EMB2, hidden_size: int = 32):
        super().__init__()
        self.embedding_size1 = emb_size1
        self.embedding_size2 = emb_size2
        self.hidden_size = hidden_size
        self.in_layer = _block(emb_size1, hidden_size)
        self.mid1 = nn.Sequential(_block(hidden_size, hidden_size))
        self.cat_layer = _block(hidden_size + emb_size1, hidden_size)
        self.mid2 = nn.Sequential(_block(hidden_size, hidden_size))
        self.out_alpha = nn.Linear(hidden_size, 1)
        self.color_linear = _block(emb_size2 + hidden_size, hidden_size)
        self.out_color = nn.Linear(hidden_size, 3)

    def forward(self, emb: torch.Tensor):
        low = emb[..., : self.embedding_size1]
        high = emb[..., self.embedding_size1:]
        h = self.mid1(self.in_layer(low))
        h = self.mid2(self.cat_layer(torch.cat((h, low), dim=-1)))
        alpha = self.out_alpha(h) * 10.0
        color = torch.sigmoid(self.out_color(self.color_linear(torch.cat((h, high), dim=-1))))
        return alpha, color


def init_weights(m: nn.Module):
    """Xavier-normal on every Linear weight (what the reference applies at trainer.py:32)."""
    if isinstance(m, nn.Linear):
        nn.init.xavier_normal_(m.weight)


class UniDirsEmbed(nn.Module):
    """Directional Fourier encoding: xyz/scale projected on 21 trainable directions, 6 octaves of sin(pi .)."""

    def __init__(self, min_deg: int = 0, max_deg: int = 5, scale: float = 2.0):
        super().__init__()
        self.min_deg, self.max_deg = min_deg, max_deg
        self.n_freqs = max_deg - min_deg + 1
        self.B_layer = nn.Linear(3, layout.N_DIRS, bias=False)
        with torch.no_grad():
            self.B_layer.weight.copy_(torch.from_numpy(ICOSA_DIRS))
        bands = 2.0 ** torch.linspace(float(min_deg), float(max_deg), self.n_freqs)
        self.register_buffer("frequency_bands", bands, persistent=False)
        self.register_buffer("scale", torch.tensor(float(scale)), persistent=True)

    def forward(self, x: torch.Tensor):
        t = x / self.scale
        xb = (self.B_layer(t).unsqueeze(-2) * self.frequency_bands.view(-1, 1)).flatten(-2)
        return torch.cat((t, torch.sin(xb * math.pi)), dim=-1)
