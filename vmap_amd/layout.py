"""Parameter layout of one object field (occupancy/colour MLP + directional encoding).

Mirrors the tensors the reference creates per object and stacks with
``combine_state_for_ensemble``:

* field MLP  : reference ``model.py:16-52`` (``OccupancyMap.__init__``), 14 tensors in
  ``nn.Module.parameters()`` order;
* encoding   : reference ``embedding.py:43-80`` (``UniDirsEmbed.__init__``), one trainable
  ``B_layer.weight`` [21, 3] plus the buffers ``frequency_bands`` [6] and ``scale`` [].

Sizes follow ``trainer.py:16-17``: emb_size1 = 21*(3+1)+3 = 87, emb_size2 = 129-87 = 42.
Everything here is plain integer bookkeeping shared by the host wrapper, the tests and bench.
"""
from __future__ import annotations

N_DIRS = 21          # icosahedron directions, embedding.py:51-73
N_FREQS = 6          # 2**0 .. 2**5, embedding.py:78 with max_deg = n_unidir_funcs = 5
EMB = 3 + N_DIRS * N_FREQS      # 129
EMB1 = 3 + N_DIRS * 4           # 87  (xyz + octaves 0..3)  -> in_layer / cat_layer
EMB2 = EMB - EMB1               # 42  (octaves 4..5)        -> color_linear
N_FC = 14

FC_NAMES = (
    "in_layer.0.weight", "in_layer.0.bias",
    "mid1.0.0.weight", "mid1.0.0.bias",
    "cat_layer.0.weight", "cat_layer.0.bias",
    "mid2.0.0.weight", "mid2.0.0.bias",
    "out_alpha.weight", "out_alpha.bias",
    "color_linear.0.weight", "color_linear.0.bias",
    "out_color.weight", "out_color.bias",
)


def fc_shapes(H: int):
    """Shapes of the 14 field tensors for hidden width H (model.py:28-49)."""
    return (
        (H, EMB1), (H,),
        (H, H), (H,),
        (H, H + EMB1), (H,),
        (H, H), (H,),
        (1, H), (1,),
        (H, H + EMB2), (H,),
        (3, H), (3,),
    )


def numel(shape) -> int:
    n = 1
    for s in shape:
        n *= int(s)
    return n


def fc_sizes(H: int):
    return tuple(numel(s) for s in fc_shapes(H))


def fc_param_count(H: int) -> int:
    """H(4H+225)+4: 11 300 at H=32, 30 788 at H=64, 94 340 at H=128, 319 748 at H=256."""
    return sum(fc_sizes(H))


PE_B_SHAPE = (N_DIRS, 3)
PE_B_SIZE = N_DIRS * 3


def param_count(H: int) -> int:
    """All trainable scalars of one object: field MLP + B_layer.weight."""
    return fc_param_count(H) + PE_B_SIZE


def flat_offsets(H: int):
    """Start offset of each of the 15 trainable tensors (14 field + B) in the flat per-object order."""
    offs = []
    o = 0
    for s in fc_sizes(H):
        offs.append(o)
        o += s
    offs.append(o)  # B_layer.weight
    return tuple(offs)


def macs_per_point(H: int) -> int:
    """Forward multiply-accumulates per sample point: H(4H+220)."""
    return H * (4 * H + 220)


def step_flops(n_obj: int, R: int, S: int, H: int) -> int:
    """Algorithmic FLOPs of one training step (fwd 2*MAC + bwd 4*MAC), SURVEY.md 8(d)."""
    return n_obj * R * S * 6 * macs_per_point(H)


def step_bytes(n_obj: int, R: int, S: int, H: int) -> int:
    """Algorithmic HBM bytes of one step: sample data once, weights read once, grads written once."""
    return n_obj * (R * (16 * S + 18) + 8 * param_count(H))


def stack_in_slab(tensors, pe_B):
    """The 15 stacked tensors re-homed as views of ONE ``[n, P]`` slab in flat order (what ``vmap_amd.driver`` does at every
    re-stack): same shapes and values, but the per-frame step loop recognises the layout (one base pointer, object stride P)
    and its optimiser pass indexes the slab directly instead of looking every element's tensor up.
    Returns (slab, [14 field views], B view)."""
    import torch
    n = tensors[0].shape[0]
    H = tensors[2].shape[-1]
    offs = flat_offsets(H)
    slab = torch.empty(n, param_count(H), dtype=torch.float32, device=tensors[0].device)
    views = []
    for t, src in enumerate(list(tensors) + [pe_B]):
        v = slab[:, offs[t]:offs[t] + numel(src.shape[1:])].view(src.shape)
        v.copy_(src)
        views.append(v)
    return slab, views[:-1], views[-1]
