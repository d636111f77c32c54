"""Object-list <-> stacked-tensor plumbing (the reference's ``utils.update_vmap`` and write-back).

``update_vmap(models, optimiser)`` keeps the reference signature (utils.py:30-34): it stacks every parameter
and buffer of the per-object modules on a new leading dimension, marks the stacked parameters as trainable
leaves and registers them as a NEW param group of the optimiser - which, exactly as in the reference, means
the Adam moments of previously stacked tensors are abandoned on every re-stack.  It returns
``(fmodel, params, buffers)`` like ``functorch.combine_state_for_ensemble`` does; ``fmodel(params_k, buffers_k, x)``
is a functional single-object forward (usable under ``torch.vmap`` for the reference's own strategy).
``write_back`` is train.py:331-338.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
from torch.func import functional_call


def update_vmap(models: Sequence[torch.nn.Module], optimiser=None):
    if len(models) == 0:
        raise ValueError("update_vmap: empty object list")
    base = models[0]
    p_names = [n for n, _ in base.named_parameters()]
    b_names = [n for n, _ in base.named_buffers()]
    params = tuple(torch.stack([dict(m.named_parameters())[n].detach() for m in models]).requires_grad_()
                   for n in p_names)
    buffers = tuple(torch.stack([dict(m.named_buffers())[n] for m in models]) for n in b_names)
    if optimiser is not None:
        optimiser.add_param_group({"params": list(params)})      # utils.py:33

    def fmodel(p, b, *args, **kwargs):
        state = {n: t for n, t in zip(p_names, p)}
        state.update({n: t for n, t in zip(b_names, b)})
        return functional_call(base, state, args, kwargs)

    return fmodel, params, buffers


@torch.no_grad()
def write_back(models: Sequence[torch.nn.Module], params: Sequence[torch.Tensor]):
    """Copy the trained stacked parameters into each object's module (train.py:331-338)."""
    for k, m in enumerate(models):
        for i, p in enumerate(m.parameters()):
            p.copy_(params[i][k])


def shard_objects(n_obj: int, world_size: int, rank: int) -> List[int]:
    """Object ids owned by ``rank``: round-robin by insertion order (objects are independent units)."""
    return list(range(rank, n_obj, world_size))
