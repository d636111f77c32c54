"""TEST / BASELINE INFRASTRUCTURE ONLY - runs the *real* reference (kxhit/vMAP) hot path.

This module imports ``model.py``, ``embedding.py``, ``render_rays.py`` and ``loss.py`` unmodified - from the source tree
``/root/reference`` (read-only, present only in the authoring container) or, where that tree does not exist (the GPU box),
from ``oracle/_ref/``: the same four files byte-compiled by ``oracle/make_ref.py`` (sourceless ``.pyc``, sha256 of every
source recorded in ``oracle/_ref/MANIFEST.json``; no reference source is copied into this repository) - and drives them
exactly the way the reference does:

* ``utils.update_vmap``   (utils.py:30-34)  -> ``combine_state_for_ensemble`` + ``requires_grad_``
* ``train.py:293-294``    -> ``vmap(pe_model)(...)``, ``vmap(fc_model)(...)``
* ``train.py:303-306``    -> ``loss.step_batch_loss``
* ``train.py:324-326``    -> ``backward()``, ``AdamW.step()``, ``zero_grad``

Users: ``tests/golden/make_*.py`` (the committed fixtures that pin the oracle), the ``-m gpu`` test that regenerates a fixture
on the GPU box, and ``bench.py``'s baseline legs (``ReferenceTrainer``: ``cpu_baseline`` kind "reference" and
``gpu_reference_baseline``).  Nothing in the product package may import it.
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get("VMAP_REFERENCE_ROOT", "/root/reference")
_COMPILED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
_mods = None
SOURCE = None          # "source tree <path>" | "compiled <path> (sha256 in MANIFEST.json)" once imported


def reference_available() -> bool:
    """Can the reference's modules be imported here (source tree, or the compiled files of oracle/make_ref.py)?"""
    if os.path.isdir(REFERENCE_ROOT):
        return True
    from oracle import make_ref
    return make_ref.available()


def _import_reference():
    global _mods, SOURCE
    if _mods is not None:
        return _mods
    import importlib
    if os.path.isdir(REFERENCE_ROOT):
        root, SOURCE = REFERENCE_ROOT, f"source tree {REFERENCE_ROOT}"
    else:
        from oracle import make_ref
        if not make_ref.available():
            raise RuntimeError(f"reference not available: no source tree at {REFERENCE_ROOT} and no compiled modules in {_COMPILED} "
                               "(python oracle/make_ref.py builds them where the source tree exists)")
        root, SOURCE = _COMPILED, f"compiled {_COMPILED} (byte-compiled unmodified; source sha256 in MANIFEST.json)"
    sys.dont_write_bytecode = True
    if root not in sys.path:
        sys.path.insert(0, root)
    mods = {}
    for name in ("model", "embedding", "render_rays", "loss"):        # loss.py does `import render_rays`: the directory is on sys.path
        mods[name] = importlib.import_module(name)
        assert os.path.dirname(os.path.abspath(mods[name].__file__)) == os.path.abspath(root), mods[name].__file__
    _mods = mods
    return mods


def build_reference_models(fc_np, B_np, scale_np, H, dtype=torch.float32):
    """One ``OccupancyMap`` + ``UniDirsEmbed`` per object (trainer.py:26-33) loaded with given values."""
    mods = _import_reference()
    from vmap_amd import layout
    n = B_np.shape[0]
    fc_models, pe_models = [], []
    for k in range(n):
        m = mods["model"].OccupancyMap(layout.EMB1, layout.EMB2, hidden_size=H)
        with torch.no_grad():
            for p, a in zip(m.parameters(), fc_np):
                p.copy_(torch.from_numpy(a[k]))
        pe = mods["embedding"].UniDirsEmbed(max_deg=5, scale=float(scale_np[k]))
        with torch.no_grad():
            pe.B_layer.weight.copy_(torch.from_numpy(B_np[k]))
        fc_models.append(m.to(dtype))
        pe_models.append(pe.to(dtype))
    return fc_models, pe_models


class ExitTrap:
    """``render_rays.reduce_batch_loss`` ends the PROCESS on "loss explode" (render_rays.py:88-90: ``print("loss explode"); exit(-1)``).
    Inside this context the name ``exit`` resolves, for that module only, to a recorder (a module global shadows the builtin; the
    reference's code is untouched), so the call is counted and the function goes on to return the loss it had computed - the
    values a caller that survives the check (ours: the device flag VMAPSTEP_FLAG_EXPLODE) must reproduce.  ``codes``: the exit codes
    the reference asked for, in order."""

    def __init__(self):
        self.mod = _import_reference()["render_rays"]
        self.codes = []

    def __enter__(self):
        self.mod.exit = lambda code=0: self.codes.append(code)
        return self

    def __exit__(self, *exc):
        del self.mod.exit
        return False


def reference_step(fc_np, B_np, scale_np, batch, H, dtype=torch.float32, strategy="vmap",
                   adamw_steps=0, lr=1e-3, weight_decay=0.013):
    """Run the reference training step; returns dict of numpy arrays.

    ``adamw_steps`` > 0 additionally runs that many full optimisation steps (same batch each step)
    with ``torch.optim.AdamW`` the way train.py:67,324-326 does and returns the final parameters.
    """
    mods = _import_reference()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from functorch import combine_state_for_ensemble, vmap
    render_rays, loss_mod = mods["render_rays"], mods["loss"]

    fc_models, pe_models = build_reference_models(fc_np, B_np, scale_np, H, dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fc_model, fc_param, fc_buffer = combine_state_for_ensemble(fc_models)
        pe_model, pe_param, pe_buffer = combine_state_for_ensemble(pe_models)
    [p.requires_grad_() for p in fc_param]
    [p.requires_grad_() for p in pe_param]

    pcs = torch.from_numpy(batch["pcs"]).to(dtype)
    z = torch.from_numpy(batch["z"]).to(dtype)
    gt_depth = torch.from_numpy(batch["gt_depth"]).to(dtype)
    gt_rgb = torch.from_numpy(batch["gt_rgb"]).to(dtype)
    sem = torch.from_numpy(batch["sem"])
    dmask = torch.from_numpy(batch["depth_mask"]).bool()

    def forward():
        if strategy == "vmap":
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                emb = vmap(pe_model)(pe_param, pe_buffer, pcs)
                alpha, color = vmap(fc_model)(fc_param, fc_buffer, emb)
        else:  # "forloop", train.py:278-290
            al, co = [], []
            for k in range(len(fc_models)):
                e = pe_model([p[k] for p in pe_param], [b[k] for b in pe_buffer], pcs[k])
                a, c = fc_model([p[k] for p in fc_param], [b[k] for b in fc_buffer], e)
                al.append(a)
                co.append(c)
            alpha, color = torch.stack(al), torch.stack(co)
        return alpha, color

    alpha, color = forward()
    # re-derive the rendered quantities the way loss.py:24-31 does (it does not return them)
    a2 = alpha.squeeze(-1)
    occ = render_rays.occupancy_activation(a2)
    term = render_rays.occupancy_to_termination(occ, is_batch=True)
    r_depth = render_rays.render(term, z)
    var = render_rays.render(term, (z - r_depth[..., None]) ** 2)
    r_color = render_rays.render(term[..., None], color, dim=-2)
    r_opacity = term.sum(-1)

    loss, _ = loss_mod.step_batch_loss(alpha, color, gt_depth, gt_rgb, sem, dmask, z)
    out = {
        "loss": loss.detach().numpy().astype(np.float64),
        "alpha": a2.detach().numpy(), "color": color.detach().numpy(),
        "render_depth": r_depth.detach().numpy(), "render_color": r_color.detach().numpy(),
        "opacity": r_opacity.detach().numpy(), "var": var.detach().numpy(),
    }
    if loss.requires_grad:
        loss.backward()
        for t, p in enumerate(fc_param):
            out[f"g_fc{t}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
        pB = pe_param[0]
        out["g_B"] = (pB.grad if pB.grad is not None else torch.zeros_like(pB)).numpy().copy()
    else:  # every loss term dropped (render_rays.py:68-73): no graph, all gradients are zero/None
        for t, p in enumerate(fc_param):
            out[f"g_fc{t}"] = torch.zeros_like(p).numpy()
        out["g_B"] = torch.zeros_like(pe_param[0]).numpy()

    if adamw_steps > 0:
        # train.py:67 builds the optimiser on a dummy variable and adds groups (utils.py:33)
        opt = torch.optim.AdamW([torch.autograd.Variable(torch.tensor(0.0))], lr=lr, weight_decay=weight_decay)
        opt.add_param_group({"params": fc_param})
        opt.add_param_group({"params": pe_param})
        for p in list(fc_param) + list(pe_param):
            p.grad = None
        losses = []
        for _ in range(adamw_steps):
            alpha, color = forward()
            l, _ = loss_mod.step_batch_loss(alpha, color, gt_depth, gt_rgb, sem, dmask, z)
            if l.requires_grad:
                l.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            losses.append(float(l))
        out["adamw_losses"] = np.array(losses, dtype=np.float64)
        for t, p in enumerate(fc_param):
            out[f"p_fc{t}"] = p.detach().numpy().copy()
        out["p_B"] = pe_param[0].detach().numpy().copy()
    return out


def reference_frame(fc_np, B_np, scale_np, frame, H, rays_per_step, n_steps, dtype=torch.float32,
                    lr=1e-3, weight_decay=0.013, weights_bf16=False, strategy="vmap", first_step_grad_delta=None):
    """The reference's OWN step loop over one frame (train.py:270-326, vmap strategy): the per-frame sample tensors
    ``[n, n_steps * rays_per_step, ...]`` are sliced with ``data_idx = slice(i * R, (i + 1) * R)`` on dimension 1 (strided
    views, exactly like train.py:271-277), every step is vmap(pe) -> vmap(fc) -> loss.step_batch_loss -> backward ->
    ``AdamW.step()`` -> ``zero_grad(set_to_none=True)`` on an optimiser built like train.py:67 + utils.py:33.

    ``weights_bf16`` (BASELINE configs[3]/[4] "bf16 weights + fp32 accumulate"; the reference itself has no reduced
    precision, train.py:64-66): the stacked tensors the optimiser owns stay full-precision MASTERS; every step evaluates
    the unmodified reference modules on a copy of the masters rounded to bfloat16 (round-to-nearest-even, all 15
    tensors), and the gradients of that copy become the masters' ``.grad`` before ``AdamW.step()`` - the semantics of
    ``vmapstep_shape::weight_dtype = VMAPSTEP_WEIGHTS_BF16`` (include/vmapstep.h).

    ``first_step_grad_delta`` (15 arrays shaped like the stacked tensors, or None): added to the FIRST step's gradients before
    ``AdamW.step()`` - the trajectory the same loop follows when its float32 evaluation of step 0 lands on the other side of a
    ReLU kink (a hidden unit whose pre-activation lies inside forward rounding of 0: value ~0 either way, derivative bit 0 or 1;
    the delta is the exact effect of that bit, oracle.vmap_oracle.kink_deltas).  Both branches are valid float32 evaluations of
    the reference; tests/golden/make_frame_goldens.py stores both where the reference's own run and the oracle disagree on a bit.

    Returns the per-step losses (float64 array), the final parameters and the gradients of the FIRST step."""
    mods = _import_reference()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from functorch import combine_state_for_ensemble, vmap
    loss_mod = mods["loss"]
    fc_models, pe_models = build_reference_models(fc_np, B_np, scale_np, H, dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fc_model, fc_param, fc_buffer = combine_state_for_ensemble(fc_models)
        pe_model, pe_param, pe_buffer = combine_state_for_ensemble(pe_models)
    [p.requires_grad_() for p in fc_param]
    [p.requires_grad_() for p in pe_param]
    opt = torch.optim.AdamW([torch.autograd.Variable(torch.tensor(0.0))], lr=lr, weight_decay=weight_decay)   # train.py:67
    opt.add_param_group({"params": fc_param})                                                                  # utils.py:33
    opt.add_param_group({"params": pe_param})

    N_pcs = torch.from_numpy(frame["pcs"]).to(dtype)
    N_z = torch.from_numpy(frame["z"]).to(dtype)
    N_gt_depth = torch.from_numpy(frame["gt_depth"]).to(dtype)
    N_gt_rgb = torch.from_numpy(frame["gt_rgb"]).to(dtype)
    N_sem = torch.from_numpy(frame["sem"])
    N_dmask = torch.from_numpy(frame["depth_mask"]).bool()
    R = rays_per_step
    losses, first_grads = [], None
    for it in range(n_steps):
        data_idx = slice(it * R, (it + 1) * R)                                   # train.py:271
        pcs, z = N_pcs[:, data_idx, ...], N_z[:, data_idx, ...]
        gt_depth, gt_rgb = N_gt_depth[:, data_idx, ...], N_gt_rgb[:, data_idx, ...]
        sem, dmask = N_sem[:, data_idx, ...], N_dmask[:, data_idx, ...]
        if weights_bf16:
            run_fc = tuple(p.detach().to(torch.bfloat16).to(dtype).requires_grad_() for p in fc_param)
            run_pe = tuple(p.detach().to(torch.bfloat16).to(dtype).requires_grad_() for p in pe_param)
        else:
            run_fc, run_pe = fc_param, pe_param
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if strategy == "vmap":
                emb = vmap(pe_model)(run_pe, pe_buffer, pcs)                     # train.py:293
                alpha, color = vmap(fc_model)(run_fc, fc_buffer, emb)            # train.py:294
            else:                                                                # "forloop", train.py:278-290: the reference's other path
                al, co = [], []
                for k in range(len(fc_models)):
                    e = pe_model([p[k] for p in run_pe], [b[k] for b in pe_buffer], pcs[k])
                    a_k, c_k = fc_model([p[k] for p in run_fc], [b[k] for b in fc_buffer], e)
                    al.append(a_k)
                    co.append(c_k)
                alpha, color = torch.stack(al), torch.stack(co)
        l, _ = loss_mod.step_batch_loss(alpha, color, gt_depth.detach(), gt_rgb.detach(), sem.detach(), dmask.detach(),
                                        z.detach())                             # train.py:303-306
        if l.requires_grad:
            l.backward()                                                         # train.py:324
        if weights_bf16:
            for m_, r_ in zip(list(fc_param) + list(pe_param), list(run_fc) + list(run_pe)):
                m_.grad = r_.grad
        if it == 0:
            first_grads = [(p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy().copy()
                           for p in list(fc_param) + list(pe_param)]
            if first_step_grad_delta is not None:
                with torch.no_grad():
                    for p, d in zip(list(fc_param) + list(pe_param), first_step_grad_delta):
                        if p.grad is not None:
                            p.grad += torch.from_numpy(np.ascontiguousarray(d)).to(p.grad.dtype)
        opt.step()                                                               # train.py:325
        opt.zero_grad(set_to_none=True)                                          # train.py:326
        losses.append(float(l))
    out = {"losses": np.array(losses, dtype=np.float64)}
    for t, p in enumerate(fc_param):
        out[f"p_fc{t}"] = p.detach().numpy().copy()
        out[f"g0_fc{t}"] = first_grads[t]
    out["p_B"] = pe_param[0].detach().numpy().copy()
    out["g0_B"] = first_grads[14]
    return out


class ReferenceTrainer:
    """The reference's vectorised training step as train.py runs it, on any torch device - the live BASELINE of bench.py:
    ``utils.update_vmap`` (utils.py:30-34: ``combine_state_for_ensemble`` + a new AdamW param group on an optimiser built like
    train.py:67), then per step ``vmap(pe_model)`` / ``vmap(fc_model)`` (train.py:293-294), ``loss.step_batch_loss`` (:303-306, with
    its host-synchronising ``.any()`` checks), ``backward()``, ``AdamW.step()``, ``zero_grad(set_to_none=True)`` (:324-326).
    The modules are the reference's own, unmodified (source tree or oracle/_ref)."""

    def __init__(self, fc_np, B_np, scale_np, hidden, device="cpu", lr=1e-3, weight_decay=0.013):
        mods = _import_reference()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            from functorch import combine_state_for_ensemble, vmap
        self._vmap = vmap
        self.loss_mod = mods["loss"]
        self.device = torch.device(device)
        fc_models, pe_models = build_reference_models(fc_np, B_np, scale_np, hidden)
        fc_models = [m.to(self.device) for m in fc_models]
        pe_models = [m.to(self.device) for m in pe_models]
        self.optimiser = torch.optim.AdamW([torch.autograd.Variable(torch.tensor(0.0))], lr=lr, weight_decay=weight_decay)   # train.py:67
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            self.fc_model, self.fc_param, self.fc_buffer = combine_state_for_ensemble(fc_models)      # utils.py:31
            self.pe_model, self.pe_param, self.pe_buffer = combine_state_for_ensemble(pe_models)
        [p.requires_grad_() for p in self.fc_param]                                                    # utils.py:32
        [p.requires_grad_() for p in self.pe_param]
        self.optimiser.add_param_group({"params": self.fc_param})                                      # utils.py:33
        self.optimiser.add_param_group({"params": self.pe_param})

    def to_device(self, batch):
        """numpy batch -> the tensors train.py:255-260 keeps on the training device"""
        t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(self.device) for k, v in batch.items()}
        t["depth_mask"] = t["depth_mask"].bool()
        return t

    def step(self, b):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            emb = self._vmap(self.pe_model)(self.pe_param, self.pe_buffer, b["pcs"])                   # train.py:293
            alpha, color = self._vmap(self.fc_model)(self.fc_param, self.fc_buffer, emb)               # train.py:294
        loss, _ = self.loss_mod.step_batch_loss(alpha, color, b["gt_depth"].detach(), b["gt_rgb"].detach(), b["sem"].detach(),
                                                b["depth_mask"].detach(), b["z"].detach())             # train.py:303-306
        if loss.requires_grad:
            loss.backward()                                                                            # train.py:324
        self.optimiser.step()                                                                          # train.py:325
        self.optimiser.zero_grad(set_to_none=True)                                                     # train.py:326
        return loss.detach()
