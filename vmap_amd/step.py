"""Host side of the fused per-object training step: PyTorch tensors in, HIP kernels via the C ABI.

Drop-in for the call triple of the reference step loop (train.py:293-294 forward, :303-306 loss, :324
backward) and, through ``train_steps``, for the whole 20-iteration loop of one frame including the AdamW
update (train.py:270-326).  PyTorch is plumbing here (device memory, current stream); all arithmetic runs in
``libvmapstep.so``.  There is no fallback path: a missing library or an unsupported shape raises.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import torch

from . import _lib, layout


def _obj_dense(t: torch.Tensor) -> bool:
    return t.dim() == 1 or t[0].is_contiguous()


class RayPoints:
    """The sample points of a batch / frame given as RAYS instead of the points tensor (ABI v7; SURVEY.md 8(f) row 1, second half):
    ``origins`` / ``dirs`` float32 [n, R, 3] (world frame), ``centers`` float32 [n, 3] or None (= zeros).  Pass an instance wherever
    an operator takes ``pcs``: the kernels rebuild point s of ray r as (origins + dirs * z[..., s]) - centers, every operation rounded
    on its own (vmap.py:452-454) - bit-identical to the [n, R, S, 3] tensor it replaces, at 24 + 4 S instead of 16 S bytes per ray.
    ``sampler.FrameSampler(..., rays=True)`` produces one per frame."""

    def __init__(self, origins: torch.Tensor, dirs: torch.Tensor, centers: Optional[torch.Tensor] = None):
        one = origins.dim() == 2                        # a one-object frame without the object dimension: [R, 3] (+ centre [3]); see unsqueeze()
        if origins.shape != dirs.shape or origins.dim() not in (2, 3) or origins.shape[-1] != 3:
            raise ValueError(f"origins / dirs: need two [n, R, 3] tensors, got {tuple(origins.shape)} / {tuple(dirs.shape)}")
        if centers is not None and (tuple(centers.shape) != ((3,) if one else (origins.shape[0], 3)) or centers.stride(-1) != 1):
            raise ValueError(f"centers: need {'[3]' if one else '[n, 3]'} with unit inner stride, got {tuple(centers.shape)}")
        self.origins, self.dirs, self.centers = origins, dirs, centers

    @classmethod
    def __new_unchecked(cls, origins, dirs, centers):
        r = cls.__new__(cls)
        r.origins, r.dirs, r.centers = origins, dirs, centers
        return r

    @property
    def rays(self) -> int:             # rays per object
        return int(self.origins.shape[-2])

    @property
    def shape(self):                   # (n, rays, None, 3) like the points tensor, for callers that read pcs.shape[0] / [1]; the sample
                                       # count is not a property of the rays - read it from z
        if self.origins.dim() != 3:
            raise ValueError("RayPoints.shape: a one-object bundle without the object dimension ([R, 3]) has no [n, R, S, 3] shape; "
                             "unsqueeze(0) it first (RayPoints.rays gives its ray count)")
        return tuple(self.origins.shape[:2]) + (None, 3)

    def __getitem__(self, idx):
        """Slices along (object, ray) like ``pcs[:, i*R:(i+1)*R]`` (train.py:271-272); an object slice applies to the centres too."""
        idx = idx if isinstance(idx, tuple) else (idx,)
        if len(idx) > 2 or not all(isinstance(i, slice) for i in idx) or self.origins.dim() != 3:
            raise IndexError("RayPoints supports [object_slice, ray_slice] of an [n, R, 3] bundle")
        c = self.centers[idx[0]] if self.centers is not None else None
        return RayPoints(self.origins[idx], self.dirs[idx], c)

    def unsqueeze(self, dim: int) -> "RayPoints":
        """``unsqueeze(0)`` of a one-object frame given without the object dimension (origins / dirs [R, 3], centres [3] or None): what the
        background classes of ``vmap_amd.parallel`` do to every tensor of their frame."""
        if dim != 0:
            raise IndexError("RayPoints.unsqueeze: only dim 0")
        return RayPoints.__new_unchecked(self.origins.unsqueeze(0), self.dirs.unsqueeze(0),
                                         self.centers.reshape(1, 3) if self.centers is not None else None)

    def signature(self):
        """(address, shape, strides) of the tensors: what a caller compares to see whether the frame buffers moved"""
        return tuple((t.data_ptr(), tuple(t.shape), tuple(t.stride())) for t in (self.origins, self.dirs, self.centers) if t is not None)

    def points(self, z: torch.Tensor) -> torch.Tensor:
        """The [n, R, S, 3] tensor these rays stand for (host-side helper for tests / tools; eager torch ops round each operation on
        its own like the kernels do)."""
        p = self.origins.unsqueeze(2) + self.dirs.unsqueeze(2) * z.unsqueeze(-1)
        return p - self.centers[:, None, None, :] if self.centers is not None else p


class StepResult:
    """What one fused step produced (device tensors; nothing is synchronised)."""

    def __init__(self, loss, flags, render_depth=None, render_color=None, opacity=None, var=None):
        self.loss = loss                # [n_steps] float32
        self.flags = flags              # [n_steps, 4] int32: drop_depth, drop_colour, drop_opacity, explode
        self.render_depth, self.render_color, self.opacity, self.var = render_depth, render_color, opacity, var


class FusedAdamWState:
    """Optimiser state of one stacked ensemble: moments as [n, padded_params] slabs + the step count.

    A fresh state (zeros, step 0) is what the reference gets whenever ``update_vmap`` re-stacks the objects
    and adds a new param group (utils.py:33): Adam moments restart - preserved here on purpose.
    """

    def __init__(self, n_obj: int, hidden: int, device, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.013):
        self.padded = (layout.param_count(hidden) + 63) // 64 * 64      # == vmapstep_param_layout(...).padded_params
        self.exp_avg = torch.zeros(n_obj, self.padded, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.step = 0
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay

    def c_struct(self, device_steps: bool = False) -> _lib.AdamW:
        if device_steps:
            return _lib.AdamW(self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step,
                              self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.bias_table.data_ptr(),
                              self.bias_table.shape[0], self.step_counter.data_ptr())
        return _lib.AdamW(self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step,
                          self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), None, 0, None)

    # ---- device-resident step count (for frame calls replayed as a graph: their kernel arguments are frozen) ----------
    bias_table: Optional[torch.Tensor] = None
    step_counter: Optional[torch.Tensor] = None

    def enable_device_steps(self, max_len: int = 1 << 16):
        """Build the table the kernels index with the device-side step count: for t = 1, 2, ... the two step-dependent AdamW
        scalars exactly as the host path forms them (double arithmetic, rounded once to float32: lr / (1 - beta1^t) and
        sqrt(1 - beta2^t)), until both stop changing in float32 (beta 0.9 / 0.999: 16 610 entries) - later steps use the
        last entry.  The counter starts at the host count; from here on ``vmapstep_train_steps`` advances it on the device."""
        import math
        if self.bias_table is not None:
            return
        import numpy as np
        rows = []
        # the C side forms them from the float32 members of vmapstep_adamw widened to double: do the same, bit for bit
        lr, b1, b2 = (float(np.float32(x)) for x in (self.lr, self.betas[0], self.betas[1]))
        last = None
        saturated = False
        for t in range(1, max_len + 1):
            cur = (np.float32(lr / (1.0 - math.pow(b1, float(t)))), np.float32(math.sqrt(1.0 - math.pow(b2, float(t)))))
            rows.append(cur)
            if cur == (np.float32(lr), np.float32(1.0)) and cur == last:          # saturated at the exact limits
                saturated = True
                break
            last = cur
        if not saturated:
            # the kernels clamp the step count to the last entry: a table that stops short of the limits would hand later steps
            # wrong bias corrections without a sound (beta2 = 0.9999 saturates near step 166 000)
            raise ValueError(f"AdamW bias-correction table did not saturate within {max_len} steps (betas={self.betas}): "
                             "pass a larger max_len to enable_device_steps()")
        dev = self.exp_avg.device
        self.bias_table = torch.tensor(np.asarray(rows, dtype=np.float32), device=dev).contiguous()
        self.step_counter = torch.tensor([self.step, 0], dtype=torch.int32, device=dev)

    def note_host_steps(self, n: int):
        """A call that took the step count from the host (prepared / apply paths) advanced the optimiser by n steps: keep the
        device-side count in step (a tiny stream-ordered add; only when both kinds of calls are mixed on one state)."""
        if self.step_counter is not None:
            # on the state's OWN device and that device's current stream (= the stream the operator's launches are on), whatever
            # device is current on the calling thread
            with torch.cuda.device(self.step_counter.device):
                self.step_counter[1] += n


class VmapStep:
    """The fused step operator for a fixed (n_obj, rays, samples, hidden) problem shape."""

    def __init__(self, n_obj: int, rays: int, samples: int, hidden: int, device="cuda:0", max_steps: int = 32,
                 color_scaling: float = 5.0, opacity_scaling: float = 10.0, weights: str = "f32", tuning: Optional[dict] = None,
                 library: Optional[str] = None):
        """``tuning``: optional overrides of the automatic launch plan for measurements / A-B tests (fields of
        ``vmapstep_tuning``: workgroups_per_object, kernel, generic_finalize, ws_flags).  They belong to THIS operator (the C library
        keeps no tuning state).  ``library``: path of another build of the C ABI to run this operator on (default: the product
        library next to this file, ``_lib.LIB_PATH``).  The operator may live on any GPU of the process: every C call runs on the
        device that owns the stream it is given (``torch.cuda.current_stream(self.device)``), whatever device is current on the thread."""
        self.lib = _lib.load(library)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.VmapStepError("VmapStep runs on the GPU only (no CPU fallback)")
        if weights not in ("f32", "bf16"):
            raise ValueError("weights must be 'f32' or 'bf16'")
        self.shape = _lib.Shape(n_obj, rays, samples, hidden, _lib.WEIGHTS_BF16 if weights == "bf16" else _lib.WEIGHTS_F32)
        self._tuning = None
        if tuning:
            self._tuning = _lib.Tuning(**tuning)          # kept alive by the operator; the shape points at it
            self.shape.tuning = ctypes.pointer(self._tuning)
        self.n_obj, self.rays, self.samples, self.hidden = n_obj, rays, samples, hidden
        self.max_steps = max_steps
        self.color_scaling, self.opacity_scaling = float(color_scaling), float(opacity_scaling)
        nbytes = ctypes.c_size_t(0)
        _lib.check(lib=self.lib, rc=self.lib.vmapstep_workspace_bytes(ctypes.byref(self.shape), max_steps, ctypes.byref(nbytes)))
        self.workspace = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=self.device)
        off = (-self.workspace.data_ptr()) % 256
        self._ws_ptr = self.workspace.data_ptr() + off
        self._ws_bytes = nbytes.value
        self._shapes = layout.fc_shapes(hidden)

    # ---- argument marshalling -------------------------------------------------------------------------
    def _params(self, fc: Sequence[torch.Tensor], B: torch.Tensor, what="params") -> _lib.Params:
        if len(fc) != _lib.NUM_FC:
            raise ValueError(f"{what}: expected {_lib.NUM_FC} field tensors, got {len(fc)}")
        p = _lib.Params()
        for t, (x, shp) in enumerate(zip(fc, self._shapes)):
            self._check(x, (self.n_obj,) + tuple(shp), f"{what}.fc[{t}]")
            p.fc[t] = _lib.Tensor(x.data_ptr(), x.stride(0))
        self._check(B, (self.n_obj,) + layout.PE_B_SHAPE, f"{what}.B")
        p.pe_B = _lib.Tensor(B.data_ptr(), B.stride(0))
        return p

    def _check(self, x: torch.Tensor, shape, name: str):
        if tuple(x.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {tuple(x.shape)} != {tuple(shape)}")
        if x.dtype != torch.float32 or x.device != self.device:
            raise ValueError(f"{name}: need float32 on {self.device}, got {x.dtype} on {x.device}")
        if not _obj_dense(x):
            raise ValueError(f"{name}: each object's block must be dense")

    def _batch(self, pcs, z, gt_depth, gt_rgb, sem, depth_mask, rays_total: Optional[int] = None) -> _lib.Batch:
        n, R, S = self.n_obj, (rays_total or self.rays), self.samples
        rays = pcs if isinstance(pcs, RayPoints) else None
        named = [("z", z, (n, R, S), torch.float32), ("gt_depth", gt_depth, (n, R), torch.float32), ("gt_rgb", gt_rgb, (n, R, 3), torch.float32)]
        named += [("pcs", pcs, (n, R, S, 3), torch.float32)] if rays is None else \
                 [("origins", rays.origins, (n, R, 3), torch.float32), ("dirs", rays.dirs, (n, R, 3), torch.float32)]
        if rays is not None and rays.centers is not None:
            named.append(("centers", rays.centers, (n, 3), torch.float32))
        for name, x, shp, dt in named:
            if tuple(x.shape) != shp or x.dtype != dt or x.device != self.device:
                raise ValueError(f"{name}: need {dt} {shp} on {self.device}, got {x.dtype} {tuple(x.shape)} on {x.device}")
        if sem.dtype != torch.uint8 or tuple(sem.shape) != (n, R):
            raise ValueError(f"sem: need uint8 {(n, R)}, got {sem.dtype} {tuple(sem.shape)}")
        if depth_mask.dtype == torch.bool:
            depth_mask = depth_mask.view(torch.uint8)      # same bytes (vmap.py:376 bool mask)
        if depth_mask.dtype != torch.uint8 or tuple(depth_mask.shape) != (n, R):
            raise ValueError(f"depth_mask: need bool/uint8 {(n, R)}")
        b = _lib.Batch()
        b.z, b.gt_depth, b.gt_rgb = z.data_ptr(), gt_depth.data_ptr(), gt_rgb.data_ptr()
        b.sem, b.depth_mask = sem.data_ptr(), depth_mask.data_ptr()
        if rays is None:
            b.pcs = pcs.data_ptr()
            b.pcs_stride[:] = pcs.stride()
        else:
            b.pcs = None
            b.ray_o, b.ray_d = rays.origins.data_ptr(), rays.dirs.data_ptr()
            b.ray_o_stride[:] = rays.origins.stride()
            b.ray_d_stride[:] = rays.dirs.stride()
            if rays.centers is not None:
                if rays.centers.stride(-1) != 1:             # the ABI carries one stride for the centres (per object); bundles built by
                    raise ValueError("centers: need unit inner stride")      # unsqueeze() / slicing bypass __init__'s check
                b.center, b.center_stride = rays.centers.data_ptr(), rays.centers.stride(0)
        b.z_stride[:] = z.stride()
        b.gt_depth_stride[:] = gt_depth.stride()
        b.gt_rgb_stride[:] = gt_rgb.stride()
        b.sem_stride[:] = sem.stride()
        b.depth_mask_stride[:] = depth_mask.stride()
        return b

    def _outputs(self, n_steps: int, render: bool):
        dev = self.device
        loss = torch.empty(n_steps, dtype=torch.float32, device=dev)
        flags = torch.empty(n_steps, 4, dtype=torch.int32, device=dev)
        res = StepResult(loss, flags)
        o = _lib.Outputs(loss.data_ptr(), flags.data_ptr(), None, None, None, None, None)
        if render:
            n, R = self.n_obj, self.rays
            res.render_depth = torch.empty(n, R, dtype=torch.float32, device=dev)
            res.render_color = torch.empty(n, R, 3, dtype=torch.float32, device=dev)
            res.opacity = torch.empty(n, R, dtype=torch.float32, device=dev)
            res.var = torch.empty(n, R, dtype=torch.float32, device=dev)
            o.render_depth, o.render_color = res.render_depth.data_ptr(), res.render_color.data_ptr()
            o.opacity, o.var = res.opacity.data_ptr(), res.var.data_ptr()
        return res, o

    def plan(self) -> dict:
        """The launch plan of this operator (``vmapstep_describe_plan``): kernel name, rays per round, rounds and workgroups per
        object, tiles per round, waves per workgroup, single_round."""
        info = _lib.PlanInfo()
        _lib.check(lib=self.lib, rc=self.lib.vmapstep_describe_plan(ctypes.byref(self.shape), self.max_steps, ctypes.byref(info)))
        return {"kernel": info.kernel.decode(), **{k: int(getattr(info, k)) for k, _ in _lib.PlanInfo._fields_[1:]}}

    def _stream(self) -> int:
        # the current stream of THIS operator's device (not of whatever device happens to be current)
        return torch.cuda.current_stream(self.device).cuda_stream

    # ---- operators ------------------------------------------------------------------------------------
    def _ws_view(self, byte_offset: int, nbytes: int) -> torch.Tensor:
        start = (self._ws_ptr - self.workspace.data_ptr()) + byte_offset
        return self.workspace[start:start + nbytes]

    def fwd_bwd(self, fc, B, pe_scale, pcs, z, gt_depth, gt_rgb, sem, depth_mask, grads_fc=None, grad_B=None,
                render: bool = False, prepared_step: Optional[int] = None, loss_terms: Optional[torch.Tensor] = None) -> StepResult:
        """Loss + gradients of all 15 stacked tensors (train.py:293-306 + :324). Gradients are written to
        ``grads_fc``/``grad_B`` if given, else into freshly allocated ``p.grad`` of the parameters.

        ``prepared_step``: run step i of a frame prepared with ``prepare_frame`` (mask counts / switches possibly reduced
        over ranks by the caller, parameter image kept current by ``adamw_apply``); the batch tensors are that step's slice.
        ``loss_terms``: optional float32 [n, 4] device tensor receiving, per object, the depth / colour / opacity terms before their
        weights and l_batch (loss.py:59) - what a ray-sharded caller sums over ranks next to the gradients."""
        if grads_fc is None:
            grads_fc = []
            for p in list(fc) + [B]:
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                grads_fc.append(p.grad)
            grad_B = grads_fc.pop()
        pp = self._params(fc, B)
        gp = self._params(grads_fc, grad_B, "grads")
        sc = _lib.Tensor(pe_scale.data_ptr(), pe_scale.stride(0) if pe_scale.dim() else 0)
        bt = self._batch(pcs, z, gt_depth, gt_rgb, sem, depth_mask)
        res, out = self._outputs(1, render)
        if loss_terms is not None:
            if tuple(loss_terms.shape) != (self.n_obj, 4) or loss_terms.dtype != torch.float32 or not loss_terms.is_contiguous() \
                    or loss_terms.device != self.device:
                raise ValueError(f"loss_terms: need contiguous float32 {(self.n_obj, 4)} on {self.device}")
            out.loss_terms = loss_terms.data_ptr()
        if prepared_step is not None:
            _lib.check(lib=self.lib, rc=self.lib.vmapstep_fwd_bwd_prepared(ctypes.byref(self.shape), ctypes.byref(pp), ctypes.byref(sc), ctypes.byref(bt),
                                                          int(prepared_step), self.color_scaling, self.opacity_scaling,
                                                          ctypes.byref(gp), ctypes.byref(out), self._ws_ptr, self._ws_bytes, self._stream()))
        else:
            _lib.check(lib=self.lib, rc=self.lib.vmapstep_fwd_bwd(ctypes.byref(self.shape), ctypes.byref(pp), ctypes.byref(sc), ctypes.byref(bt),
                                                 self.color_scaling, self.opacity_scaling, ctypes.byref(gp), ctypes.byref(out),
                                                 self._ws_ptr, self._ws_bytes, self._stream()))
        return res

    def prepare_frame(self, fc, B, pcs, z, gt_depth, gt_rgb, sem, depth_mask, n_steps: int, ray_step: Optional[int] = None):
        """vmapstep_prepare for a whole frame ([n, rays_total, ...] tensors): packs the parameter image and computes the
        per-step mask counts and empty-mask switches.  Returns (counts float32 [n_steps, n, 4], flags int32 [n_steps, 4]) as
        VIEWS of the workspace: a ray-sharded caller sums the counts over ranks and rewrites the flags in place, an
        object-sharded one max-reduces the flags - ONE collective per frame either way."""
        if n_steps > self.max_steps:
            raise ValueError(f"n_steps={n_steps} > max_steps={self.max_steps} this operator was sized for")
        ray_step = self.rays if ray_step is None else int(ray_step)
        pp = self._params(fc, B)
        bt = self._batch(pcs, z, gt_depth, gt_rgb, sem, depth_mask, rays_total=pcs.shape[1])
        foff, coff = ctypes.c_size_t(0), ctypes.c_size_t(0)
        _lib.check(lib=self.lib, rc=self.lib.vmapstep_prepare(ctypes.byref(self.shape), ctypes.byref(pp), ctypes.byref(bt), ray_step, n_steps,
                                             self._ws_ptr, self._ws_bytes, ctypes.byref(foff), self._stream()))
        _lib.check(lib=self.lib, rc=self.lib.vmapstep_workspace_counts_offset(ctypes.byref(self.shape), n_steps, ctypes.byref(coff)))
        counts = self._ws_view(coff.value, n_steps * self.n_obj * 16).view(torch.float32).view(n_steps, self.n_obj, 4)
        flags = self._ws_view(foff.value, n_steps * 16).view(torch.int32).view(n_steps, 4)
        return counts, flags

    def adamw_apply(self, fc, B, grad_slab: torch.Tensor, opt: "FusedAdamWState", loss_terms: Optional[torch.Tensor] = None,
                    step_index: int = 0, loss_out: Optional[torch.Tensor] = None, flags_out: Optional[torch.Tensor] = None):
        """torch.optim.AdamW's update from an externally reduced gradient slab ([n, padded_params] float32, flat parameter
        order) + rewrite of the packed parameter image (vmapstep_adamw_apply); ``opt.step`` is advanced.

        ``loss_terms`` (float32 [n, 4], the rank-summed ``fwd_bwd(..., loss_terms=)`` rows): the SAME launch also writes the
        step's global loss to ``loss_out[0]`` and its flags to ``flags_out[0:4]`` (the reduced empty-mask switches of prepared
        step ``step_index`` + render_rays.py:88-90's explode test on the summed terms)."""
        if tuple(grad_slab.shape) != (self.n_obj, opt.padded) or grad_slab.dtype != torch.float32 or not grad_slab.is_contiguous() \
                or grad_slab.device != self.device:
            raise ValueError(f"grad_slab: need contiguous float32 {(self.n_obj, opt.padded)} on {self.device}")
        pp = self._params(fc, B)
        oc = opt.c_struct()
        out, lt = None, None
        if loss_terms is not None:
            # the same checks fwd_bwd applies to its loss_terms: the kernel reads n_obj rows of four floats through a raw pointer
            if tuple(loss_terms.shape) != (self.n_obj, 4) or loss_terms.dtype != torch.float32 or not loss_terms.is_contiguous() \
                    or loss_terms.device != self.device:
                raise ValueError(f"loss_terms: need contiguous float32 {(self.n_obj, 4)} on {self.device}")
            if loss_out is None or flags_out is None or loss_out.dtype != torch.float32 or flags_out.dtype != torch.int32 \
                    or loss_out.numel() < 1 or flags_out.numel() < 4 or not flags_out.is_contiguous() \
                    or loss_out.device != self.device or flags_out.device != self.device:
                raise ValueError(f"loss_terms given: loss_out (float32 [>=1]) and flags_out (contiguous int32 [>=4]) on {self.device} are required")
            out = ctypes.byref(_lib.Outputs(loss_out.data_ptr(), flags_out.data_ptr(), None, None, None, None, None))
            lt = loss_terms.data_ptr()
        _lib.check(lib=self.lib, rc=self.lib.vmapstep_adamw_apply(ctypes.byref(self.shape), ctypes.byref(pp), grad_slab.data_ptr(), opt.padded,
                                                 ctypes.byref(oc), lt, int(step_index), self.color_scaling, self.opacity_scaling, out,
                                                 self._ws_ptr, self._ws_bytes, self._stream()))
        opt.step += 1
        opt.note_host_steps(1)

    def render(self, fc, B, pe_scale, pcs, z, gt_depth, gt_rgb, sem, depth_mask) -> StepResult:
        """Forward + loss only: rendered depth / colour / opacity / variance (loss.py:24-31)."""
        pp = self._params(fc, B)
        sc = _lib.Tensor(pe_scale.data_ptr(), pe_scale.stride(0) if pe_scale.dim() else 0)
        bt = self._batch(pcs, z, gt_depth, gt_rgb, sem, depth_mask)
        res, out = self._outputs(1, True)
        _lib.check(lib=self.lib, rc=self.lib.vmapstep_render(ctypes.byref(self.shape), ctypes.byref(pp), ctypes.byref(sc),
                                            ctypes.byref(bt), self.color_scaling, self.opacity_scaling,
                                            ctypes.byref(out), self._ws_ptr, self._ws_bytes, self._stream()))
        return res

    def profile_main_kernel(self, fc, B, pe_scale, pcs, z, gt_depth, gt_rgb, sem, depth_mask, reps: int) -> float:
        """Average duration (ms) of one launch of the dominant kernel, timed with events on the current stream."""
        pp = self._params(fc, B)
        sc = _lib.Tensor(pe_scale.data_ptr(), pe_scale.stride(0) if pe_scale.dim() else 0)
        bt = self._batch(pcs, z, gt_depth, gt_rgb, sem, depth_mask)
        args = (ctypes.byref(self.shape), ctypes.byref(pp), ctypes.byref(sc), ctypes.byref(bt))
        _lib.check(lib=self.lib, rc=self.lib.vmapstep_profile_main_kernel(*args, 3, self._ws_ptr, self._ws_bytes, self._stream()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        _lib.check(lib=self.lib, rc=self.lib.vmapstep_profile_main_kernel(*args, reps, self._ws_ptr, self._ws_bytes, self._stream()))
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def profile_phases(self, fc, B, pe_scale, pcs, z, gt_depth, gt_rgb, sem, depth_mask):
        """In-kernel phase timestamps of one forward+backward launch: int64 array [workgroups, 4, 16] (shader clocks,
        relative to the earliest stamp)."""
        pp = self._params(fc, B)
        sc = _lib.Tensor(pe_scale.data_ptr(), pe_scale.stride(0) if pe_scale.dim() else 0)
        bt = self._batch(pcs, z, gt_depth, gt_rgb, sem, depth_mask)
        cap = 8 * ((self.n_obj + 7) // 8) * 512 * 4 * 16      # up to 512 workgroups per object (step_main_wp at hidden 64: two per CU)
        buf = torch.zeros(cap, dtype=torch.int32, device=self.device)
        nwg = ctypes.c_int32(0)
        _lib.check(lib=self.lib, rc=self.lib.vmapstep_profile_phases(ctypes.byref(self.shape), ctypes.byref(pp), ctypes.byref(sc),
                                                    ctypes.byref(bt), buf.data_ptr(), cap, ctypes.byref(nwg),
                                                    self._ws_ptr, self._ws_bytes, self._stream()))
        torch.cuda.synchronize()
        t = buf[: nwg.value * 64].cpu().numpy().astype("int64") & 0xFFFFFFFF
        t = t.reshape(nwg.value, 4, 16)
        t = t[t[:, 0, 15] != 0]                  # idle blocks of the XCD-affine grid never stamp
        return t - t[:, :, 0].min()

    def train_steps(self, fc, B, pe_scale, pcs, z, gt_depth, gt_rgb, sem, depth_mask, opt: FusedAdamWState,
                    n_steps: int, ray_step: Optional[int] = None, grads_fc=None, grad_B=None,
                    render: bool = False, flag_reduce=None) -> StepResult:
        """The step loop of one frame (train.py:270-326): step i trains on rays [i*ray_step, i*ray_step+R) of the
        per-frame tensors ([n, rays_total, ...]) and applies the fused AdamW update in place.

        ``flag_reduce``: optional callable taking the int32 [n_steps, 4] device tensor of this rank's empty-mask switches
        and max-reducing it in place across ranks (vmap_amd.parallel); when given, the call is split into
        vmapstep_prepare -> flag_reduce -> vmapstep_train_steps_prepared."""
        if n_steps > self.max_steps:
            raise ValueError(f"n_steps={n_steps} > max_steps={self.max_steps} this operator was sized for")
        ray_step = self.rays if ray_step is None else int(ray_step)
        rays_total = pcs.shape[1]
        if (n_steps - 1) * ray_step + self.rays > rays_total:
            raise ValueError(f"frame tensors hold {rays_total} rays, need {(n_steps - 1) * ray_step + self.rays}")
        pp = self._params(fc, B)
        gp = self._params(grads_fc, grad_B, "grads") if grads_fc is not None else None
        sc = _lib.Tensor(pe_scale.data_ptr(), pe_scale.stride(0) if pe_scale.dim() else 0)
        bt = self._batch(pcs, z, gt_depth, gt_rgb, sem, depth_mask, rays_total=rays_total)
        res, out = self._outputs(n_steps, render)
        on_device = opt.bias_table is not None and flag_reduce is None
        oc = opt.c_struct(device_steps=on_device)
        fn = self.lib.vmapstep_train_steps
        if flag_reduce is not None:
            off = ctypes.c_size_t(0)
            _lib.check(lib=self.lib, rc=self.lib.vmapstep_prepare(ctypes.byref(self.shape), ctypes.byref(pp), ctypes.byref(bt), ray_step,
                                                 n_steps, self._ws_ptr, self._ws_bytes, ctypes.byref(off), self._stream()))
            flags_view = self._ws_view(off.value, n_steps * 16).view(torch.int32).view(n_steps, 4)
            flag_reduce(flags_view)
            fn = self.lib.vmapstep_train_steps_prepared
        _lib.check(lib=self.lib, rc=fn(ctypes.byref(self.shape), ctypes.byref(pp), ctypes.byref(sc),
                      ctypes.byref(bt), ray_step, n_steps, self.color_scaling,
                      self.opacity_scaling, ctypes.byref(oc),
                      ctypes.byref(gp) if gp is not None else None, ctypes.byref(out),
                      self._ws_ptr, self._ws_bytes, self._stream()))
        opt.step += n_steps
        if not on_device:
            opt.note_host_steps(n_steps)
        return res


class BoundFrame:
    """``VmapStep.train_steps`` with the argument marshalling done ONCE: the ctypes blocks of the stacked parameters, the
    per-frame sample tensors and the outputs are built at bind time (the tensors are kept alive here), so a frame call is
    one C call.  For callers whose buffers do not move between frames (the slab of ``driver.HipMapper``, a sampler that
    writes into fixed frame tensors, ``bench.py``): per-call Python shrinks from ~0.1 ms of shape checks and struct
    filling - a tenth of a 20-step frame - to the call itself.  Outputs (loss [max_steps], flags [max_steps, 4]) are
    reused by every call: read or copy them before the next one.

    ``graph=True`` (opt-in; single-rank callers only): the frame call - 1 + 2 n launches - is captured ONCE per step count as
    a hipGraph and replayed; the optimiser's step count then lives on the device (``FusedAdamWState.enable_device_steps``),
    because a replayed launch cannot take a new scalar from the host.  Same kernels, same arguments, bit-identical results
    (tests/test_gpu_parity.py).  Off by default because it buys nothing on this stack: 29.58 / 29.76 us per step replayed
    against 29.44 / 29.61 launched kernel by kernel at the headline shape (profiles/r03i_graph_replay.json; round 1 had
    measured +2 % on a slower step) - what separates consecutive kernels is the dispatch boundary on the device, not the host."""

    def __init__(self, op: "VmapStep", fc, B, pe_scale, pcs, z, gt_depth, gt_rgb, sem, depth_mask, opt: "FusedAdamWState",
                 ray_step: Optional[int] = None, render: bool = False, flag_reduce=None, graph: bool = False):
        self.op, self.opt, self.flag_reduce = op, opt, flag_reduce
        self.ray_step = op.rays if ray_step is None else int(ray_step)
        self.rays_total = pcs.shape[1]
        self._keep = (list(fc), B, pe_scale, pcs, z, gt_depth, gt_rgb, sem, depth_mask)
        self._pp = op._params(fc, B)
        self._sc = _lib.Tensor(pe_scale.data_ptr(), pe_scale.stride(0) if pe_scale.dim() else 0)
        self._bt = op._batch(pcs, z, gt_depth, gt_rgb, sem, depth_mask, rays_total=self.rays_total)
        self.result, self._out = op._outputs(op.max_steps, render)
        self.graph = bool(graph) and flag_reduce is None
        self._graphs, self._warm = {}, set()
        if self.graph:
            opt.enable_device_steps()

    def _call(self, n_steps: int, stream: int):
        op, opt = self.op, self.opt
        lib, sh = op.lib, ctypes.byref(op.shape)
        fn = lib.vmapstep_train_steps
        if self.flag_reduce is not None:
            off = ctypes.c_size_t(0)
            _lib.check(lib=lib, rc=lib.vmapstep_prepare(sh, ctypes.byref(self._pp), ctypes.byref(self._bt), self.ray_step, n_steps,
                                            op._ws_ptr, op._ws_bytes, ctypes.byref(off), stream))
            self.flag_reduce(op._ws_view(off.value, n_steps * 16).view(torch.int32).view(n_steps, 4))
            fn = lib.vmapstep_train_steps_prepared
        on_device = opt.bias_table is not None and self.flag_reduce is None
        oc = opt.c_struct(device_steps=on_device)
        _lib.check(lib=lib, rc=fn(sh, ctypes.byref(self._pp), ctypes.byref(self._sc), ctypes.byref(self._bt), self.ray_step, n_steps,
                      op.color_scaling, op.opacity_scaling, ctypes.byref(oc), None, ctypes.byref(self._out),
                      op._ws_ptr, op._ws_bytes, stream))
        return on_device

    def train_steps(self, n_steps: int) -> StepResult:
        op, opt = self.op, self.opt
        if n_steps > op.max_steps or (n_steps - 1) * self.ray_step + op.rays > self.rays_total:
            raise ValueError(f"n_steps={n_steps}: the bound frame holds {self.rays_total} rays, max_steps={op.max_steps}")
        g = self._graphs.get(n_steps) if self.graph else None
        if g is not None:
            g.replay()                                   # on the current stream of the device
            opt.step += n_steps
            return self.result
        if self.graph and n_steps in self._warm:
            # second call with this step count: capture it (nothing runs during capture), then replay the capture as this call
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._call(n_steps, torch.cuda.current_stream(op.device).cuda_stream)
            self._graphs[n_steps] = g
            g.replay()
            opt.step += n_steps
            return self.result
        # the CURRENT stream of the operator's device at every call (a stream frozen at bind time could be stale - or destroyed -
        # by the time a later frame runs under another stream: nothing would order it against the sampler's writes)
        on_device = self._call(n_steps, op._stream())    # first call of a graph-bound frame: eager (sets the kernels' per-device attributes once)
        self._warm.add(n_steps)
        opt.step += n_steps
        if not on_device:
            opt.note_host_steps(n_steps)
        return self.result


def _bind(self, fc, B, pe_scale, pcs, z, gt_depth, gt_rgb, sem, depth_mask, opt: "FusedAdamWState", ray_step=None,
          render: bool = False, flag_reduce=None, graph: bool = False) -> BoundFrame:
    """Marshal once, call many times: see ``BoundFrame``.  Every call runs on the operator's device's current stream."""
    return BoundFrame(self, fc, B, pe_scale, pcs, z, gt_depth, gt_rgb, sem, depth_mask, opt, ray_step, render, flag_reduce, graph)


VmapStep.bind = _bind


def _profile_train_steps(self, fc, B, pe_scale, pcs, z, gt_depth, gt_rgb, sem, depth_mask, opt: "FusedAdamWState", n_steps: int) -> float:
    """Average duration (ms) of the dominant kernel over ``n_steps`` real training steps, in the prep / main / finalize sequence
    of ``train_steps``; waits for the device.  Returns (the dispatch's own begin -> end time - what a kernel trace reports -,
    a pair of stream events around the launch)."""
    pp = self._params(fc, B)
    sc = _lib.Tensor(pe_scale.data_ptr(), pe_scale.stride(0) if pe_scale.dim() else 0)
    bt = self._batch(pcs, z, gt_depth, gt_rgb, sem, depth_mask, rays_total=pcs.shape[1])
    res, out = self._outputs(n_steps, False)
    oc = opt.c_struct()
    ms = (ctypes.c_float * 2)(0.0, 0.0)
    _lib.check(lib=self.lib, rc=self.lib.vmapstep_profile_train_steps(ctypes.byref(self.shape), ctypes.byref(pp), ctypes.byref(sc), ctypes.byref(bt),
                                                     self.rays, n_steps, self.color_scaling, self.opacity_scaling, ctypes.byref(oc),
                                                     ctypes.byref(out), self._ws_ptr, self._ws_bytes, self._stream(), ms))
    opt.step += n_steps
    opt.note_host_steps(n_steps)
    return float(ms[0]), float(ms[1])


VmapStep.profile_train_steps = _profile_train_steps


class _BatchLossFn(torch.autograd.Function):
    """``batch_loss`` of train.py:303-306 as a differentiable tensor: the fused kernel produces the loss AND all 15
    gradients in its forward; ``backward`` hands them to autograd scaled by the incoming gradient, so the reference's
    ``batch_loss += bg_loss; batch_loss.backward(); optimiser.step()`` (train.py:316-325) works unchanged with any
    ``torch.optim`` optimiser."""

    @staticmethod
    def forward(ctx, op, pe_scale, batch, *params):
        fc, B = list(params[:-1]), params[-1]
        grads = [torch.empty_like(p) for p in params]
        with torch.no_grad():
            res = op.fwd_bwd([p.detach() for p in fc], B.detach(), pe_scale, *batch, grads_fc=grads[:-1], grad_B=grads[-1])
        ctx.grads = grads
        ctx.flags = res.flags
        return res.loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        return (None, None, None) + tuple(gr * g for gr in ctx.grads)


def batch_loss(op: "VmapStep", fc, B, pe_scale, pcs, z, gt_depth, gt_rgb, sem, depth_mask) -> torch.Tensor:
    """0-dim loss tensor with a grad_fn: ``batch_loss(...).backward()`` fills ``p.grad`` of the stacked parameters
    (``fc``: the 14 stacked field tensors, ``B``: the stacked ``B_layer.weight``), like ``loss.step_batch_loss(...)``
    followed by ``.backward()`` in the reference."""
    return _BatchLossFn.apply(op, pe_scale, (pcs, z, gt_depth, gt_rgb, sem, depth_mask), *fc, B)
