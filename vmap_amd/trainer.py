"""Per-object model holder with the reference's attribute surface (trainer.py:9-33).

``Trainer(cfg)`` exposes ``fc_occ_map``, ``pe``, ``obj_scale``, ``device``, ``hidden_feature_size``, ``obj_id``
and ``bound_extent`` so that code written against the reference's ``sceneObject.trainer`` keeps working.
``cfg`` is any object with the attributes the reference's ``cfg.Config`` provides (cfg.py:6-91).
Mesh extraction (trainer.py:35-75) is outside the hot path and not provided; ``eval_points`` is.
"""
from __future__ import annotations

import torch

from . import fields, layout


class Trainer:
    def __init__(self, cfg):
        self.obj_id = cfg.obj_id
        self.device = cfg.training_device
        self.hidden_feature_size = cfg.hidden_feature_size
        self.obj_scale = cfg.obj_scale
        self.n_unidir_funcs = cfg.n_unidir_funcs
        self.emb_size1 = layout.EMB1
        self.emb_size2 = layout.EMB2
        self.load_network()
        self.bound_extent = 0.995 if self.obj_id == 0 else 0.9      # trainer.py:21-24

    def load_network(self):
        self.fc_occ_map = fields.OccupancyMap(self.emb_size1, self.emb_size2, hidden_size=self.hidden_feature_size)
        self.fc_occ_map.apply(fields.init_weights).to(self.device)
        self.pe = fields.UniDirsEmbed(max_deg=self.n_unidir_funcs, scale=self.obj_scale).to(self.device)

    @torch.no_grad()
    def eval_points(self, points: torch.Tensor, chunk_size: int = 100000):
        """Occupancy and colour at arbitrary points of this object's frame (trainer.py:77-95): ONE launch of the HIP query
        kernel (vmapstep_query_points).  There is no CPU / PyTorch-ops path: points on another device, or a hidden width
        the kernels do not implement, raise (``chunk_size`` is accepted for signature compatibility and unused)."""
        from . import _lib
        if not points.is_cuda:
            raise _lib.VmapStepError("Trainer.eval_points runs on the GPU only (no CPU fallback): move the points to the training device")
        if self.hidden_feature_size % 32 != 0 or not 32 <= self.hidden_feature_size <= 256:
            raise _lib.VmapStepError(f"hidden width {self.hidden_feature_size}: supported widths are multiples of 32 up to 256")
        occ, color = self._eval_points_hip(points)
        if occ.max() == 0:
            return None
        return occ, color

    def _eval_points_hip(self, points: torch.Tensor):
        import ctypes
        from . import _lib
        lib = _lib.load()
        pts = points if points.dtype == torch.float32 else points.float()
        n = pts.shape[0]
        dev = pts.device
        nb = ctypes.c_size_t(0)
        _lib.check(lib.vmapstep_query_workspace_bytes(self.hidden_feature_size, ctypes.byref(nb)), lib)
        ws = torch.empty(nb.value + 256, dtype=torch.uint8, device=dev)
        ws_ptr = ws.data_ptr() + (-ws.data_ptr()) % 256
        pp = _lib.Params()
        for t, p in enumerate(self.fc_occ_map.parameters()):
            if not p.is_contiguous() or p.device != dev:
                raise ValueError("field parameters must be contiguous and on the points' device")
            pp.fc[t] = _lib.Tensor(p.data_ptr(), 0)
        pp.pe_B = _lib.Tensor(self.pe.B_layer.weight.data_ptr(), 0)
        scale = self.pe.scale.reshape(1).to(dev, torch.float32).contiguous()
        sc = _lib.Tensor(scale.data_ptr(), 0)
        occ = torch.empty(n, dtype=torch.float32, device=dev)
        col = torch.empty(n, 3, dtype=torch.float32, device=dev)
        strides = (ctypes.c_int64 * 2)(pts.stride(0), pts.stride(1))
        _lib.check(lib.vmapstep_query_points(self.hidden_feature_size, ctypes.byref(pp), ctypes.byref(sc), 0, pts.data_ptr(), n,
                                             strides, occ.data_ptr(), col.data_ptr(), ws_ptr, nb.value,
                                             torch.cuda.current_stream(dev).cuda_stream), lib)
        return occ, col


class SimpleConfig:
    """The subset of cfg.Config (cfg.py) the hot path reads, with the Replica room0 vMAP values as defaults."""

    def __init__(self, **kw):
        self.obj_id = -1
        self.training_device = "cuda:0"
        self.data_device = "cuda:0"
        self.training_strategy = "hip"        # third option next to the reference's "vmap" / "forloop" (cfg.py:20)
        self.hidden_feature_size = 32
        self.hidden_feature_size_bg = 128
        self.obj_scale = 2.0
        self.bg_scale = 5.0
        self.n_unidir_funcs = 5
        self.n_per_optim = 120
        self.n_per_optim_bg = 1200
        self.n_iter_per_frame = 20
        self.win_size = 5
        self.n_samples_per_frame = self.n_per_optim // self.win_size
        self.n_bins_cam2surface = 1
        self.n_bins_cam2surface_bg = 5
        self.n_bins = 9
        self.learning_rate = 1e-3
        self.weight_decay = 0.013
        self.do_bg = False
        for k, v in kw.items():
            setattr(self, k, v)
