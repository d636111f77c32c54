"""ctypes binding of libvmapstep.so (C ABI in include/vmapstep.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is NO fallback: if the
shared object is missing or a call fails, the product path raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VMAPSTEP_LIBRARY", os.path.join(_HERE, "libvmapstep.so"))   # override: measurement builds only

NUM_FC = 14
ABI_VERSION = 7
WEIGHTS_F32, WEIGHTS_BF16 = 0, 1


KERNEL_AUTO, KERNEL_GEN, KERNEL_WIDE4, KERNEL_H32_F32, KERNEL_WS1, KERNEL_WP, KERNEL_S32_BWD6 = 0, 1, 2, 4, 5, 6, 8


class Tuning(ctypes.Structure):
    """vmapstep_tuning: measurement / test overrides of the automatic launch plan, passed per call through Shape.tuning."""
    _fields_ = [("workgroups_per_object", ctypes.c_int32), ("kernel", ctypes.c_int32), ("generic_finalize", ctypes.c_int32),
                ("ws_flags", ctypes.c_int32)]


class Shape(ctypes.Structure):
    _fields_ = [("n_obj", ctypes.c_int32), ("rays", ctypes.c_int32), ("samples", ctypes.c_int32),
                ("hidden", ctypes.c_int32), ("weight_dtype", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("tuning", ctypes.POINTER(Tuning))]


class PlanInfo(ctypes.Structure):
    """vmapstep_plan_info (ABI v6): the launch plan the library derives from a shape."""
    _fields_ = [("kernel", ctypes.c_char * 48), ("rays_per_round", ctypes.c_int32), ("rounds_per_object", ctypes.c_int32),
                ("workgroups_per_object", ctypes.c_int32), ("tiles_per_round", ctypes.c_int32), ("waves_per_workgroup", ctypes.c_int32),
                ("single_round", ctypes.c_int32)]


class Tensor(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("obj_stride", ctypes.c_int64)]


class Params(ctypes.Structure):
    _fields_ = [("fc", Tensor * NUM_FC), ("pe_B", Tensor)]


class Batch(ctypes.Structure):
    _fields_ = [
        ("pcs", ctypes.c_void_p), ("pcs_stride", ctypes.c_int64 * 4),
        ("z", ctypes.c_void_p), ("z_stride", ctypes.c_int64 * 3),
        ("gt_depth", ctypes.c_void_p), ("gt_depth_stride", ctypes.c_int64 * 2),
        ("gt_rgb", ctypes.c_void_p), ("gt_rgb_stride", ctypes.c_int64 * 3),
        ("sem", ctypes.c_void_p), ("sem_stride", ctypes.c_int64 * 2),
        ("depth_mask", ctypes.c_void_p), ("depth_mask_stride", ctypes.c_int64 * 2),
        # ABI v7: the hand-off as rays (pcs == NULL): point = (ray_o + ray_d * z) - center
        ("ray_o", ctypes.c_void_p), ("ray_o_stride", ctypes.c_int64 * 3),
        ("ray_d", ctypes.c_void_p), ("ray_d_stride", ctypes.c_int64 * 3),
        ("center", ctypes.c_void_p), ("center_stride", ctypes.c_int64),
    ]


class Outputs(ctypes.Structure):
    _fields_ = [("loss", ctypes.c_void_p), ("flags", ctypes.c_void_p), ("render_depth", ctypes.c_void_p),
                ("render_color", ctypes.c_void_p), ("opacity", ctypes.c_void_p), ("var", ctypes.c_void_p),
                ("loss_terms", ctypes.c_void_p)]


class AdamW(ctypes.Structure):
    _fields_ = [("lr", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float),
                ("eps", ctypes.c_float), ("weight_decay", ctypes.c_float), ("step", ctypes.c_int32),
                ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("bias_table", ctypes.c_void_p), ("table_len", ctypes.c_int32), ("step_counter", ctypes.c_void_p)]


class SampleObject(ctypes.Structure):
    _fields_ = [("rgbs", ctypes.c_void_p), ("depth", ctypes.c_void_p), ("t_wc", ctypes.c_void_p), ("bbox", ctypes.c_void_p),
                ("n_keyframes", ctypes.c_int32), ("last2", ctypes.c_int32 * 2), ("center", ctypes.c_float * 3),
                ("obj_id", ctypes.c_int32), ("slots", ctypes.c_void_p), ("inst", ctypes.c_void_p)]


class SampleCfg(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("frames", ctypes.c_int32),
                ("samples_per_frame", ctypes.c_int32), ("n_bins_cam2surface", ctypes.c_int32), ("n_bins", ctypes.c_int32),
                ("fx", ctypes.c_float), ("fy", ctypes.c_float), ("cx", ctypes.c_float), ("cy", ctypes.c_float),
                ("min_depth", ctypes.c_float), ("surface_eps", ctypes.c_float), ("stop_eps", ctypes.c_float)]


class SampleRandoms(ctypes.Structure):
    _fields_ = [("kf_ids", ctypes.c_void_p), ("u_w", ctypes.c_void_p), ("u_h", ctypes.c_void_p), ("u_z", ctypes.c_void_p),
                ("g_z", ctypes.c_void_p)]


EXPORTS = (
    "vmapstep_last_error", "vmapstep_abi_version", "vmapstep_param_layout", "vmapstep_workspace_bytes",
    "vmapstep_fwd_bwd", "vmapstep_render", "vmapstep_train_steps",
    "vmapstep_profile_main_kernel", "vmapstep_profile_phases", "vmapstep_prepare", "vmapstep_train_steps_prepared",
    "vmapstep_workspace_counts_offset", "vmapstep_fwd_bwd_prepared", "vmapstep_sample_frame",
    "vmapstep_query_workspace_bytes", "vmapstep_query_points", "vmapstep_profile_train_steps", "vmapstep_adamw_apply",
    "vmapstep_sample_workspace_bytes", "vmapstep_describe_plan", "vmapstep_sample_frame_rays",
)

_libs = {}


class VmapStepError(RuntimeError):
    pass


def load(path=None):
    """Load the HIP library (``path``: another build of it - the measurement build of tests/tools - next to the product's; each is
    loaded once). Import torch first so that its libamdhip64 is the one both sides share."""
    path = os.path.abspath(path or LIB_PATH)
    if path in _libs:
        return _libs[path]
    import torch  # noqa: F401  (maps torch/lib/libamdhip64.so before our NEEDED entry is resolved)
    if not os.path.exists(path):
        raise VmapStepError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the training step.")
    lib = ctypes.CDLL(path)
    lib.vmapstep_last_error.restype = ctypes.c_char_p
    lib.vmapstep_abi_version.restype = ctypes.c_int
    lib.vmapstep_param_layout.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_int64),
                                          ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    lib.vmapstep_workspace_bytes.argtypes = [ctypes.POINTER(Shape), ctypes.c_int32, ctypes.POINTER(ctypes.c_size_t)]
    lib.vmapstep_fwd_bwd.argtypes = [ctypes.POINTER(Shape), ctypes.POINTER(Params), ctypes.POINTER(Tensor),
                                     ctypes.POINTER(Batch), ctypes.c_float, ctypes.c_float, ctypes.POINTER(Params),
                                     ctypes.POINTER(Outputs), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.vmapstep_render.argtypes = [ctypes.POINTER(Shape), ctypes.POINTER(Params), ctypes.POINTER(Tensor),
                                    ctypes.POINTER(Batch), ctypes.c_float, ctypes.c_float,
                                    ctypes.POINTER(Outputs), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.vmapstep_train_steps.argtypes = [ctypes.POINTER(Shape), ctypes.POINTER(Params), ctypes.POINTER(Tensor),
                                         ctypes.POINTER(Batch), ctypes.c_int64, ctypes.c_int32, ctypes.c_float,
                                         ctypes.c_float, ctypes.POINTER(AdamW), ctypes.POINTER(Params),
                                         ctypes.POINTER(Outputs), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.vmapstep_prepare.argtypes = [ctypes.POINTER(Shape), ctypes.POINTER(Params), ctypes.POINTER(Batch), ctypes.c_int64,
                                     ctypes.c_int32, ctypes.c_void_p, ctypes.c_size_t,
                                     ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
    lib.vmapstep_train_steps_prepared.argtypes = lib.vmapstep_train_steps.argtypes
    lib.vmapstep_fwd_bwd_prepared.argtypes = [ctypes.POINTER(Shape), ctypes.POINTER(Params), ctypes.POINTER(Tensor),
                                              ctypes.POINTER(Batch), ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.POINTER(Params),
                                              ctypes.POINTER(Outputs), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.vmapstep_adamw_apply.argtypes = [ctypes.POINTER(Shape), ctypes.POINTER(Params), ctypes.c_void_p, ctypes.c_int64,
                                         ctypes.POINTER(AdamW), ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                         ctypes.POINTER(Outputs), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.vmapstep_workspace_counts_offset.argtypes = [ctypes.POINTER(Shape), ctypes.c_int32, ctypes.POINTER(ctypes.c_size_t)]
    lib.vmapstep_sample_frame.argtypes = [ctypes.POINTER(SampleCfg), ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(SampleRandoms), ctypes.c_void_p, ctypes.c_size_t,
                                          ctypes.c_void_p]
    lib.vmapstep_sample_frame_rays.argtypes = [ctypes.POINTER(SampleCfg), ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(SampleRandoms), ctypes.c_void_p, ctypes.c_size_t,
                                               ctypes.c_void_p]
    lib.vmapstep_sample_workspace_bytes.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_size_t)]
    lib.vmapstep_query_workspace_bytes.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_size_t)]
    lib.vmapstep_query_points.argtypes = [ctypes.c_int32, ctypes.POINTER(Params), ctypes.POINTER(Tensor), ctypes.c_int32,
                                          ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.vmapstep_profile_train_steps.argtypes = [ctypes.POINTER(Shape), ctypes.POINTER(Params), ctypes.POINTER(Tensor),
                                                 ctypes.POINTER(Batch), ctypes.c_int64, ctypes.c_int32, ctypes.c_float,
                                                 ctypes.c_float, ctypes.POINTER(AdamW), ctypes.POINTER(Outputs),
                                                 ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    lib.vmapstep_profile_main_kernel.argtypes = [ctypes.POINTER(Shape), ctypes.POINTER(Params), ctypes.POINTER(Tensor),
                                                 ctypes.POINTER(Batch), ctypes.c_int32, ctypes.c_void_p,
                                                 ctypes.c_size_t, ctypes.c_void_p]
    lib.vmapstep_profile_phases.argtypes = [ctypes.POINTER(Shape), ctypes.POINTER(Params), ctypes.POINTER(Tensor),
                                            ctypes.POINTER(Batch), ctypes.c_void_p, ctypes.c_size_t,
                                            ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p, ctypes.c_size_t,
                                            ctypes.c_void_p]
    lib.vmapstep_describe_plan.argtypes = [ctypes.POINTER(Shape), ctypes.c_int32, ctypes.POINTER(PlanInfo)]
    for fn in ("vmapstep_describe_plan", "vmapstep_param_layout", "vmapstep_workspace_bytes", "vmapstep_fwd_bwd", "vmapstep_render",
               "vmapstep_train_steps", "vmapstep_profile_main_kernel",
               "vmapstep_profile_phases", "vmapstep_prepare", "vmapstep_train_steps_prepared",
               "vmapstep_workspace_counts_offset", "vmapstep_fwd_bwd_prepared", "vmapstep_sample_frame",
               "vmapstep_query_workspace_bytes", "vmapstep_query_points", "vmapstep_profile_train_steps", "vmapstep_adamw_apply",
               "vmapstep_sample_workspace_bytes", "vmapstep_sample_frame_rays"):
        getattr(lib, fn).restype = ctypes.c_int
    if lib.vmapstep_abi_version() != ABI_VERSION:
        raise VmapStepError(f"ABI mismatch: library {lib.vmapstep_abi_version()} != binding {ABI_VERSION}")
    _libs[path] = lib
    return lib


def describe_plan(n_obj: int, rays: int, samples: int, hidden: int, weights_bf16: bool = False, max_steps: int = 20, tuning: dict = None,
                  library: str = None) -> dict:
    """The launch plan for a shape as a dict (kernel name, rays per round, rounds / workgroups per object, tiles per round, waves per
    workgroup, single_round) - no device needed.  ``library``: the build to ask (default: the product library)."""
    lib = load(library)
    sh = Shape(n_obj, rays, samples, hidden, WEIGHTS_BF16 if weights_bf16 else WEIGHTS_F32)
    t = Tuning(**tuning) if tuning else None
    if t is not None:
        sh.tuning = ctypes.pointer(t)
    info = PlanInfo()
    check(lib.vmapstep_describe_plan(ctypes.byref(sh), max_steps, ctypes.byref(info)), lib)
    return {"kernel": info.kernel.decode(), **{k: int(getattr(info, k)) for k, _ in PlanInfo._fields_[1:]}}


def check(rc: int, lib):
    """Raise on a non-zero status; the message comes from the library that made the call (each build has its own thread-local
    last-error string)."""
    if rc != 0:
        msg = lib.vmapstep_last_error().decode("utf-8", "replace")
        raise VmapStepError(f"vmapstep error {rc}: {msg}")
