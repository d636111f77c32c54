"""CPU tier: libvmapstep.so loads, exports every symbol include/vmapstep.h declares, and its argument checks work
without a GPU (no kernel is launched here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from vmap_amd import _lib, layout


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "vmapstep.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vmapstep_[a-z_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared_functions() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in _declared_functions():
        assert hasattr(lib, name), name
    assert lib.vmapstep_abi_version() == _lib.ABI_VERSION == 7


def test_struct_layouts_match_the_header(tmp_path):
    """The ctypes mirrors of vmap_amd/_lib.py against include/vmapstep.h as a C compiler lays it out (gcc, C99): size of every struct
    that crosses the ABI and the offsets of the fields ABI v7 appended to vmapstep_batch."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    structs = {"vmapstep_tuning": _lib.Tuning, "vmapstep_shape": _lib.Shape, "vmapstep_plan_info": _lib.PlanInfo, "vmapstep_tensor": _lib.Tensor,
               "vmapstep_params": _lib.Params, "vmapstep_batch": _lib.Batch, "vmapstep_outputs": _lib.Outputs, "vmapstep_adamw": _lib.AdamW,
               "vmapstep_sample_object": _lib.SampleObject, "vmapstep_sample_cfg": _lib.SampleCfg, "vmapstep_sample_randoms": _lib.SampleRandoms}
    fields = ("pcs", "z", "depth_mask", "ray_o", "ray_o_stride", "ray_d", "ray_d_stride", "center", "center_stride")
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "vmapstep.h"\nint main(void) {\n'
                   + "".join(f'  printf("{n} %zu\\n", sizeof({n}));\n' for n in structs)
                   + "".join(f'  printf("batch.{f} %zu\\n", offsetof(vmapstep_batch, {f}));\n' for f in fields)
                   + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for n, c in structs.items():
        assert int(got[n]) == ctypes.sizeof(c), n
    for f in fields:
        assert int(got[f"batch.{f}"]) == getattr(_lib.Batch, f).offset, f


@pytest.mark.parametrize("H", [32, 64, 128, 256])
def test_param_layout_matches_reference_shapes(H):
    lib = _lib.load()
    sizes = (ctypes.c_int64 * 15)()
    P, PP = ctypes.c_int64(), ctypes.c_int64()
    assert lib.vmapstep_param_layout(H, sizes, ctypes.byref(P), ctypes.byref(PP)) == 0
    assert list(sizes)[:14] == list(layout.fc_sizes(H)) and sizes[14] == 63
    assert P.value == layout.param_count(H) == H * (4 * H + 225) + 4 + 63
    assert PP.value % 64 == 0 and PP.value >= P.value


def test_workspace_and_error_reporting():
    lib = _lib.load()
    nbytes = ctypes.c_size_t()
    sh = _lib.Shape(20, 120, 10, 32, 0)
    assert lib.vmapstep_workspace_bytes(ctypes.byref(sh), 20, ctypes.byref(nbytes)) == 0
    assert nbytes.value > 20 * 10 * layout.param_count(32) * 4
    bad = _lib.Shape(20, 120, 10, 48, 0)
    assert lib.vmapstep_workspace_bytes(ctypes.byref(bad), 1, ctypes.byref(nbytes)) == -2
    assert b"hidden=48" in lib.vmapstep_last_error()
    assert lib.vmapstep_workspace_bytes(ctypes.byref(_lib.Shape(0, 1, 1, 32, 0)), 1, ctypes.byref(nbytes)) == -1
    # null arguments are rejected before anything touches the device
    assert lib.vmapstep_fwd_bwd(ctypes.byref(sh), None, None, None, 5.0, 10.0, None, None, None, 0, None) == -1
    with pytest.raises(_lib.VmapStepError):
        _lib.check(-1, lib)


def test_tuning_is_per_call_state_not_library_state():
    """The plan overrides travel in vmapstep_shape::tuning (since ABI v4); the library holds no tuning state, so sizing the
    workspace for one operator cannot change the plan of another."""
    lib = _lib.load()
    auto, tuned = ctypes.c_size_t(), ctypes.c_size_t()
    sh = _lib.Shape(20, 120, 10, 32, 0)
    assert lib.vmapstep_workspace_bytes(ctypes.byref(sh), 20, ctypes.byref(auto)) == 0
    t = _lib.Tuning(workgroups_per_object=2)
    sh2 = _lib.Shape(20, 120, 10, 32, 0)
    sh2.tuning = ctypes.pointer(t)
    assert lib.vmapstep_workspace_bytes(ctypes.byref(sh2), 20, ctypes.byref(tuned)) == 0
    assert tuned.value < auto.value                              # 2 instead of 10 gradient-partial rows per object
    again = ctypes.c_size_t()
    assert lib.vmapstep_workspace_bytes(ctypes.byref(sh), 20, ctypes.byref(again)) == 0
    assert again.value == auto.value                             # ... and nothing of it stuck to the library
    for k in (17, 7):                  # 7: a measurement prototype of ABI v4 (16-point forward tiles), no longer in the library
        bad = _lib.Tuning(kernel=k)
        sh2.tuning = ctypes.pointer(bad)
        assert lib.vmapstep_workspace_bytes(ctypes.byref(sh2), 20, ctypes.byref(tuned)) == -1
    assert not hasattr(lib, "vmapstep_set_workgroups_per_object")


def test_step_operator_refuses_cpu():
    from vmap_amd import step
    with pytest.raises(_lib.VmapStepError):
        step.VmapStep(2, 12, 10, 32, device="cpu")


def test_kernel_choice_per_width_shows_in_the_workspace_plan():
    """hidden 64 / 128 with at most 64 samples per ray run on the split-bf16 kernels (step_main_wp / step_main_ws): their
    workspace carries the W / W^T images and per-workgroup scratch, the exact-fp32 kernels' does not; VMAPSTEP_KERNEL_WS1 /
    _WP are refused where those kernels do not exist (hidden 32, hidden 256 with more than 32 samples per ray or as _WP, long
    rays); hidden 256 with short rays runs step_main_ws with eight waves while every tile gets a compute unit (round 3).
    Plan overrides are a matter of the measurement build (tests/tools/libvmapstep_ab.so: the same plan code + the A/B kernel forms)."""
    from conftest import AB_LIBRARY
    lib = _lib.load(AB_LIBRARY)

    def need(shape, kernel=None):
        n = ctypes.c_size_t()
        if kernel is not None:
            t = _lib.Tuning(kernel=kernel)
            shape.tuning = ctypes.pointer(t)
        rc = lib.vmapstep_workspace_bytes(ctypes.byref(shape), 20, ctypes.byref(n))
        return rc, n.value

    for H in (64, 128):
        rc_a, auto = need(_lib.Shape(1, 1200, 14, H, 0))
        rc_g, gen = need(_lib.Shape(1, 1200, 14, H, 0), _lib.KERNEL_GEN)
        rc_1, ws1 = need(_lib.Shape(1, 1200, 14, H, 0), _lib.KERNEL_WS1)
        rc_p, wp = need(_lib.Shape(1, 1200, 14, H, 0), _lib.KERNEL_WP)
        assert (rc_a, rc_g, rc_1, rc_p) == (0, 0, 0, 0)
        assert auto in (ws1, wp) and auto != gen
    for sh in (_lib.Shape(4, 120, 10, 32, 0), _lib.Shape(1, 100, 40, 256, 0), _lib.Shape(1, 8, 100, 128, 0)):
        assert need(sh, _lib.KERNEL_WS1)[0] != 0
        assert b"WS1" in lib.vmapstep_last_error()
    assert need(_lib.Shape(1, 100, 14, 256, 0), _lib.KERNEL_WP)[0] != 0
    # hidden 256 with short rays: the automatic plan = the eight-wave step_main_ws (one round per workgroup for 50 tiles, several for
    # the reference's 4800-ray iMAP batch); long rays: the exact-fp32 kernels
    for R in (100, 4800):
        rc_a, auto = need(_lib.Shape(1, R, 14, 256, 0))
        rc_1, ws1 = need(_lib.Shape(1, R, 14, 256, 0), _lib.KERNEL_WS1)
        rc_g, gen = need(_lib.Shape(1, R, 14, 256, 0), _lib.KERNEL_GEN)
        assert (rc_a, rc_1, rc_g) == (0, 0, 0) and auto == ws1 and auto != gen
    rc_a, auto = need(_lib.Shape(1, 100, 40, 256, 0))
    rc_g, gen = need(_lib.Shape(1, 100, 40, 256, 0), _lib.KERNEL_GEN)
    assert (rc_a, rc_g) == (0, 0) and auto == gen


def test_adamw_apply_checks_its_arguments_without_a_device():
    """vmapstep_adamw_apply (ABI v5: + reduced loss terms, step index, outputs): null / inconsistent arguments are refused
    before anything is enqueued."""
    lib = _lib.load()
    sh = _lib.Shape(1, 150, 14, 128, 0)
    assert lib.vmapstep_adamw_apply(ctypes.byref(sh), None, None, 0, None, None, 0, 5.0, 10.0, None, None, 0, None) == -1
    assert b"params" in lib.vmapstep_last_error()
    assert lib.vmapstep_adamw_apply(ctypes.byref(sh), None, None, 0, None, None, -1, 5.0, 10.0, None, None, 0, None) == -1
    assert b"step_index" in lib.vmapstep_last_error()


def test_launch_plans_of_the_baseline_shapes():
    """vmapstep_describe_plan (ABI v6): the plan rules pinned on the shapes the repository quotes - which fused-step kernel runs, how
    many rays a round takes, how many workgroups (= partial-gradient rows) an object gets, and whether every workgroup runs exactly
    one round (the specialised kernel forms).  No device needed."""
    P = _lib.describe_plan
    p = P(20, 120, 10, 32)                                   # BASELINE configs[1], the headline: 12 rays per workgroup, one pass
    assert (p["kernel"], p["rays_per_round"], p["workgroups_per_object"], p["single_round"]) == ("step_main_s32", 12, 10, 1)
    assert P(20, 120, 10, 32, tuning={"kernel": _lib.KERNEL_H32_F32})["kernel"] == "step_main_h32"
    p = P(50, 120, 10, 32, weights_bf16=True)                # configs[3]: five workgroups per object, two passes each
    assert (p["kernel"], p["workgroups_per_object"], p["rounds_per_object"], p["single_round"]) == ("step_main_s32", 5, 10, 0)
    p = P(256, 256, 10, 64, weights_bf16=True)               # configs[4]: two workgroups per object (two per compute unit), 21-22 rounds each
    assert (p["kernel"], p["rays_per_round"], p["workgroups_per_object"], p["rounds_per_object"], p["waves_per_workgroup"]) == ("step_main_wp<2>", 6, 2, 43, 4)
    # the background model (hidden 128, 14 samples): tiles per round by batch size
    p = P(1, 1200, 14, 128)                                  # one GPU: 300 two-tile rounds > 256 compute units -> 200 three-tile rounds
    assert (p["kernel"], p["tiles_per_round"], p["rays_per_round"], p["workgroups_per_object"], p["single_round"]) == ("step_main_ws<4>", 3, 6, 200, 1)
    p = P(1, 1200, 14, 128, tuning={"ws_flags": 4})          # the round-2 plan: 150 workgroups x two two-tile rounds
    assert (p["tiles_per_round"], p["rays_per_round"], p["workgroups_per_object"], p["rounds_per_object"], p["single_round"]) == (2, 4, 150, 300, 0)
    p = P(1, 600, 14, 128)                                   # 2 ranks: 150 two-tile rounds, one each
    assert (p["tiles_per_round"], p["workgroups_per_object"], p["single_round"]) == (2, 150, 1)
    for R, rounds in ((300, 150), (150, 75)):                # 4 / 8 ranks: every tile gets a compute unit -> single-tile rounds
        p = P(1, R, 14, 128)
        assert (p["tiles_per_round"], p["rays_per_round"], p["workgroups_per_object"], p["single_round"]) == (1, 2, rounds, 1)
    assert P(1, 150, 14, 128, tuning={"ws_flags": 1})["tiles_per_round"] == 2
    assert P(1, 150, 14, 128, tuning={"ws_flags": 2})["tiles_per_round"] == 3
    # hidden 256 (the iMAP field): eight waves, single-tile rounds; long rays -> the exact-fp32 kernels
    p = P(1, 100, 14, 256)
    assert (p["kernel"], p["tiles_per_round"], p["waves_per_workgroup"], p["workgroups_per_object"], p["single_round"]) == ("step_main_ws<8>", 1, 8, 50, 1)
    p = P(1, 4800, 14, 256)
    assert (p["kernel"], p["workgroups_per_object"], p["rounds_per_object"], p["single_round"]) == ("step_main_ws<8>", 240, 2400, 0)   # ten rounds each: evenly spread
    assert P(1, 100, 40, 256)["kernel"] in ("step_main_wide<4>", "step_main_gen")
    assert P(1, 100, 14, 96)["kernel"] == "step_main_gen"
    with pytest.raises(_lib.VmapStepError):
        P(1, 100, 14, 48)


def test_product_library_carries_the_planned_forms_only():
    """Round 4: the product library ships the kernel forms AUTOMATIC plans launch (+ the exact-fp32 training references); the A/B
    forms - step_main_ws at hidden 64, step_main_wp at hidden 128, step_main_wide<4>, three-tile rounds with several rounds per
    workgroup - are refused by its plan with a message naming the measurement build, which accepts them; both builds make the same
    plan for every automatic shape."""
    from conftest import AB_LIBRARY
    prod, ab = _lib.load(), _lib.load(AB_LIBRARY)

    def plan(lib, shape, **tun):
        n = ctypes.c_size_t()
        if tun:
            t = _lib.Tuning(**tun)
            shape.tuning = ctypes.pointer(t)
        return lib.vmapstep_workspace_bytes(ctypes.byref(shape), 20, ctypes.byref(n)), n.value

    ab_only = [(_lib.Shape(1, 1200, 14, 64, 0), dict(kernel=_lib.KERNEL_WS1)), (_lib.Shape(1, 1200, 14, 128, 0), dict(kernel=_lib.KERNEL_WP)),
               (_lib.Shape(1, 100, 14, 256, 0), dict(kernel=_lib.KERNEL_WIDE4)), (_lib.Shape(1, 1200, 14, 128, 0), dict(ws_flags=2, workgroups_per_object=50))]
    for sh, tun in ab_only:
        assert plan(prod, sh, **tun)[0] == -2 and b"measurement build" in prod.vmapstep_last_error(), tun
        assert plan(ab, sh, **tun)[0] == 0, tun
    for shape in ((20, 120, 10, 32), (50, 120, 10, 32), (256, 256, 10, 64), (32, 256, 10, 64), (1, 1200, 14, 128), (1, 600, 14, 128), (1, 150, 14, 128),
                  (2, 1200, 14, 128), (1, 100, 14, 256), (1, 4800, 14, 256), (1, 100, 40, 256), (3, 50, 12, 96)):
        for wd in (0, 1):
            a, b = plan(prod, _lib.Shape(*shape, wd)), plan(ab, _lib.Shape(*shape, wd))
            assert a[0] == 0 and a == b, shape
    # explicit overrides that select forms the product does carry stay available on it
    for sh, tun in [(_lib.Shape(20, 120, 10, 32, 0), dict(kernel=_lib.KERNEL_H32_F32)), (_lib.Shape(1, 1200, 14, 128, 0), dict(kernel=_lib.KERNEL_GEN)),
                    (_lib.Shape(1, 1200, 14, 128, 0), dict(ws_flags=4)), (_lib.Shape(1, 150, 14, 128, 0), dict(ws_flags=1))]:
        assert plan(prod, sh, **tun)[0] == 0, tun
