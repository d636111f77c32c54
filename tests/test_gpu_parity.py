"""GPU tier (`-m gpu`): the HIP path, called through the C ABI, against the reference fixtures and the oracle.

Tolerances: max-norm relative error (max|a-b| / max|b|) 1e-4 on rendered outputs and on every gradient tensor
(north_star: "within 1e-4 rel fp32"); the 'saturated' case (occupancy == 1.0f, var -> 0) is bounded by the
reference's own float32 noise floor measured between two CPU implementations (tests/test_oracle_vs_golden.py).
"""
import ctypes

import numpy as np
import pytest
import torch

import cases
from conftest import tensor_err_q, kink_aware, GRAD_KEYS, RENDER_KEYS, load_golden, relerr, make_op, TEST_TUNING
from oracle import vmap_oracle as vo
from vmap_amd import _lib, layout, step, synth

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = {"default": (2e-5, 1e-4), "saturated": (2e-3, 2e-3), "explode": (2e-3, 2e-3)}       # 'explode' is a saturated case as well (gain 4.0)


def _to_dev(c):
    fc = [torch.from_numpy(a).to(DEV) for a in c["fc"]]
    B = torch.from_numpy(c["B"]).to(DEV)
    sc = torch.from_numpy(c["scale"]).to(DEV)
    b = {k: torch.from_numpy(v).to(DEV) for k, v in c["batch"].items()}
    return fc, B, sc, b


def _run(c, fn="fwd_bwd", op=None, tuning=None, **kw):
    fc, B, sc, b = _to_dev(c)
    op = op or make_op(c["n"], c["R"], c["S"], c["H"], device=DEV, tuning=tuning)
    gfc = [torch.full_like(t, float("nan")) for t in fc]
    gB = torch.full_like(B, float("nan"))
    if fn == "fwd_bwd":
        res = op.fwd_bwd(fc, B, sc, b["pcs"], b["z"], b["gt_depth"], b["gt_rgb"], b["sem"], b["depth_mask"],
                         grads_fc=gfc, grad_B=gB, render=True, **kw)
    else:
        try:
            res = op.render(fc, B, sc, b["pcs"], b["z"], b["gt_depth"], b["gt_rgb"], b["sem"], b["depth_mask"])
        except _lib.VmapStepError as e:
            # the forward-only instantiation of the exact-fp32 kernel is not in the product (its plan is accepted, the launch refuses):
            # the "f32" leg renders on the measurement build
            if "measurement build only" not in str(e):
                raise
            from conftest import AB_LIBRARY
            op = step.VmapStep(c["n"], c["R"], c["S"], c["H"], device=DEV, tuning={**(TEST_TUNING["default"] or {}), **(tuning or {})}, library=AB_LIBRARY)
            res = op.render(fc, B, sc, b["pcs"], b["z"], b["gt_depth"], b["gt_rgb"], b["sem"], b["depth_mask"])
    torch.cuda.synchronize()
    out = dict(loss=float(res.loss[0]), flags=res.flags[0].cpu().numpy(),
               render_depth=res.render_depth.cpu().numpy(), render_color=res.render_color.cpu().numpy(),
               opacity=res.opacity.cpu().numpy(), var=res.var.cpu().numpy())
    for t in range(14):
        out[f"g_fc{t}"] = gfc[t].cpu().numpy()
    out["g_B"] = gB.cpu().numpy()
    return out


H32_CASES = [n for n, v in cases.CASES.items() if v[3] == 32]


def _var_tol(g, base=2e-5):
    """Rendered variance (loss.py:28-29; an output of the boundary) against the reference fixture: 2e-5 of its
    max like the other renders, widened only where the reference's OWN float32 and float64 runs drift further apart (the
    saturated case: occupancy == 1.0f makes var a difference of rounding errors, 8e-4 between the two precisions)."""
    return max(base, 3.0 * relerr(g["var"], g["f64_var"]))


@pytest.fixture(autouse=True, params=["split", "f32"])
def h32_kernel(request):
    """Every test of this module runs twice: hidden 32 on the default split-bf16 kernel (step_main_s32) and on the
    exact-fp32 kernel (step_main_h32, tuning.kernel = KERNEL_H32_F32).  Other widths ignore the choice."""
    old = TEST_TUNING["default"]
    TEST_TUNING["default"] = {"kernel": _lib.KERNEL_H32_F32} if request.param == "f32" else None
    yield request.param
    TEST_TUNING["default"] = old


def test_native_library_is_loaded():
    lib = _lib.load()
    assert lib.vmapstep_abi_version() == _lib.ABI_VERSION == 7
    assert torch.cuda.is_available()
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


@pytest.mark.parametrize("name", H32_CASES)
def test_fwd_bwd_matches_reference_fixture(name):
    c = cases.build_case(name)
    g = load_golden(name)
    s = _run(c)
    rt, gt = TOL.get(name, TOL["default"])
    assert abs(s["loss"] - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for k in RENDER_KEYS:
        assert relerr(s[k], g[k]) < rt, k
    assert relerr(s["var"], g["var"]) < _var_tol(g, rt)
    for k in GRAD_KEYS:
        assert not np.isnan(s[k]).any(), k
        assert relerr(s[k], g[k]) < gt, k
        # second, per-tensor criterion: 99.9 % of the elements within 5e-3 of |ref| + 1e-3 max|ref| (i.e. 5e-6 of the tensor's max for its smallest entries) (small-magnitude
        # entries are constrained too; the saturated case keeps its documented noise floor)
        # (explode: saturated AND subnormal colour-head gradients: the reference's own float32 and float64 runs differ by 2.5e-2 under
        # this criterion, the numpy oracle and the reference by 4.0e-2 - measured, tests/test_oracle_vs_golden.py's floor)
        assert tensor_err_q(s[k], g[k]) < {"saturated": 2e-2, "explode": 5e-2}.get(name, 5e-3), k
    o = vo.training_step(c["fc"], c["B"], c["scale"], c["batch"], dtype=np.float32)
    assert s["flags"][:3].tolist() == [int(x) for x in o["drop"]]
    assert int(s["flags"][3]) == int(o["explode"])


@pytest.mark.parametrize("n,R,S,seed", [(1, 1, 10, 1), (1, 120, 10, 2), (7, 33, 10, 3), (3, 9, 14, 4), (2, 40, 3, 5),
                                        (31, 13, 10, 6), (2, 5, 32, 7)])
def test_fwd_bwd_matches_oracle_on_seeded_shapes(n, R, S, seed):
    """Ragged ray counts, single ray / single object, S = 3 .. 32, more objects than fit one pass."""
    fc, B, sc = synth.make_params(n, 32, seed=100 + seed)
    batch = synth.make_batch(n, R, S, seed=200 + seed)
    c = dict(n=n, R=R, S=S, H=32, fc=fc, B=B, scale=sc, batch=batch)
    s = _run(c)
    # primary comparator: the PyTorch-CPU port (same ATen kernels the reference runs, FMA-based like the MFMA
    # chain); the numpy oracle may sit on the other side of a ReLU kink for an isolated hidden unit, which moves
    # one gradient tensor by ~1e-4..1e-2 of its max - bounded separately.
    from oracle import vmap_oracle_torch as vt
    loss_t, rend_t, grads_t = vt.CpuTrainer(fc, B, sc).step(batch, update=False)
    assert abs(s["loss"] - float(loss_t)) <= 5e-5 * abs(float(loss_t))
    for k in RENDER_KEYS:
        assert relerr(s[k], rend_t[k].detach().numpy()) < 2e-5, k
    for k, g in zip(GRAD_KEYS, grads_t):
        assert relerr(s[k], g.numpy()) < 1e-4, k
    o = vo.training_step(fc, B, sc, batch, dtype=np.float32, kinks=True)
    for k in RENDER_KEYS + ["var"]:
        assert relerr(s[k], o[k]) < 2e-5, k


def _assert_grads_match_aten_port_up_to_kinks(s, o, grads_t, n_obj, tol=1e-4):
    """Gradients against the ATen PORT (the third-party kernels the reference runs; it reproduces the reference's frame trajectories
    bit for bit) at north_star's 1e-4, ReLU kinks accounted for: the candidate list and the exact effect of each flip come from the
    numpy oracle (``o`` computed with kinks=True); a bit may differ from the oracle's state in the port, in the kernel or in both,
    hence SIGNED coefficients in {-1, 0, +1} (conftest.kink_aware)."""
    ref = dict(o)
    for k, g in zip(GRAD_KEYS, grads_t):
        ref[k] = g.numpy() if hasattr(g, "numpy") else np.asarray(g)
    corr, flipped, cand, worst = kink_aware(s, ref, n_obj, signed=True, tol=tol)
    for k in GRAD_KEYS:
        assert not np.isnan(s[k]).any(), k
        assert relerr(s[k], corr[k]) < tol, (k, flipped, cand)


def _assert_grads_match_oracle_up_to_kinks(s, o, n_obj, tol=1e-4, signed=False):
    """Gradients against a REFERENCE FIXTURE (``o``: the fixture's gradients + the numpy oracle's kink list, ``signed``) at north_star's
    1e-4 with the ReLU kinks ACCOUNTED FOR instead of tolerated: the oracle lists every hidden unit whose pre-activation lies inside
    float32 forward rounding of 0 and the exact gradient change of flipping its derivative bit; the kernel's gradients must equal the
    reference's plus a -1/0/+1 combination of those changes (conftest.kink_aware).  Round 5: the 2e-4 comparisons against the numpy
    oracle's OWN gradients are gone - wherever they ran, the ATen port (the third-party kernels the reference itself runs) is the
    comparator at 1e-4 (_assert_grads_match_aten_port_up_to_kinks)."""
    corr, flipped, cand, worst = kink_aware(s, o, n_obj, signed=signed, tol=tol)
    for k in GRAD_KEYS:
        assert not np.isnan(s[k]).any(), k
        assert relerr(s[k], corr[k]) < tol, (k, flipped, cand)
    return flipped


@pytest.mark.parametrize("H", [128, 64])
@pytest.mark.parametrize("n,R,S,seed", [(1, 1, 14, 11), (3, 9, 14, 12), (2, 37, 10, 13), (1, 21, 3, 14), (2, 7, 20, 15),
                                        (1, 5, 40, 16), (1, 3, 64, 17), (5, 300, 14, 18)])
def test_hidden128_kernel_matches_oracle_on_seeded_shapes(n, R, S, seed, H):
    """step_main_wp (hidden 128 and 64, the automatic choice) and step_main_ws on ragged shapes: single ray, rays that straddle the two tiles of a
    round (S = 10, 14, 20), one ray per round (S = 40, 64 = the kernel's limit), long rays through the general compositing
    path (S > 16), several objects, more rounds than workgroups (5 x 75 rounds)."""
    fc, B, sc = synth.make_params(n, H, seed=300 + seed)
    batch = synth.make_batch(n, R, S, seed=400 + seed)
    c = dict(n=n, R=R, S=S, H=H, fc=fc, B=B, scale=sc, batch=batch)
    s = _run(c, tuning={"kernel": _lib.KERNEL_WP})
    from oracle import vmap_oracle_torch as vt
    loss_t, rend_t, grads_t = vt.CpuTrainer(fc, B, sc).step(batch, update=False)
    assert abs(s["loss"] - float(loss_t)) <= 5e-5 * abs(float(loss_t))
    for k in RENDER_KEYS:
        assert relerr(s[k], rend_t[k].detach().numpy()) < 2e-5, k
    # gradients: the numpy oracle with its kink-adjacent hidden units accounted for bit by bit, at 1e-4 (this replaces the
    # former "up to 5 of 15 tensors may sit at 2e-2 of the ATen port" rule)
    o = vo.training_step(fc, B, sc, batch, dtype=np.float32, kinks=True)
    for k in RENDER_KEYS + ["var"]:
        assert relerr(s[k], o[k]) < 2e-5, k
    _assert_grads_match_aten_port_up_to_kinks(s, o, grads_t, n)      # round 4: the ATen port at 1e-4, signed kink coefficients
    a = _run(c)                                                       # the automatic plan (product library)
    _assert_grads_match_aten_port_up_to_kinks(a, o, grads_t, n)
    e = _run(c, tuning={"kernel": _lib.KERNEL_GEN})
    w1 = _run(c, tuning={"kernel": _lib.KERNEL_WS1})              # step_main_ws: one wave per output block (single-tile rounds where they fit)
    w2 = _run(c, tuning={"kernel": _lib.KERNEL_WS1, "ws_flags": 1})    # ... two-tile rounds
    _assert_grads_match_aten_port_up_to_kinks(w1, o, grads_t, n)
    _assert_grads_match_aten_port_up_to_kinks(w2, o, grads_t, n)
    if H == 128:
        w3 = _run(c, tuning={"kernel": _lib.KERNEL_WS1, "ws_flags": 2})    # ... three-tile rounds (hidden 128)
        _assert_grads_match_aten_port_up_to_kinks(w3, o, grads_t, n)
        for k in RENDER_KEYS:
            assert relerr(w3[k], e[k]) < 2e-5, k
        for k in GRAD_KEYS:
            assert relerr(w3[k], e[k]) < 1e-4, k
    for k in RENDER_KEYS:
        assert relerr(s[k], e[k]) < 2e-5, k
        assert relerr(w1[k], e[k]) < 2e-5, k
        assert relerr(w2[k], e[k]) < 2e-5, k
    for k in GRAD_KEYS:
        assert relerr(s[k], e[k]) < 1e-4, k
        assert relerr(w1[k], e[k]) < 1e-4, k
        assert relerr(w2[k], e[k]) < 1e-4, k


@pytest.mark.parametrize("n,R,S,seed", [(1, 1, 14, 21), (3, 9, 14, 22), (2, 37, 10, 23), (1, 21, 3, 24), (2, 7, 20, 25), (1, 5, 32, 26), (1, 100, 14, 27),
                                        (1, 600, 14, 28)])
def test_hidden256_kernel_matches_oracle_on_seeded_shapes(n, R, S, seed):
    """step_main_ws<8> (hidden 256, eight waves, single-tile rounds: the automatic choice for rays of at most 32 samples) on
    ragged shapes - single ray, partly filled tiles, one ray per round (S = 20, 32 = its limit), long rays through the general
    compositing path, several objects, more rounds than compute units (600 rays = 300 rounds: 150 workgroups x 2, the
    multi-round form) - against the oracle with its ReLU kinks accounted for, the exact-fp32 general kernel, and itself with
    several rounds per workgroup."""
    H = 256
    fc, B, sc = synth.make_params(n, H, seed=500 + seed)
    batch = synth.make_batch(n, R, S, seed=600 + seed)
    c = dict(n=n, R=R, S=S, H=H, fc=fc, B=B, scale=sc, batch=batch)
    s = _run(c)                                                          # automatic plan
    o = vo.training_step(fc, B, sc, batch, dtype=np.float32, kinks=True)
    assert abs(s["loss"] - o["loss"]) <= 5e-5 * abs(o["loss"])
    for k in RENDER_KEYS + ["var"]:
        assert relerr(s[k], o[k]) < 2e-5, k
    from oracle import vmap_oracle_torch as vt
    loss_t, rend_t, grads_t = vt.CpuTrainer(fc, B, sc).step(batch, update=False)
    assert abs(s["loss"] - float(loss_t)) <= 5e-5 * abs(float(loss_t))
    _assert_grads_match_aten_port_up_to_kinks(s, o, grads_t, n)      # round 4: the ATen port at 1e-4, signed kink coefficients
    e = _run(c, tuning={"kernel": _lib.KERNEL_GEN})
    m = _run(c, tuning={"kernel": _lib.KERNEL_WS1, "workgroups_per_object": 2})   # several rounds per workgroup
    _assert_grads_match_aten_port_up_to_kinks(m, o, grads_t, n)
    for k in RENDER_KEYS:
        assert relerr(s[k], e[k]) < 2e-5, k
        assert relerr(m[k], e[k]) < 2e-5, k
    for k in GRAD_KEYS:
        assert relerr(s[k], e[k]) < 1e-4, k
        assert relerr(m[k], e[k]) < 1e-4, k
    s2 = _run(c)
    for k in GRAD_KEYS:
        assert np.array_equal(s[k], s2[k]), k                                     # bit-repeatable


def test_explode_flag_through_every_entry_point():
    """render_rays.py:88-90 with explode = 1 (the reference calls exit(-1) on these inputs: fixture 'explode', exit_code -1): the
    device flag VMAPSTEP_FLAG_EXPLODE comes up - and the values the reference had computed when it gave up are reproduced - through
    vmapstep_fwd_bwd, vmapstep_render, vmapstep_train_steps (step 0 of a frame; driver.HipMapper.check_flags turns it into an
    exception) and the global-loss path of vmapstep_adamw_apply (the ray-sharded caller's rank-summed loss terms)."""
    from vmap_amd import driver
    c = cases.build_case("explode")
    g = load_golden("explode")
    assert int(g["exit_code"]) == -1
    n, R, S, H = c["n"], c["R"], c["S"], c["H"]
    # vmapstep_fwd_bwd / vmapstep_render
    for fn in ("fwd_bwd", "render"):
        s = _run(c, fn=fn)
        assert s["flags"].tolist() == [0, 0, 0, 1], (fn, s["flags"])
        assert abs(s["loss"] - float(g["loss"])) <= 2e-5 * float(g["loss"]), fn
    # the same inputs with the far depths back in range: the flag stays down (it is the inputs, not the entry point)
    c0 = cases.build_case("explode")
    c0["batch"]["gt_depth"][1, :] = np.where(c0["batch"]["gt_depth"][1, :] > 0, np.float32(1.5), 0).astype(np.float32)
    assert _run(c0)["flags"].tolist() == [0, 0, 0, 0]
    # vmapstep_train_steps: a two-step frame whose FIRST step is the explode batch and whose second is the in-range one
    fc, B, sc, b = _to_dev(c)
    _, _, _, b0 = _to_dev(c0)
    frame = {k: torch.cat([b[k], b0[k]], dim=1).contiguous() for k in b}
    op = make_op(n, R, S, H, device=DEV, max_steps=2)
    opt = step.FusedAdamWState(n, H, DEV, lr=1e-3, weight_decay=0.013)
    res = op.train_steps(fc, B, sc, frame["pcs"], frame["z"], frame["gt_depth"], frame["gt_rgb"], frame["sem"], frame["depth_mask"],
                         opt=opt, n_steps=2)
    torch.cuda.synchronize()
    fl = res.flags.cpu().numpy()
    assert fl[0].tolist() == [0, 0, 0, 1] and fl[1].tolist() == [0, 0, 0, 0], fl
    assert abs(float(res.loss[0]) - float(g["loss"])) <= 2e-5 * float(g["loss"])
    with pytest.raises(RuntimeError, match="loss explode"):
        driver.HipMapper.check_flags(None, res)
    # vmapstep_adamw_apply, global-loss path: prepared step -> per-object loss terms + gradient slab -> the optimiser launch
    # raises the flag from the (here: single-rank) summed terms
    fc, B, sc, b = _to_dev(c)
    P, PP = layout.param_count(H), opt.padded
    gslab = torch.zeros(n, PP, device=DEV)
    offs = layout.flat_offsets(H)
    shapes = list(layout.fc_shapes(H)) + [layout.PE_B_SHAPE]
    gviews = [gslab[:, offs[t]:offs[t] + layout.numel(shp)].view((n,) + tuple(shp)) for t, shp in enumerate(shapes)]
    op = make_op(n, R, S, H, device=DEV, max_steps=1)
    opt = step.FusedAdamWState(n, H, DEV, lr=1e-3, weight_decay=0.013)
    op.prepare_frame(fc, B, b["pcs"], b["z"], b["gt_depth"], b["gt_rgb"], b["sem"], b["depth_mask"], n_steps=1)
    terms = torch.zeros(n, 4, device=DEV)
    r1 = op.fwd_bwd(fc, B, sc, b["pcs"], b["z"], b["gt_depth"], b["gt_rgb"], b["sem"], b["depth_mask"], grads_fc=gviews[:14],
                    grad_B=gviews[14], prepared_step=0, loss_terms=terms)
    loss_out = torch.zeros(1, device=DEV)
    flags_out = torch.zeros(4, dtype=torch.int32, device=DEV)
    op.adamw_apply(fc, B, gslab, opt, loss_terms=terms, step_index=0, loss_out=loss_out, flags_out=flags_out)
    torch.cuda.synchronize()
    assert r1.flags[0].cpu().tolist() == [0, 0, 0, 1]
    assert flags_out.cpu().tolist() == [0, 0, 0, 1]
    assert abs(float(loss_out[0]) - float(g["loss"])) <= 2e-5 * float(g["loss"])
    assert float(terms[1, 0]) > 1e5 and float(terms[:, 0].max()) == float(terms[1, 0])      # object 1's depth term is the one that fired


def test_reference_regenerated_on_this_box_equals_the_committed_fixture():
    """The reference ITSELF (oracle/_ref: model / embedding / render_rays / loss byte-compiled unmodified, oracle/make_ref.py; or the
    source tree where it exists), run on THIS box's CPU through functorch exactly like tests/golden/make_goldens.py, against the
    committed 'tiny' and 'explode' fixtures: guards the fixtures against drifting from a future torch / another host CPU without
    anyone noticing (torch_version is asserted; the values must agree to float32 rounding, bit-equality is reported)."""
    from oracle import ref_runner
    if not ref_runner.reference_available():
        pytest.skip("neither /root/reference nor oracle/_ref (python oracle/make_ref.py) on this box")
    g = load_golden("tiny")
    assert str(g["torch_version"]) == torch.__version__, "fixtures were generated with another torch: regenerate them (tests/golden/make_*.py)"
    c = cases.build_case("tiny")
    r = ref_runner.reference_step(c["fc"], c["B"], c["scale"], c["batch"], c["H"], torch.float32)
    keys = RENDER_KEYS + ["var"] + GRAD_KEYS
    bit_equal = all(np.array_equal(np.asarray(r[k], np.float32), g[k]) for k in keys) and float(r["loss"]) == float(g["loss"])
    print(f"reference ({ref_runner.SOURCE}) regenerated 'tiny': bit-identical to the committed fixture = {bit_equal}")
    assert abs(float(r["loss"]) - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    for k in keys:
        assert relerr(r[k], g[k]) < 5e-6, k
    # and the HIP path against what was just regenerated (not only against the stored copy)
    s = _run(c)
    for k in RENDER_KEYS:
        assert relerr(s[k], r[k]) < 2e-5, k
    for k in GRAD_KEYS:
        assert relerr(s[k], r[k]) < 1e-4, k
    # the reference's exit(-1) on the explode inputs happens here as well
    ce = cases.build_case("explode")
    with pytest.raises(SystemExit):
        ref_runner.reference_step(ce["fc"], ce["B"], ce["scale"], ce["batch"], ce["H"], torch.float32)


def _ray_case(n, R, S, H, seed, steps=1):
    """A synthetic frame given as rays: origins / directions / centres drawn like vmap.py:31-41 would produce them, z from the usual
    generator; the points tensor they stand for is formed with eager torch ops (one rounding per operation, like the kernels)."""
    fc, B, sc = synth.make_params(n, H, seed=seed)
    frame = synth.make_batch(n, R * steps, S, seed=seed + 1)
    rng = np.random.default_rng(seed + 2)
    o = rng.uniform(-1.0, 1.0, (n, R * steps, 3)).astype(np.float32)
    d = np.concatenate([rng.uniform(-1.0, 1.0, (n, R * steps, 1)), rng.uniform(-0.57, 0.57, (n, R * steps, 1)), np.ones((n, R * steps, 1))], -1).astype(np.float32)
    cen = rng.uniform(-0.5, 0.5, (n, 3)).astype(np.float32)
    rays = step.RayPoints(torch.from_numpy(o).to(DEV), torch.from_numpy(d).to(DEV), torch.from_numpy(cen).to(DEV))
    fr = {k: torch.from_numpy(v).to(DEV) for k, v in frame.items()}
    fr["pcs"] = rays.points(fr["z"]).contiguous()
    return fc, B, sc, fr, rays


@pytest.mark.parametrize("H,n,R,S,tuning", [(32, 5, 37, 10, None), (32, 40, 24, 10, None), (64, 3, 33, 10, None), (128, 2, 21, 14, None),
                                            (256, 1, 10, 14, None), (96, 2, 17, 10, None), (128, 1, 9, 14, {"kernel": 2})])
def test_ray_handoff_is_bit_identical_to_points(H, n, R, S, tuning):
    """ABI v7 (second half; vmap.py:452-457): the step given (origin, direction, z, centre) per ray instead of the
    points tensor - every kernel family (step_main_s32 single- and multi-pass / step_main_h32 through the module's legs, _wp, _ws, _ws<8>,
    _gen, _wide) rebuilds the points in its prologue with the sampler's own arithmetic, so loss, renders, all 15 gradient tensors and a
    3-step training trajectory (strided ray slices, fused AdamW) are BIT-identical to the run on the points tensor."""
    if H != 32 and TEST_TUNING["default"] is not None:
        pytest.skip("the module's hidden-32 kernel legs do not apply; run once")
    steps = 3
    fc, B, sc, fr, rays = _ray_case(n, R, S, H, seed=900 + H, steps=steps)
    keys = ("z", "gt_depth", "gt_rgb", "sem", "depth_mask")
    outs = []
    for pts in (fr["pcs"], rays):
        tfc = [torch.from_numpy(a).to(DEV) for a in fc]
        tB, tsc = torch.from_numpy(B).to(DEV), torch.from_numpy(sc).to(DEV)
        op = make_op(n, R, S, H, device=DEV, max_steps=steps, tuning=tuning)
        gfc, gB = [torch.zeros_like(t) for t in tfc], torch.zeros_like(tB)
        sl = slice(R, 2 * R)                                         # a strided slice of the frame, like train.py:271-277
        res = op.fwd_bwd(tfc, tB, tsc, pts[:, sl], *(fr[k][:, sl] for k in keys), grads_fc=gfc, grad_B=gB, render=True)
        one = [res.loss.clone(), res.render_depth, res.render_color, res.opacity, res.var] + gfc + [gB]
        st = step.FusedAdamWState(n, H, DEV)
        r2 = op.train_steps(tfc, tB, tsc, pts, *(fr[k] for k in keys), opt=st, n_steps=steps, ray_step=R)
        torch.cuda.synchronize()
        outs.append([x.cpu() for x in one + [r2.loss] + tfc + [tB]])
    assert float(outs[0][0][0]) != 0.0 and bool(torch.isfinite(outs[0][0]).all())
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)
    with pytest.raises(_lib.VmapStepError, match="neither pcs nor"):
        b = op._batch(fr["pcs"], *(fr[k] for k in keys), rays_total=R * steps)
        b.pcs = None
        res, out = op._outputs(1, False)
        _lib.check(op.lib.vmapstep_render(ctypes.byref(op.shape), ctypes.byref(op._params(tfc, tB)), ctypes.byref(_lib.Tensor(tsc.data_ptr(), 1)),
                                          ctypes.byref(b), 5.0, 10.0, ctypes.byref(out), op._ws_ptr, op._ws_bytes, op._stream()), op.lib)


def test_background_classes_take_the_frame_as_rays():
    """parallel.SharedBackgroundHip (prepare / per-step launches + the optimiser launch) and parallel.OwnerBackgroundHip (bound frame call)
    on a one-object frame handed over WITHOUT the object dimension as step.RayPoints (origins / dirs [R_total, 3], centre [3]): the same
    per-step losses and the same final parameter slab, bit for bit, as on the points tensor."""
    if TEST_TUNING["default"] is not None:
        pytest.skip("hidden 128: the module's hidden-32 kernel legs do not apply; run once")
    from vmap_amd import fields, parallel
    R, S, steps, H = 60, 14, 3, 128
    fc, B, sc, fr, rays = _ray_case(1, R, S, H, seed=1500, steps=steps)
    flat = {k: v[0] for k, v in fr.items()}
    ray1 = step.RayPoints(rays.origins[0], rays.dirs[0], rays.centers[0])
    outs = []
    for cls in (parallel.SharedBackgroundHip, parallel.OwnerBackgroundHip):
        for pts in (flat["pcs"], ray1):
            torch.manual_seed(3)
            m = fields.OccupancyMap(hidden_size=H)
            m.apply(fields.init_weights)
            pe = fields.UniDirsEmbed(max_deg=5, scale=5.0)
            bg = cls(m, pe, R, S, DEV, max_steps=steps)
            args = (pts, flat["z"], flat["gt_depth"], flat["gt_rgb"], flat["sem"], flat["depth_mask"])
            if cls is parallel.SharedBackgroundHip:
                losses = bg.train_frame(*args, n_steps=steps).clone()
            else:
                losses = bg.train_frame(*args, n_steps=steps).loss[:steps].clone()
            torch.cuda.synchronize()
            outs.append((losses.cpu(), bg.slab.cpu().clone()))
    for a_, b_ in ((outs[0], outs[1]), (outs[2], outs[3])):
        assert bool(torch.isfinite(a_[0]).all()) and torch.equal(a_[0], b_[0]) and torch.equal(a_[1], b_[1])


@pytest.mark.parametrize("H,n,R,S,weights", [(64, 32, 256, 10, "f32"), (64, 32, 256, 10, "bf16"), (256, 1, 4800, 14, "f32"), (32, 50, 120, 10, "f32")])
def test_co_resident_waves_stay_bit_repeatable(H, n, R, S, weights):
    """Thirty launches of the forms that put TWO waves on a SIMD (step_main_wp<2>: two workgroups per CU at 480 workgroups; step_main_ws<8>:
    eight waves per workgroup) - and the multi-pass hidden-32 form for reference - on the points tensor and on the ray hand-off: loss,
    renders and every gradient tensor are bit-identical run to run.  (Round 5: an unrelated edit of step_main_wp's prologue - the ray
    hand-off's extra branch - produced a code object whose results differed run to run by 1e-6 .. 1e-3 exactly when two of its
    workgroups shared a CU, and only then; a three-launch repeat check did not always see it.  HISTORY.md, round 5.)"""
    if H != 32 and TEST_TUNING["default"] is not None:
        pytest.skip("the module's hidden-32 kernel legs do not apply; run once")
    fc, B, sc, fr, rays = _ray_case(n, R, S, H, seed=1200 + H)
    keys = ("z", "gt_depth", "gt_rgb", "sem", "depth_mask")
    tfc = [torch.from_numpy(a).to(DEV) for a in fc]
    tB, tsc = torch.from_numpy(B).to(DEV), torch.from_numpy(sc).to(DEV)
    op = make_op(n, R, S, H, device=DEV, max_steps=1, weights=weights)
    first = None
    for rep in range(30):
        pts = rays if rep % 2 else fr["pcs"]
        gfc, gB = [torch.zeros_like(t) for t in tfc], torch.zeros_like(tB)
        res = op.fwd_bwd(tfc, tB, tsc, pts, *(fr[k] for k in keys), grads_fc=gfc, grad_B=gB, render=True)
        torch.cuda.synchronize()
        out = [res.loss.cpu(), res.render_depth.cpu(), res.render_color.cpu(), res.opacity.cpu()] + [g.cpu() for g in gfc] + [gB.cpu()]
        if first is None:
            first = out
        for i, (x, y) in enumerate(zip(out, first)):
            assert torch.equal(x, y), (rep, i)


def test_loss_block_reads_its_partials_from_memory_when_they_do_not_fit_the_staging_area():
    """The finalize kernels' loss workgroup copies the n_obj x NW loss partials to LDS before one thread per object sums them in order
    (FinalizeArgs::loss_stage); with more partials than the launch's LDS has room for (1024 at hidden 32) it reads them from memory as
    it did before round 5.  600 objects x 2 workgroups: loss, per-object terms and gradients against the ATen port."""
    n, R, S, H = 600, 24, 10, 32
    fc, B, sc = synth.make_params(n, H, seed=2300)
    batch = synth.make_batch(n, R, S, seed=2301)
    c = dict(n=n, R=R, S=S, H=H, fc=fc, B=B, scale=sc, batch=batch)
    op = make_op(n, R, S, H, device=DEV, tuning={"workgroups_per_object": 2})
    assert op.plan()["workgroups_per_object"] == 2
    s = _run(c, op=op)
    from oracle import vmap_oracle_torch as vt
    loss_t, rend_t, grads_t = vt.CpuTrainer(fc, B, sc).step(batch, update=False)
    assert abs(s["loss"] - float(loss_t)) <= 2e-5 * abs(float(loss_t))
    for k in RENDER_KEYS:
        assert relerr(s[k], rend_t[k].detach().numpy()) < 2e-5, k
    if max(relerr(s[k], g.numpy()) for k, g in zip(GRAD_KEYS, grads_t)) >= 1e-4:
        o = vo.training_step(fc, B, sc, batch, dtype=np.float32, kinks=True)
        _assert_grads_match_aten_port_up_to_kinks(s, o, grads_t, n)


@pytest.mark.parametrize("H", [32, 64, 128, 256, 96])
def test_far_point_takes_the_library_sincos_path_on_the_device(H):
    """One sample point 3e5 units away: 32 pi |proj| exceeds the fast sincos' range (2^20), so its whole wave takes the library sincosf for
    octave 0 (a path ordinary scenes never reach; on the simulator it runs the HOST libm, here the device's) - loss, renders and all 15
    gradient tensors against the ATen port at the usual bars, every kernel family."""
    if H != 32 and TEST_TUNING["default"] is not None:
        pytest.skip("the module's hidden-32 kernel legs do not apply; run once")
    n, R, S = (3, 12, 10) if H != 256 else (1, 4, 14)
    fc, B, sc = synth.make_params(n, H, seed=1700 + H)
    batch = synth.make_batch(n, R, S, seed=1701 + H)
    batch["pcs"][n - 1, 3, 4, :] = [3.0e5, -2.0e5, 1.0e5]
    c = dict(n=n, R=R, S=S, H=H, fc=fc, B=B, scale=sc, batch=batch)
    s = _run(c)
    from oracle import vmap_oracle_torch as vt
    loss_t, rend_t, grads_t = vt.CpuTrainer(fc, B, sc).step(batch, update=False)
    assert abs(s["loss"] - float(loss_t)) <= 5e-5 * abs(float(loss_t))
    for k in RENDER_KEYS:
        assert relerr(s[k], rend_t[k].detach().numpy()) < 2e-5, k
    if max(relerr(s[k], g.numpy()) for k, g in zip(GRAD_KEYS, grads_t)) >= 1e-4:
        o = vo.training_step(fc, B, sc, batch, dtype=np.float32, kinks=True)
        _assert_grads_match_aten_port_up_to_kinks(s, o, grads_t, n)


def test_render_only_equals_fwd_bwd_renders():
    c = cases.build_case("ragged")
    a, b = _run(c), _run(c, fn="render")
    assert a["loss"] == pytest.approx(b["loss"], rel=1e-6)
    for k in RENDER_KEYS + ["var"]:
        np.testing.assert_allclose(a[k], b[k], rtol=1e-6, atol=1e-7)


def test_strided_frame_slices_like_train_py():
    """train.py:271-277 hands over non-contiguous slices [:, i*R:(i+1)*R] of the per-frame tensors."""
    n, R, S, iters = 4, 24, 10, 3
    fc, B, sc = synth.make_params(n, 32, seed=11)
    frame = synth.make_batch(n, R * iters, S, seed=12)
    op = make_op(n, R, S, 32, device=DEV)
    tfc = [torch.from_numpy(a).to(DEV) for a in fc]
    tB, tsc = torch.from_numpy(B).to(DEV), torch.from_numpy(sc).to(DEV)
    fr = {k: torch.from_numpy(v).to(DEV) for k, v in frame.items()}
    i = 1
    sl = slice(i * R, (i + 1) * R)
    gfc = [torch.zeros_like(t) for t in tfc]
    gB = torch.zeros_like(tB)
    res = op.fwd_bwd(tfc, tB, tsc, fr["pcs"][:, sl], fr["z"][:, sl], fr["gt_depth"][:, sl], fr["gt_rgb"][:, sl],
                     fr["sem"][:, sl], fr["depth_mask"].bool()[:, sl], grads_fc=gfc, grad_B=gB, render=True)
    assert not fr["pcs"][:, sl].is_contiguous()
    sub = {k: np.ascontiguousarray(v[:, sl]) for k, v in frame.items()}
    o = vo.training_step(fc, B, sc, sub, dtype=np.float32)
    torch.cuda.synchronize()
    assert abs(float(res.loss[0]) - o["loss"]) <= 5e-5 * abs(o["loss"])
    for t in range(14):
        assert relerr(gfc[t].cpu().numpy(), o[f"g_fc{t}"]) < 1e-4
    assert relerr(gB.cpu().numpy(), o["g_B"]) < 1e-4
    assert relerr(res.render_depth.cpu().numpy(), o["render_depth"]) < 2e-5


def test_slab_views_as_parameters():
    """Parameters may be strided views into one [n, P] slab (object stride = P), as well as torch.stack outputs."""
    c = cases.build_case("tiny")
    n, H = c["n"], c["H"]
    P = layout.param_count(H)
    slab = torch.zeros(n, P, device=DEV)
    gslab = torch.full((n, P), float("nan"), device=DEV)
    offs = layout.flat_offsets(H)
    views, gviews = [], []
    shapes = list(layout.fc_shapes(H)) + [layout.PE_B_SHAPE]
    arrays = c["fc"] + [c["B"]]
    for t, shp in enumerate(shapes):
        sz = layout.numel(shp)
        v = slab[:, offs[t]:offs[t] + sz].view((n,) + tuple(shp))
        v.copy_(torch.from_numpy(arrays[t]).to(DEV))
        views.append(v)
        gviews.append(gslab[:, offs[t]:offs[t] + sz].view((n,) + tuple(shp)))
    _, _, sc, b = _to_dev(c)
    op = make_op(n, c["R"], c["S"], H, device=DEV)
    op.fwd_bwd(views[:14], views[14], sc, b["pcs"], b["z"], b["gt_depth"], b["gt_rgb"], b["sem"], b["depth_mask"],
               grads_fc=gviews[:14], grad_B=gviews[14])
    torch.cuda.synchronize()
    g = load_golden("tiny")
    for t in range(14):
        assert relerr(gviews[t].cpu().numpy(), g[f"g_fc{t}"]) < 1e-4
    assert relerr(gviews[14].cpu().numpy(), g["g_B"]) < 1e-4


@pytest.mark.parametrize("weights", ["f32", "bf16"])
@pytest.mark.parametrize("nw", [1, 2, 3, 10])
def test_workgroups_per_object_does_not_change_results(nw, weights):
    """1, 2, 3 workgroups per object = the multi-pass instantiation (4, 5, 10 passes), 10 = single pass; bf16: the W3 = false
    instantiations against the reference evaluated on bfloat16-rounded parameters (fixture scannet_scale_bf16)."""
    c = cases.build_case("scannet_scale")      # R=120 -> 10 ray groups per object
    g = load_golden("scannet_scale" if weights == "f32" else "scannet_scale_bf16")
    op = make_op(c["n"], c["R"], c["S"], c["H"], device=DEV, weights=weights, tuning={"workgroups_per_object": nw})
    s = _run(c, op=op)
    assert abs(s["loss"] - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for k in RENDER_KEYS + GRAD_KEYS:
        assert relerr(s[k], g[k]) < 1e-4, k
    assert relerr(s["var"], g["var"]) < _var_tol(g)


def test_far_point_cold_path():
    c = cases.build_case("tiny")
    c["batch"]["pcs"][1, 3, 4, :] = [3.0e5, -2.0e5, 1.0e5]
    o = vo.training_step(c["fc"], c["B"], c["scale"], c["batch"], dtype=np.float32)
    s = _run(c)
    for k in RENDER_KEYS + GRAD_KEYS:
        assert relerr(s[k], o[k]) < 1e-4, k


def test_fused_adamw_equals_torch_adamw_on_same_gradients():
    """One train_steps() step == fwd_bwd gradients fed to torch.optim.AdamW (train.py:67,325)."""
    c = cases.build_case("tiny")
    fc, B, sc, b = _to_dev(c)
    op = make_op(c["n"], c["R"], c["S"], c["H"], device=DEV)
    ref_p = [t.clone().requires_grad_() for t in fc + [B]]
    opt = torch.optim.AdamW(ref_p, lr=1e-3, weight_decay=0.013)
    st = step.FusedAdamWState(c["n"], c["H"], DEV)
    args = (b["pcs"], b["z"], b["gt_depth"], b["gt_rgb"], b["sem"], b["depth_mask"])
    for it in range(3):
        g = [torch.zeros_like(t) for t in ref_p]
        with torch.no_grad():
            op.fwd_bwd([p.detach() for p in ref_p[:14]], ref_p[14].detach(), sc, *args, grads_fc=g[:14], grad_B=g[14])
        for p, gg in zip(ref_p, g):
            p.grad = gg
        opt.step()
        op.train_steps(fc, B, sc, *args, opt=st, n_steps=1)
        torch.cuda.synchronize()
        for p, q in zip(ref_p, fc + [B]):
            d = (p.detach() - q).abs()
            # the two paths see gradients that differ by summation order (~1e-7): Adam may flip the sign of an
            # lr-sized update where a gradient is ~0, everything else agrees to rounding
            assert float(d.max()) <= (it + 1) * 2.1e-3
            assert float(d.median()) < 1e-7
    assert st.step == 3


@pytest.mark.parametrize("name", ["tiny", "scannet_scale"])
def test_train_steps_tracks_reference_adamw_trajectory(name):
    """3 full steps on a fixed batch vs the reference's functorch + torch.optim.AdamW run (fixture)."""
    c = cases.build_case(name)
    g = load_golden(name)
    fc, B, sc, b = _to_dev(c)
    op = make_op(c["n"], c["R"], c["S"], c["H"], device=DEV)
    st = step.FusedAdamWState(c["n"], c["H"], DEV)
    frame = {k: torch.cat([v, v, v], dim=1).contiguous() for k, v in b.items()}      # same batch 3 times
    res = op.train_steps(fc, B, sc, frame["pcs"], frame["z"], frame["gt_depth"], frame["gt_rgb"], frame["sem"],
                         frame["depth_mask"], opt=st, n_steps=3)
    torch.cuda.synchronize()
    losses = res.loss.cpu().numpy()
    for i in range(3):
        assert abs(losses[i] - g["adamw_losses"][i]) <= 2e-4 * abs(g["adamw_losses"][i])
    ref = [g[f"adamw_p_fc{t}"] for t in range(14)] + [g["adamw_p_B"]]
    diff = np.concatenate([np.abs(p.cpu().numpy().astype(np.float64) - r).ravel() for p, r in zip(fc + [B], ref)])
    assert diff.max() <= 3 * 1e-3 * 1.2
    assert np.quantile(diff, 0.99) < 2e-5 and np.median(diff) < 1e-6


def test_full_size_properties_headline_config():
    """BASELINE configs[1] (20 x 120 x 10, H=32): fixture parity + size-independent properties."""
    c = cases.build_case("cfg2")
    g = load_golden("cfg2")
    op = make_op(c["n"], c["R"], c["S"], c["H"], device=DEV)
    s = _run(c, op=op)
    for k in RENDER_KEYS:
        assert relerr(s[k], g[k]) < 2e-5, k
    for k in GRAD_KEYS:
        assert relerr(s[k], g[k]) < 1e-4, k
    # objects are independent units: permuting them permutes every output (flags/loss unchanged)
    perm = np.random.default_rng(0).permutation(c["n"])
    cp = dict(c, fc=[a[perm] for a in c["fc"]], B=c["B"][perm], scale=c["scale"][perm],
              batch={k: np.ascontiguousarray(v[perm]) for k, v in c["batch"].items()})
    sp = _run(cp, op=op)
    assert sp["loss"] == pytest.approx(s["loss"], rel=2e-6)
    for k in RENDER_KEYS + GRAD_KEYS:
        assert relerr(sp[k], s[k][perm]) < 2e-6, k
    # ray order inside an object is a pure summation order
    rperm = np.random.default_rng(1).permutation(c["R"])
    cr = dict(c, batch={k: np.ascontiguousarray(v[:, rperm]) for k, v in c["batch"].items()})
    sr = _run(cr, op=op)
    assert relerr(sr["render_depth"], s["render_depth"][:, rperm]) < 1e-6
    for k in GRAD_KEYS:
        assert relerr(sr[k], s[k]) < 2e-5, k
    # repeatability (LDS accumulation order across the 4 waves is the only non-determinism)
    s2 = _run(c, op=op)
    for k in GRAD_KEYS:
        assert relerr(s2[k], s[k]) < 2e-6, k


@pytest.mark.parametrize("name", ["cfg2", "scannet_scale"])
def test_results_are_bitwise_repeatable(name):
    """No atomics anywhere and every cross-wave / cross-workgroup sum is ordered: 40 runs of the same step (single-pass
    and multi-pass kernels) must be bit-identical.  A missing barrier or a reused exchange tile in the hand-pipelined
    backward would show up here as a flipped bit long before it moves a 1e-4 tolerance."""
    c = cases.build_case(name)
    fc, B, sc, b = _to_dev(c)
    # scannet_scale: 3 of 10 ray groups per workgroup -> the multi-pass kernel
    op = make_op(c["n"], c["R"], c["S"], c["H"], device=DEV,
                       tuning={"workgroups_per_object": 0 if name == "cfg2" else 3})
    ref = None
    for it in range(40):
        gfc = [torch.full_like(t, float("nan")) for t in fc]
        gB = torch.full_like(B, float("nan"))
        res = op.fwd_bwd(fc, B, sc, b["pcs"], b["z"], b["gt_depth"], b["gt_rgb"], b["sem"], b["depth_mask"],
                         grads_fc=gfc, grad_B=gB, render=True)
        cur = [res.loss.clone(), res.render_depth.clone(), res.render_color.clone()] + gfc + [gB]
        if ref is None:
            ref = cur
        else:
            for x, y in zip(cur, ref):
                assert torch.equal(x, y), it


def test_driver_background_on_second_stream_equals_sequential():
    """HipMapper.train_frame_with_background: the object stack and the one-object background stack (hidden 128) train
    side by side on two streams; both end bit-identical to training them one after the other."""
    from vmap_amd.driver import HipMapper
    from vmap_amd.trainer import SimpleConfig, Trainer
    cfg = SimpleConfig(training_device=DEV, n_iter_per_frame=4)
    iters, n, R, S, Rb, Sb = 4, 3, 24, 10, 40, 14

    def build():
        torch.manual_seed(5)
        m = HipMapper(cfg, device=DEV)
        for _ in range(n):
            m.add_object(Trainer(SimpleConfig(training_device=DEV, hidden_feature_size=32)))
        tb = Trainer(SimpleConfig(training_device=DEV, hidden_feature_size=128, obj_scale=5.0))
        m.attach_background(tb, Rb, Sb)
        return m, tb

    ob = synth.make_batch(n, R * iters, S, seed=31)
    bb = synth.make_batch(1, Rb * iters, Sb, seed=32)
    keys = ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask")
    obj_batch = tuple(torch.from_numpy(ob[k]).to(DEV) for k in keys)
    bg_batch = tuple(torch.from_numpy(bb[k]).to(DEV) for k in keys)
    a, ta = build()
    for _ in range(2):
        ra, rab = a.train_frame_with_background(obj_batch, bg_batch)
    b, tb = build()
    for _ in range(2):
        rb = b.train_frame(*obj_batch)
        bgs = b.bg
        rbb = bgs["op"].train_steps(bgs["views"][:14], bgs["views"][14], bgs["scale"], *bg_batch, opt=bgs["opt"], n_steps=iters, ray_step=Rb)
    torch.cuda.synchronize()
    assert torch.equal(a.slab, b.slab) and torch.equal(a.bg["slab"], b.bg["slab"])
    assert torch.equal(ra.loss, rb.loss) and torch.equal(rab.loss, rbb.loss)
    assert not torch.equal(a.bg["slab"], torch.zeros_like(a.bg["slab"]))
    # the background trainer's modules are views of the trained slab
    assert ta.fc_occ_map.in_layer[0].weight.data_ptr() == a.bg["views"][0].data_ptr() if hasattr(ta.fc_occ_map, "in_layer") else True


def test_autograd_batch_loss_drives_torch_adamw_like_train_py():
    """train.py:303-325 shape of use: loss tensor (+ another differentiable term) -> backward() -> torch.optim.AdamW.step()
    on the stacked leaf tensors; gradients equal the fused call's, the extra term's gradient is added by autograd."""
    c = cases.build_case("ragged")
    fc, B, sc, b = _to_dev(c)
    params = [t.clone().requires_grad_() for t in fc] + [B.clone().requires_grad_()]
    op = make_op(c["n"], c["R"], c["S"], c["H"], device=DEV)
    optimiser = torch.optim.AdamW(params, lr=1e-3, weight_decay=0.013)
    loss = step.batch_loss(op, params[:14], params[14], sc, b["pcs"], b["z"], b["gt_depth"], b["gt_rgb"], b["sem"], b["depth_mask"])
    assert loss.requires_grad and loss.dim() == 0
    extra = 0.5 * (params[1] ** 2).sum()            # stands for `batch_loss += bg_loss` (train.py:316)
    total = loss + extra
    total.backward()
    ref = _run(c)
    assert float(loss) == pytest.approx(ref["loss"], rel=1e-6)
    for t in range(14):
        want = ref[f"g_fc{t}"] + (fc[1].cpu().numpy() if t == 1 else 0.0)
        assert relerr(params[t].grad.cpu().numpy(), want) < 2e-6, t
    assert relerr(params[14].grad.cpu().numpy(), ref["g_B"]) < 2e-6
    before = [p.detach().clone() for p in params]
    optimiser.step()
    assert all(not torch.equal(a, p.detach()) for a, p in zip(before, params))


def test_unsupported_hidden_width_fails_loudly():
    with pytest.raises(_lib.VmapStepError, match="hidden=48"):
        make_op(4, 32, 10, 48, device=DEV)
    with pytest.raises(_lib.VmapStepError, match="hidden=512"):
        make_op(4, 32, 10, 512, device=DEV)


@pytest.mark.parametrize("name,kernel", [("h64", "gen"), ("bg_h128_s14", "wide"), ("imap_h256", "wide"),
                                         ("bg_h128_s14", "gen"), ("imap_h256", "gen"), ("bg_h128_s14", "wide_multipass"),
                                         ("bg_h128_s14", "ws"), ("bg_h128_s14", "ws_multipass"), ("h64", "ws"), ("h64", "ws_multipass"),
                                         ("bg_h128_s14", "ws1"), ("bg_h128_s14", "ws1_multipass"), ("h64", "ws1"), ("h64", "ws1_multipass"),
                                         ("bg_h128_s14", "ws1_two_tile"), ("bg_h128_s14", "ws1_two_tile_multipass"), ("h64", "ws1_two_tile"),
                                         ("bg_h128_s14", "ws1_three_tile"), ("bg_h128_s14", "ws1_three_tile_multipass"),
                                         ("imap_h256", "auto"), ("imap_h256", "ws1"), ("imap_h256", "ws1_multipass")])
def test_generic_width_kernel_matches_reference_fixture(name, kernel):
    """hidden = 64 / 128 (background model shapes) / 256 (iMAP, BASELINE configs[0]): step_main_wide (tile per
    workgroup, also with fewer workgroups than ray groups), step_main_gen, and - hidden 64 / 128 -
    step_main_wp / step_main_ws (bf16 matrix pipe, split operands: two waves / one wave per output block; automatic
    choice: the first at hidden 64, the second at hidden 128)."""
    c = cases.build_case(name)
    g = load_golden(name)
    tuning = {"kernel": {"auto": _lib.KERNEL_AUTO,                                 # imap_h256: step_main_ws<8> (eight waves), round 3
                         "gen": _lib.KERNEL_GEN, "wide": _lib.KERNEL_WIDE4, "wide_multipass": _lib.KERNEL_WIDE4,
                         "ws": _lib.KERNEL_WP, "ws_multipass": _lib.KERNEL_WP,         # step_main_wp (two waves per output block)
                         "ws1": _lib.KERNEL_WS1, "ws1_multipass": _lib.KERNEL_WS1,            # step_main_ws (one wave per block): small batches
                         "ws1_two_tile": _lib.KERNEL_WS1, "ws1_two_tile_multipass": _lib.KERNEL_WS1,   # run single-tile rounds, ws_flags = 1: two-tile rounds
                         "ws1_three_tile": _lib.KERNEL_WS1, "ws1_three_tile_multipass": _lib.KERNEL_WS1}[kernel],   # ws_flags = 2: three-tile rounds
              "workgroups_per_object": {"wide_multipass": 3, "ws_multipass": 3, "ws1_multipass": 3, "ws1_two_tile_multipass": 3,
                                        "ws1_three_tile_multipass": 3}.get(kernel, 0),
              "ws_flags": 1 if "two_tile" in kernel else 2 if "three_tile" in kernel else 0}
    s = _run(c, tuning=tuning)
    assert abs(s["loss"] - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for k in RENDER_KEYS:
        assert relerr(s[k], g[k]) < 2e-5, k
    assert relerr(s["var"], g["var"]) < _var_tol(g)
    for k in GRAD_KEYS:
        assert not np.isnan(s[k]).any(), k
        assert relerr(s[k], g[k]) < 1e-4, k


@pytest.mark.parametrize("weights", ["f32", "bf16"])
def test_reference_imap_batch_matches_reference_fixture(weights):
    """The reference's OWN iMAP batch (config_replica_room0_iMAP.json:31: n_per_optim 4800 rays x 14 samples, hidden 256): 2400
    single-tile rounds on 240 workgroups = the MULTI-ROUND form of step_main_ws<8> (ten rounds per workgroup, read-modify-write of
    its 1.4 MB gradient row), float32 weights and the bfloat16 form - against the unmodified reference run on the same inputs
    (fixture imap_full / imap_full_bf16: loss, renders, all 15 gradient tensors).  Gradients at north_star's 1e-4; should the
    reference's run and the kernel differ on the derivative bit of kink-adjacent hidden units (67 200 points x 1024 units), the
    bits are accounted for one by one (signed coefficients, conftest.kink_aware)."""
    if TEST_TUNING["default"] is not None:
        pytest.skip("hidden 256: the module's hidden-32 kernel legs do not apply; run once")
    from conftest import round_bf16
    bf16 = weights == "bf16"
    c = cases.build_case("imap_full")
    g = load_golden("imap_full_bf16" if bf16 else "imap_full")
    op = make_op(c["n"], c["R"], c["S"], c["H"], device=DEV, weights=weights)
    plan = op.plan()
    assert plan["kernel"] == "step_main_ws<8>" and plan["single_round"] == 0 and plan["rounds_per_object"] == 2400
    s = _run(c, op=op)
    assert abs(s["loss"] - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for k in RENDER_KEYS:
        assert relerr(s[k], g[k]) < 2e-5, k
    assert relerr(s["var"], g["var"]) < _var_tol(g)
    for k in GRAD_KEYS:
        assert not np.isnan(s[k]).any(), k
    if max(relerr(s[k], g[k]) for k in GRAD_KEYS) >= 1e-4:
        rnd = round_bf16 if bf16 else (lambda a: a)
        o = vo.training_step([rnd(a) for a in c["fc"]], rnd(c["B"]), c["scale"], c["batch"], dtype=np.float32, kinks=True, kink_min_effect=2.5e-5)
        fix = {k: g[k] for k in GRAD_KEYS}
        fix["kink_deltas"] = o["kink_deltas"]
        _assert_grads_match_oracle_up_to_kinks(s, fix, c["n"], tol=1e-4, signed=True)


def test_generic_width_multi_pass_and_train_steps():
    """hidden = 64 with fewer workgroups than ray groups (partials accumulated over passes) + fused AdamW steps."""
    c = cases.build_case("h64")
    g = load_golden("h64")
    s = _run(c, tuning={"workgroups_per_object": 2})
    for k in RENDER_KEYS + GRAD_KEYS:
        assert relerr(s[k], g[k]) < 1e-4, k
    fc, B, sc, b = _to_dev(c)
    op = make_op(c["n"], c["R"], c["S"], c["H"], device=DEV)
    st = step.FusedAdamWState(c["n"], c["H"], DEV)
    frame = {k: torch.cat([v, v], dim=1).contiguous() for k, v in b.items()}
    res = op.train_steps(fc, B, sc, frame["pcs"], frame["z"], frame["gt_depth"], frame["gt_rgb"], frame["sem"],
                         frame["depth_mask"], opt=st, n_steps=2)
    torch.cuda.synchronize()
    losses = res.loss.cpu().numpy()
    assert losses[0] == pytest.approx(float(g["loss"]), rel=2e-5)
    assert np.isfinite(losses).all() and losses[1] < losses[0]      # one AdamW step on the same batch lowers the loss


def test_prepared_split_applies_externally_reduced_flags():
    """vmapstep_prepare -> (flag reduction across ranks) -> vmapstep_train_steps_prepared: a switch raised by ANOTHER
    rank's objects must drop the term here too (render_rays.py:68-73 is batch-wide)."""
    c = cases.build_case("tiny")
    fc, B, sc, b = _to_dev(c)
    op = make_op(c["n"], c["R"], c["S"], c["H"], device=DEV)
    st = step.FusedAdamWState(c["n"], c["H"], DEV, lr=0.0, weight_decay=0.0)      # lr 0: parameters stay put
    gfc = [torch.zeros_like(t) for t in fc]
    gB = torch.zeros_like(B)
    seen = {}

    def inject(flags):                      # what ObjectShard.reduce_flags would deliver if a peer had an empty depth mask
        seen["local"] = flags.clone()
        flags[:, 0] = 1

    res = op.train_steps(fc, B, sc, b["pcs"], b["z"], b["gt_depth"], b["gt_rgb"], b["sem"], b["depth_mask"], opt=st,
                         n_steps=1, grads_fc=gfc, grad_B=gB, flag_reduce=inject)
    torch.cuda.synchronize()
    assert seen["local"][0, :3].tolist() == [0, 0, 0]
    o = vo.training_step(c["fc"], c["B"], c["scale"], c["batch"], dtype=np.float32, drop=[True, False, False])
    assert float(res.loss[0]) == pytest.approx(o["loss"], rel=2e-5)
    assert res.flags[0].tolist()[:3] == [1, 0, 0]
    for t in range(14):
        assert relerr(gfc[t].cpu().numpy(), o[f"g_fc{t}"]) < 1e-4
    assert relerr(gB.cpu().numpy(), o["g_B"]) < 1e-4


def test_shared_background_hip_tracks_the_aten_port_over_a_frame():
    """The background model (hidden 128, 14 samples; train.py:308-316) through parallel.SharedBackgroundHip at world size 1
    (the collectives are identities; the per-frame count plumbing, vmapstep_fwd_bwd_prepared(step i) and
    vmapstep_adamw_apply are exercised) against the ATen port of the oracle: 4 distinct steps of one frame."""
    from oracle import vmap_oracle_torch as vt
    from vmap_amd import fields, parallel
    H, R, S, steps = 128, 60, 14, 4
    torch.manual_seed(4)
    fc = fields.OccupancyMap(hidden_size=H)
    fc.apply(fields.init_weights)
    pe = fields.UniDirsEmbed(max_deg=5, scale=5.0)
    fc_np = [p.detach().numpy()[None].copy() for p in fc.parameters()]
    B_np = pe.B_layer.weight.detach().numpy()[None].copy()
    sc_np = np.array([5.0], dtype=np.float32)
    b = synth.make_batch(1, R * steps, S, seed=9)
    ref = vt.CpuTrainer(fc_np, B_np, sc_np)
    ref_losses = []
    for i in range(steps):
        l, _, _ = ref.step({k: np.ascontiguousarray(v[:, i * R:(i + 1) * R]) for k, v in b.items()})
        ref_losses.append(float(l))
    hip = parallel.SharedBackgroundHip(fc, pe, R, S, DEV, max_steps=steps)
    bd = {k: torch.from_numpy(v[0]).to(DEV) for k, v in b.items()}
    losses = hip.train_frame(bd["pcs"], bd["z"], bd["gt_depth"], bd["gt_rgb"], bd["sem"], bd["depth_mask"], steps).cpu().numpy()
    assert hip.opt.step == steps
    for i in range(steps):
        assert losses[i] == pytest.approx(ref_losses[i], rel=2e-4), i
    hip.write_back()
    for p, q in zip(list(fc.parameters()) + [pe.B_layer.weight], ref.fc + [ref.B]):
        d = (p.detach().cpu() - q.detach()[0]).abs()
        assert float(d.max()) <= steps * 1.1e-3 and float(d.median()) < 1e-6


@pytest.mark.parametrize("weights", ["f32", "bf16"])
def test_hidden256_frame_tracks_the_aten_port(weights):
    """The iMAP field (hidden 256, BASELINE configs[0]: ONE model for the whole scene) over a frame of 6 distinct steps through
    vmapstep_train_steps on step_main_ws<8> + step_finalize_ws<8> (fused AdamW, W / W^T images kept current) against the ATen
    port of the oracle (torch.optim.AdamW; bf16: run-time weights rounded from the float32 masters every step)."""
    from oracle import vmap_oracle_torch as vt
    n, R, S, H, steps = 1, 100, 14, 256, 6
    fc, B, sc = synth.make_params(n, H, scale=10.0, seed=801)
    b = synth.make_batch(n, R * steps, S, seed=802)
    ref = vt.CpuTrainer(fc, B, sc, weights_bf16=weights == "bf16")
    ref_losses = []
    for i in range(steps):
        l, _, _ = ref.step({k: np.ascontiguousarray(v[:, i * R:(i + 1) * R]) for k, v in b.items()})
        ref_losses.append(float(l))
    tfc = [torch.from_numpy(a).to(DEV) for a in fc]
    tB, tsc = torch.from_numpy(B).to(DEV), torch.from_numpy(sc).to(DEV)
    fr = {k: torch.from_numpy(v).to(DEV) for k, v in b.items()}
    op = make_op(n, R, S, H, device=DEV, max_steps=steps, weights=weights)
    st = step.FusedAdamWState(n, H, DEV)
    res = op.train_steps(tfc, tB, tsc, fr["pcs"], fr["z"], fr["gt_depth"], fr["gt_rgb"], fr["sem"], fr["depth_mask"], opt=st, n_steps=steps)
    torch.cuda.synchronize()
    losses = res.loss.cpu().numpy()
    assert int(res.flags[:, 3].max()) == 0
    for i in range(steps):
        assert losses[i] == pytest.approx(ref_losses[i], rel=2e-4), (i, losses, ref_losses)
    for p, q in zip(tfc + [tB], ref.fc + [ref.B]):
        d = (p.cpu() - q.detach()).abs()
        assert float(d.max()) <= steps * 1.1e-3 and float(d.median()) < 1e-6


def test_headless_driver_object_list_semantics():
    """HipMapper: add objects, train a frame (20 steps), add another object -> re-stack with optimiser restart,
    modules stay live views of the trained weights; first frame tracks the PyTorch port's AdamW trajectory."""
    from oracle import vmap_oracle_torch as vt
    from vmap_amd import driver, trainer
    cfg = trainer.SimpleConfig(training_device=DEV, n_iter_per_frame=4)
    cfg.obj_id = 1
    torch.manual_seed(0)
    n, R, S = 3, 24, 10
    trs = [trainer.Trainer(cfg) for _ in range(n + 1)]
    snap = [[p.detach().cpu().numpy().copy() for p in list(t.fc_occ_map.parameters()) + [t.pe.B_layer.weight]] for t in trs]
    m = driver.HipMapper(cfg)
    for t in trs[:n]:
        m.add_object(t)
    frame = synth.make_batch(n, R * cfg.n_iter_per_frame, S, seed=77)
    fr = {k: torch.from_numpy(v).to(DEV) for k, v in frame.items()}
    res = m.train_frame(fr["pcs"], fr["z"], fr["gt_depth"], fr["gt_rgb"], fr["sem"], fr["depth_mask"].bool())
    m.check_flags(res)
    # reference trajectory: the PyTorch port stepping over the same slices with torch.optim.AdamW
    fc0 = [np.stack([snap[k][t] for k in range(n)]) for t in range(14)]
    B0 = np.stack([snap[k][14] for k in range(n)])
    ref = vt.CpuTrainer(fc0, B0, np.full(n, 2.0, np.float32))
    for i in range(cfg.n_iter_per_frame):
        sub = {k: np.ascontiguousarray(v[:, i * R:(i + 1) * R]) for k, v in frame.items()}
        loss_i, _, _ = ref.step(sub)
        assert float(res.loss[i]) == pytest.approx(float(loss_i), rel=3e-4)
    # modules are live views: their parameters changed without any write-back call
    p_now = trs[0].fc_occ_map.in_layer[0].weight
    assert p_now.data_ptr() == m.views[0][0].data_ptr()
    assert float((p_now.detach().cpu() - torch.from_numpy(snap[0][0])).abs().max()) > 1e-4
    for k in range(n):
        for t, q in enumerate(ref.fc + [ref.B]):
            d = (m.views[t][k].cpu() - q.detach()[k]).abs()
            assert float(d.max()) <= cfg.n_iter_per_frame * 1.2e-3 and float(d.median()) < 2e-6
    # a new object arrives: re-stack, moments restart (utils.py:33), old objects keep their trained weights
    trained0 = m.views[2][1].clone()
    m.add_object(trs[n])
    frame2 = synth.make_batch(n + 1, R * cfg.n_iter_per_frame, S, seed=78)
    fr2 = {k: torch.from_numpy(v).to(DEV) for k, v in frame2.items()}
    assert m._dirty
    res2 = m.train_frame(fr2["pcs"], fr2["z"], fr2["gt_depth"], fr2["gt_rgb"], fr2["sem"], fr2["depth_mask"].bool())
    torch.cuda.synchronize()
    assert m.op.n_obj == n + 1 and m.opt.step == cfg.n_iter_per_frame          # fresh optimiser state
    assert float((m.views[2][1] - trained0).abs().max()) < cfg.n_iter_per_frame * 1.2e-3   # continued from trained weights
    assert torch.isfinite(res2.loss).all()


@pytest.mark.parametrize("name,kernel", [("scannet_scale", 0), ("h64", 0), ("h64", _lib.KERNEL_WS1), ("bg_h128_s14", 0),
                                         ("bg_h128_s14", _lib.KERNEL_WP)])
def test_bf16_weight_mode_equals_reference_on_rounded_weights(name, kernel):
    """BASELINE configs[3]/[4] 'bf16 weights + fp32 accumulate': masters fp32, image rounded to bfloat16.  Comparators: the
    UNMODIFIED reference evaluated on the rounded parameters (fixture <name>_bf16.npz) and the ATen port of the oracle."""
    from conftest import tensor_err_q, round_bf16
    from oracle import vmap_oracle_torch as vt
    c = cases.build_case(name)
    g = load_golden(name + "_bf16")
    fc_r = [round_bf16(a) for a in c["fc"]]
    B_r = round_bf16(c["B"])
    loss_t, rend_t, grads_t = vt.CpuTrainer(fc_r, B_r, c["scale"]).step(c["batch"], update=False)
    op = make_op(c["n"], c["R"], c["S"], c["H"], device=DEV, weights="bf16", tuning={"kernel": kernel} if kernel else None)
    s = _run(c, op=op)
    assert abs(s["loss"] - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for k in RENDER_KEYS:
        assert relerr(s[k], g[k]) < 2e-5, k
    assert relerr(s["var"], g["var"]) < _var_tol(g)
    for k in GRAD_KEYS:
        assert relerr(s[k], g[k]) < 1e-4, k
    assert abs(s["loss"] - float(loss_t)) <= 5e-5 * abs(float(loss_t))
    for k in RENDER_KEYS:
        assert relerr(s[k], rend_t[k].detach().numpy()) < 2e-5, k
    for k, gt_ in zip(GRAD_KEYS, grads_t):
        assert relerr(s[k], gt_.numpy()) < 1e-4, k
    # masters stay fp32: one fused AdamW step moves them by ~lr although that is far below bf16 resolution
    fc, B, sc, b = _to_dev(c)
    st = step.FusedAdamWState(c["n"], c["H"], DEV)
    before = fc[2].clone()
    op.train_steps(fc, B, sc, b["pcs"], b["z"], b["gt_depth"], b["gt_rgb"], b["sem"], b["depth_mask"], opt=st, n_steps=1)
    torch.cuda.synchronize()
    d = (fc[2] - before).abs()
    assert 0 < float(d.max()) < 1.2e-3


@pytest.mark.parametrize("name,weights,slab", [("cfg2", "f32", False), ("cfg2", "f32", True), ("scannet_scale", "bf16", False),
                                               ("tiny", "bf16", True)])
def test_table_driven_finalize_is_bit_identical_to_generic_finalize(name, weights, slab, h32_kernel):
    """step_finalize_h32 (image slot from step_prep's table, compile-time tensor offsets or one slab base) against the
    generic step_finalize (runtime tensor search + gen_image_index per element): same ordered sums, same adamw_elem.
    Three frames of 20 steps; steps 2..20 of a frame read the parameter image the finalize maintains (bf16: the rounded
    copy), so losses, parameters and both moments must agree bit for bit."""
    if h32_kernel != "f32":
        pytest.skip("step_finalize_h32 / step_finalize belong to the exact-fp32 kernel's image")
    c = cases.build_case(name)
    steps = 20
    outs = []
    for generic in (1, 0):                                       # generic / table-driven
        fc, B, sc, b = _to_dev(c)
        if slab:
            _, fc, B = layout.stack_in_slab(fc, B)
        op = make_op(c["n"], c["R"], c["S"], c["H"], device=DEV, max_steps=steps, weights=weights,
                           tuning={"generic_finalize": generic})
        st = step.FusedAdamWState(c["n"], c["H"], DEV)
        frame = {k: torch.cat([v.roll(i, dims=1) for i in range(steps)], dim=1).contiguous() for k, v in b.items()}
        losses = []
        for _ in range(3):
            res = op.train_steps(fc, B, sc, frame["pcs"], frame["z"], frame["gt_depth"], frame["gt_rgb"], frame["sem"],
                                 frame["depth_mask"], opt=st, n_steps=steps)
            losses.append(res.loss.clone())
        torch.cuda.synchronize()
        outs.append(dict(p=[t.clone() for t in fc + [B]], m=st.exp_avg.clone(), v=st.exp_avg_sq.clone(), losses=torch.stack(losses)))
    a, b_ = outs
    assert bool(torch.isfinite(a["losses"]).all())
    assert torch.equal(a["losses"], b_["losses"])
    for x, y in zip(a["p"], b_["p"]):
        assert torch.equal(x, y)
    assert torch.equal(a["m"], b_["m"]) and torch.equal(a["v"], b_["v"])


@pytest.mark.parametrize("H,n,R,S,weights", [(64, 24, 40, 10, "bf16"), (64, 12, 64, 10, "f32"), (128, 16, 24, 14, "f32"), (256, 2, 16, 14, "f32")])
def test_ws_finalize_one_thread_per_quad_is_bit_identical_to_the_grouped_form(H, n, R, S, weights):
    """Shapes with many finalize blocks and few gradient rows per object (>= 512 blocks, <= 16 rows: the launcher's rule, as at configs[4]):
    step_finalize_ws runs with ONE thread per quad that walks all eight row groups; tuning.generic_finalize = 1 forces the form with a
    thread per quad and row group (what few-block shapes like the background step get).  Same ordered sums: two 5-step frames agree bit
    for bit in losses, parameters and both moments.  (Round 5: at 256 objects x 2 rows the grouped form left seven of eight threads
    without a row and ran at 1.35 TB/s.)"""
    if TEST_TUNING["default"] is not None:
        pytest.skip("the module's hidden-32 kernel legs do not apply; run once")
    fc0, B0, sc0 = synth.make_params(n, H, seed=2100 + H)
    steps = 5
    fr0 = synth.make_batch(n, R * steps, S, seed=2101 + H)
    outs = []
    for grouped in (1, 0):
        fc = [torch.from_numpy(a).to(DEV) for a in fc0]
        B, sc = torch.from_numpy(B0).to(DEV), torch.from_numpy(sc0).to(DEV)
        fr = {k: torch.from_numpy(v).to(DEV) for k, v in fr0.items()}
        op = make_op(n, R, S, H, device=DEV, max_steps=steps, weights=weights, tuning={"generic_finalize": grouped})
        plan = op.plan()
        assert plan["workgroups_per_object"] <= 16, plan
        st = step.FusedAdamWState(n, H, DEV)
        losses = []
        for _ in range(2):
            res = op.train_steps(fc, B, sc, fr["pcs"], fr["z"], fr["gt_depth"], fr["gt_rgb"], fr["sem"], fr["depth_mask"], opt=st, n_steps=steps)
            losses.append(res.loss.clone())
        torch.cuda.synchronize()
        outs.append(dict(p=[t.clone() for t in fc + [B]], m=st.exp_avg.clone(), v=st.exp_avg_sq.clone(), losses=torch.stack(losses)))
    a, b_ = outs
    assert bool(torch.isfinite(a["losses"]).all())
    assert torch.equal(a["losses"], b_["losses"])
    for x, y in zip(a["p"], b_["p"]):
        assert torch.equal(x, y)
    assert torch.equal(a["m"], b_["m"]) and torch.equal(a["v"], b_["v"])


FRAME_TOL = {   # (relative loss tolerance for steps < 5, for later steps, q99 / median of |final parameter - reference|)
    # Measured per step (tests/tools/frame_drift.py, profiles/r03n_frame_drift.json): over the 20 steps of the headline frame the
    # default kernel stays within 2.3e-5 of the reference's loop (the exact-fp32 kernel within 3.1e-6; the reference's own float32
    # and float64 runs differ by up to 30 % per step on this ill-conditioned loss, but its two float32 paths - vmap / forloop -
    # agree to 3e-7, and so, nearly, does this kernel).  Bounds: 5e-5 for the first steps, north_star's 1e-4 to the end.
    "cfg2_frame20": (5e-5, 1e-4, 3e-4, 2e-5),
    "scannet50_frame": (5e-5, 1e-4, 2e-5, 2e-6),
    "h64_r256_frame": (5e-5, 1e-4, 2e-5, 2e-6),
    "bg128_frame": (5e-5, 1e-4, 2e-5, 2e-6),
    # bf16 run-time weights over fp32 masters (measured: 3.3e-6 / 7.8e-7 at the two object shapes).  bg128_frame_bf16: the
    # reference's own run and this kernel sit on opposite sides of ONE ReLU kink in step 0 (accounted for bit by bit in the
    # first-step gradient check below); the two trajectories then separate as far as one hidden unit's gradient moves the
    # masters across bfloat16 rounding boundaries (6e-5 .. 2.5e-4).  Round 4: (i) the fixture holds BOTH branches of that bit - the
    # reference's loop re-run with the first step's gradients on the other side (alt_*, tests/golden/make_frame_goldens.py): the
    # kernel matches NEITHER better than the other, so the kink is not what separates them; (ii) what does was measured on the
    # reference itself (profiles/round4c_bf16_master_rounding_sensitivity.json): perturbing its step-0 gradients by 1e-5 relative
    # (the kernel's backward products are good to 2^-16) moves its own loss of step 3 by up to 6e-5 with bf16 run-time weights and
    # by nothing (2e-7) with float32 weights - a master crossing a bfloat16 rounding boundary moves a run-time weight by 2^-8.
    # Steps 0 and 1 (before any master can differ: Adam's first update is lr * sign(g)) are held to 5e-5 (measured: exact), the
    # later steps of THIS fixture to 1e-3 - and test_bf16_frame_steps_teacher_forced holds every step of every bf16 frame to the
    # real bar (loss 2e-5, gradients 1e-4) against the ATen port evaluated on the kernel's OWN masters of that step.
    "scannet50_frame_bf16": (5e-5, 1e-4, 2e-5, 2e-6),
    "h64_r256_frame_bf16": (5e-5, 1e-4, 2e-5, 2e-6),
    "bg128_frame_bf16": (5e-5, 1e-3, 2e-5, 2e-6),
}
# steps whose loss must meet the EARLY bound (the others the late one): bg128_frame_bf16 holds steps 0 and 1 to 5e-5 (measured: exact) -
# from step 2 on a bf16-weights trajectory is as sensitive as profiles/round4c_bf16_master_rounding_sensitivity.json shows
FRAME_EARLY_STEPS = {"bg128_frame_bf16": 2}


@pytest.mark.parametrize("name", list(cases.FRAME_CASES) + [f"{n}_bf16" for n in cases.BF16_FRAME_CASES])
def test_frame_trajectory_matches_reference_step_loop(name):
    """vmapstep_train_steps over a whole frame - distinct strided ray slices per step, fused AdamW - against the
    reference's OWN loop (train.py:270-326: functorch vmap + loss.step_batch_loss + torch.optim.AdamW), fixture
    tests/golden/<name>.npz.  cfg2_frame20 is the frame bench.py times; scannet50_frame runs the multi-pass kernel at a
    real object count; h64_r256_frame is the per-GPU shape of BASELINE configs[4]; bg128_frame the background model's
    (hidden 128, 14 samples: step_main_ws, several rounds per workgroup).  <name>_bf16: weight_dtype = bf16 at the SAME shapes
    (50 objects: step_main_s32<BWD, MULTI, ., W3 = false>; 32 x 256 rays at hidden 64: multi-round step_main_wp<2, ., W3 = false>)
    against the reference loop run with bfloat16-rounded run-time weights over full-precision masters."""
    # bg128_frame (240 rays = 120 tiles) runs step_main_ws with single-tile rounds by default; its f32 "f32" leg of the module's
    # kernel parametrisation runs the two-tile form instead, so that both are held to the reference's own loop
    two_tile = name.startswith("bg128") and TEST_TUNING["default"] is not None
    _check_frame_trajectory(name, {"ws_flags": 1} if two_tile else None)


@pytest.mark.parametrize("name", list(cases.BF16_FRAME_CASES))
def test_bf16_frame_steps_teacher_forced(name):
    """Every step of the bf16-weights frames held to north_star's bar WITHOUT the chaos of the mode's trajectory (a master that
    crosses a bfloat16 rounding boundary moves a run-time weight by 2^-8: profiles/round4c_bf16_master_rounding_sensitivity.json):
    before step i the kernel's OWN float32 masters are read back, the ATen port (the kernels the reference runs) evaluates loss and
    gradients of step i's ray slice on their bfloat16 rounding, and the kernel's loss / gradients of the same step must agree to
    2e-5 / 1e-4 (ReLU kinks accounted for, signed); then the kernel's fused AdamW advances the masters and the next step is checked
    on the new ones."""
    if TEST_TUNING["default"] is not None and cases.FRAME_CASES[name][3] != 32:
        pytest.skip("hidden 64 / 128: the module's hidden-32 kernel legs do not apply; run once")
    from conftest import round_bf16
    from oracle import vmap_oracle_torch as vt
    c = cases.build_frame_case(name)
    n, R, S, H, steps = c["n"], c["R"], c["S"], c["H"], c["n_steps"]
    fc = [torch.from_numpy(a).to(DEV) for a in c["fc"]]
    B = torch.from_numpy(c["B"]).to(DEV)
    sc = torch.from_numpy(c["scale"]).to(DEV)
    fr = {k: torch.from_numpy(v).to(DEV) for k, v in c["frame"].items()}
    op = make_op(n, R, S, H, device=DEV, max_steps=1, weights="bf16")
    st = step.FusedAdamWState(n, H, DEV)
    keys = ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask")
    for i in range(steps):
        sl = slice(i * R, (i + 1) * R)
        masters = [t.cpu().numpy() for t in fc + [B]]
        sub = {k: np.ascontiguousarray(v[:, sl]) for k, v in c["frame"].items()}
        loss_t, _, grads_t = vt.CpuTrainer([round_bf16(a) for a in masters[:14]], round_bf16(masters[14]), c["scale"]).step(sub, update=False)
        gfc = [torch.zeros_like(t) for t in fc]
        gB = torch.zeros_like(B)
        res = op.fwd_bwd(fc, B, sc, *(fr[k][:, sl] for k in keys), grads_fc=gfc, grad_B=gB)
        torch.cuda.synchronize()
        assert abs(float(res.loss[0]) - float(loss_t)) <= 2e-5 * abs(float(loss_t)), i
        got = {(f"g_fc{t}" if t < 14 else "g_B"): (gfc[t] if t < 14 else gB).cpu().numpy() for t in range(15)}
        if max(relerr(got[k], g_.numpy()) for k, g_ in zip(GRAD_KEYS, grads_t)) >= 1e-4:
            o = vo.training_step([round_bf16(a) for a in masters[:14]], round_bf16(masters[14]), c["scale"], sub, dtype=np.float32, kinks=True)
            _assert_grads_match_aten_port_up_to_kinks(got, o, grads_t, n)
        op.train_steps(fc, B, sc, *(fr[k][:, sl] for k in keys), opt=st, n_steps=1)


@pytest.mark.parametrize("name", ["bg128_frame", "bg128_frame_bf16"])
def test_background_frame_with_three_tile_rounds(name):
    """The same frame fixtures through step_main_ws<4, ., ., ., NT = 3> (tuning.ws_flags = 2; 240 rays = 40 rounds of 6 rays): the
    form a ONE-GPU background step (1200 rays) runs - forward, loss, backward, fused AdamW and the maintained W / W^T images
    over the frame's steps against the reference's own loop, float32 and bfloat16 run-time weights."""
    if TEST_TUNING["default"] is not None:
        pytest.skip("hidden 128: the module's hidden-32 kernel legs do not apply; run once")
    _check_frame_trajectory(name, {"ws_flags": 2})


def _check_frame_trajectory(name, tuning):
    bf16 = name.endswith("_bf16")
    c = cases.build_frame_case(name[:-5] if bf16 else name)
    g = load_golden(name)
    n, R, S, H, steps = c["n"], c["R"], c["S"], c["H"], c["n_steps"]
    fc = [torch.from_numpy(a).to(DEV) for a in c["fc"]]
    B = torch.from_numpy(c["B"]).to(DEV)
    sc = torch.from_numpy(c["scale"]).to(DEV)
    fr = {k: torch.from_numpy(v).to(DEV) for k, v in c["frame"].items()}
    op = make_op(n, R, S, H, device=DEV, max_steps=steps, weights="bf16" if bf16 else "f32", tuning=tuning)
    st = step.FusedAdamWState(n, H, DEV)
    # first-step gradients (same state): fixture parity of the strided slice [0, R)
    gfc = [torch.zeros_like(t) for t in fc]
    gB = torch.zeros_like(B)
    op.fwd_bwd(fc, B, sc, *(fr[k][:, :R] for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask")), grads_fc=gfc, grad_B=gB)
    keep = g["keep"]
    got0 = {(f"g_fc{t}" if t < 14 else "g_B"): (gfc[t] if t < 14 else gB).cpu().numpy()[keep] for t in range(15)}
    fix0 = {(f"g_fc{t}" if t < 14 else "g_B"): g[f"g0_fc{t}" if t < 14 else "g0_B"] for t in range(15)}
    kink_bits_off_the_reference = 0
    if max(relerr(got0[k], fix0[k]) for k in GRAD_KEYS) >= 1e-4:
        # The reference's own float32 run may sit on the other side of a ReLU kink (bg128_frame_bf16: ONE hidden unit of 2.1 M moves
        # its in_layer gradient by 6e-4 against BOTH the numpy oracle and this kernel, which agree to 3e-6).  Account for it bit
        # by bit: the fixture's gradients + a {-1, 0, +1} combination of the oracle's kink deltas, at 1e-4.
        from conftest import round_bf16
        rnd = round_bf16 if bf16 else (lambda a: a)
        sub = {k: np.ascontiguousarray(v[keep][:, :R]) for k, v in c["frame"].items()}
        o = vo.training_step([rnd(a[keep]) for a in c["fc"]], rnd(c["B"][keep]), c["scale"][keep], sub, dtype=np.float32, kinks=True)
        fix0["kink_deltas"] = o["kink_deltas"]
        kink_bits_off_the_reference = _assert_grads_match_oracle_up_to_kinks(got0, fix0, len(keep), tol=1e-4, signed=True)
    res = op.train_steps(fc, B, sc, fr["pcs"], fr["z"], fr["gt_depth"], fr["gt_rgb"], fr["sem"], fr["depth_mask"], opt=st,
                         n_steps=steps, ray_step=R)
    torch.cuda.synchronize()
    losses = res.loss.cpu().numpy().astype(np.float64)
    assert int(res.flags[:, 3].max()) == 0
    early, late, q99, med = FRAME_TOL[name]
    rel = np.abs(losses - g["losses"]) / np.abs(g["losses"])
    pre = ""
    if "alt_losses" in g.files:
        # the fixture holds both sides of a ReLU kink of step 0 (two valid float32 evaluations of the reference's loop).  WHICH one
        # this kernel is on is decided by its own step-0 derivative bits (the kink accounting of the first-step gradients above:
        # raw agreement with the reference's gradients = the reference's side, a flipped bit = the other side), not by best fit
        if kink_bits_off_the_reference > 0:
            rel, pre = np.abs(losses - g["alt_losses"]) / np.abs(g["alt_losses"]), "alt_"
    assert rel[0] <= 2e-5 or not bf16, rel
    assert rel[:FRAME_EARLY_STEPS.get(name, 5)].max() <= early, rel
    assert rel.max() <= late, rel
    diffs = []
    for t in range(15):
        key = pre + (f"p_fc{t}" if t < 14 else "p_B")
        got = (fc[t] if t < 14 else B).cpu().numpy()
        diffs.append(np.abs(got[keep].astype(np.float64) - g[key]).ravel())
        # every object (not only the kept ones): L2 norm of the final parameters per object and tensor
        nk = pre + (f"pnorm_fc{t}" if t < 14 else "pnorm_B")
        pn = np.sqrt((got.astype(np.float64).reshape(n, -1) ** 2).sum(-1))
        assert np.abs(pn - g[nk]).max() <= 2e-3 * g[nk].max(), nk
    d = np.concatenate(diffs)
    assert d.max() <= steps * 1e-3 * 1.2                      # an element can at most walk lr per step
    assert np.quantile(d, 0.99) < q99 and np.median(d) < med, (np.quantile(d, 0.99), np.median(d))


@pytest.mark.parametrize("weights,case", [("f32", "scannet_scale"), ("bf16", "scannet_scale"), ("f32", "bg_h128_s14"), ("bf16", "bg_h128_s14"),
                                          ("f32", "h64"), ("bf16", "h64"), ("f32", "imap_h256"), ("bf16", "imap_h256")])
def test_parameter_image_kept_by_finalize_equals_freshly_packed_image(weights, case):
    """The finalize kernel rewrites the packed parameter image (split planes / float32 image) element by element after
    every AdamW update; a new frame call packs it from the parameter tensors.  Two steps in ONE call (step 2 reads the
    maintained image) must equal two calls of one step (step 2 reads a fresh pack): bit for bit."""
    c = cases.build_case(case)          # bg_h128_s14: the two images (W, W^T) of step_main_ws
    outs = []
    for split_calls in (False, True):
        fc, B, sc, b = _to_dev(c)
        op = make_op(c["n"], c["R"], c["S"], c["H"], device=DEV, max_steps=2, weights=weights)
        st = step.FusedAdamWState(c["n"], c["H"], DEV)
        frame = {k: torch.cat([v, v.roll(3, dims=1)], dim=1).contiguous() for k, v in b.items()}
        args = (frame["pcs"], frame["z"], frame["gt_depth"], frame["gt_rgb"], frame["sem"], frame["depth_mask"])
        if not split_calls:
            losses = op.train_steps(fc, B, sc, *args, opt=st, n_steps=2).loss.clone()
        else:
            l0 = op.train_steps(fc, B, sc, *args, opt=st, n_steps=1).loss.clone()
            l1 = op.train_steps(fc, B, sc, *(x[:, c["R"]:] for x in args), opt=st, n_steps=1).loss.clone()
            losses = torch.cat([l0, l1])
        torch.cuda.synchronize()
        outs.append((losses, [t.clone() for t in fc + [B]]))
    assert torch.equal(outs[0][0], outs[1][0])
    for x, y in zip(outs[0][1], outs[1][1]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("name,steps", [("cfg2", 20), ("bg_h128_s14", 6), ("h64", 5)])
def test_graph_replay_of_a_bound_frame_is_bit_identical_to_eager_calls(name, steps):
    """step.BoundFrame(graph=True): the frame call (1 + 2 n launches) captured once per step count as a hipGraph and replayed,
    with the optimiser's step count on the device (FusedAdamWState.enable_device_steps: bias corrections from a host-built
    table, the count advanced by the call's first launch) - against the same frames through the eager path with the
    host-side count.  Five frames, the third with a different step count (its own capture): losses, parameters and both
    moments bit for bit; the device count equals the host count."""
    c = cases.build_case(name)
    outs = []
    for graph in (False, True):
        fc, B, sc, b = _to_dev(c)
        op = make_op(c["n"], c["R"], c["S"], c["H"], device=DEV, max_steps=steps)
        st = step.FusedAdamWState(c["n"], c["H"], DEV)
        frame = {k: torch.cat([v.roll(i, dims=1) for i in range(steps)], dim=1).contiguous() for k, v in b.items()}
        bound = op.bind(fc, B, sc, frame["pcs"], frame["z"], frame["gt_depth"], frame["gt_rgb"], frame["sem"], frame["depth_mask"], opt=st,
                        graph=graph)
        assert bound.graph == graph
        losses = []
        for n in (steps, steps, 3, steps, steps, 3, 3):
            res = bound.train_steps(n)
            losses.append(res.loss[:n].clone())
        torch.cuda.synchronize()
        assert st.step == 4 * steps + 9
        if graph:
            assert sorted(bound._graphs) == [3, steps]
            assert int(st.step_counter.sum()) == st.step          # [0] + the last call's not-yet-folded steps
        outs.append(dict(p=[t.clone() for t in fc + [B]], m=st.exp_avg.clone(), v=st.exp_avg_sq.clone(), losses=torch.cat(losses)))
    a, g = outs
    assert bool(torch.isfinite(a["losses"]).all())
    assert torch.equal(a["losses"], g["losses"])
    for x, y in zip(a["p"], g["p"]):
        assert torch.equal(x, y)
    assert torch.equal(a["m"], g["m"]) and torch.equal(a["v"], g["v"])


def test_device_step_count_survives_mixed_host_and_device_calls():
    """A state with the device-resident step count used by calls that take the count from the host in between (the prepared
    path of a multi-rank caller: flag_reduce): both counts stay in step and the trajectory equals the all-host one."""
    c = cases.build_case("tiny")
    outs = []
    for device_steps in (False, True):
        fc, B, sc, b = _to_dev(c)
        op = make_op(c["n"], c["R"], c["S"], c["H"], device=DEV, max_steps=4)
        st = step.FusedAdamWState(c["n"], c["H"], DEV)
        if device_steps:
            st.enable_device_steps()
        frame = {k: torch.cat([v.roll(i, dims=1) for i in range(4)], dim=1).contiguous() for k, v in b.items()}
        args = (frame["pcs"], frame["z"], frame["gt_depth"], frame["gt_rgb"], frame["sem"], frame["depth_mask"])
        losses = []
        for i in range(6):
            kw = dict(flag_reduce=(lambda f: f)) if i % 3 == 1 else {}        # every third frame through prepare / train_steps_prepared
            losses.append(op.train_steps(fc, B, sc, *args, opt=st, n_steps=4 - (i % 2), **kw).loss.clone())
        torch.cuda.synchronize()
        if device_steps:
            assert int(st.step_counter.sum()) == st.step == 21
        outs.append((torch.cat(losses), [t.clone() for t in fc + [B]]))
    assert torch.equal(outs[0][0], outs[1][0])
    for x, y in zip(outs[0][1], outs[1][1]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("config,weights,tuning", [("scannet0024_vmap", "bf16", None), ("stress_256x64", "bf16", None), ("background", "f32", None),
                                                   ("background", "f32", {"ws_flags": 4}), ("background", "bf16", None)])
def test_full_size_properties_of_the_other_baseline_configs(config, weights, tuning):
    """BASELINE configs[3] (50 objects, hidden 32, bf16 weights: the multi-pass kernel), configs[4] (hidden 64, 256 rays per object,
    bf16 weights: multi-round step_main_wp; 32 of its 256 objects = one GPU's share at 8 GPUs) and the background model's full
    batch (1 x 1200 rays x 14, hidden 128: step_main_ws - 200 three-tile rounds, one per workgroup, the automatic plan; with
    tuning.ws_flags = 4 the former plan: 150 workgroups x two two-tile rounds) AT THEIR FULL PER-GPU SIZES, through
    size-independent properties: objects are independent units (permuting them permutes every output), ray order inside an
    object is a pure summation order, two runs are bit-identical - and, full size, the numpy oracle with its ReLU kinks
    accounted for (on bfloat16-rounded weights where the configuration says so)."""
    from conftest import round_bf16
    cfg = synth.CONFIGS[config]
    n, R, S, H = (32 if config == "stress_256x64" else cfg["n_obj"]), cfg["R"], cfg["S"], cfg["H"]
    fc, B, sc = synth.make_params(n, H, scale=cfg["scale"], seed=700)
    batch = synth.make_batch(n, R, S, seed=701)
    c = dict(n=n, R=R, S=S, H=H, fc=fc, B=B, scale=sc, batch=batch)
    op = make_op(n, R, S, H, device=DEV, weights=weights, tuning=tuning)
    s = _run(c, op=op)
    s2 = _run(c, op=op)
    for k in RENDER_KEYS + ["var"] + GRAD_KEYS:
        assert np.array_equal(s[k], s2[k]), k                                     # no atomics, ordered sums: bit-repeatable
    rnd = round_bf16 if weights == "bf16" else (lambda a: a)
    o = vo.training_step([rnd(a) for a in fc], rnd(B), sc, batch, dtype=np.float32, kinks=True)
    assert abs(s["loss"] - o["loss"]) <= 5e-5 * abs(o["loss"])
    for k in RENDER_KEYS + ["var"]:
        assert relerr(s[k], o[k]) < 2e-5, k
    from oracle import vmap_oracle_torch as vt
    _, _, grads_t = vt.CpuTrainer(fc, B, sc, weights_bf16=weights == "bf16").step(batch, update=False)
    _assert_grads_match_aten_port_up_to_kinks(s, o, grads_t, n)                   # the ATen port at 1e-4 (kink list from the oracle)
    if n > 1:
        perm = np.random.default_rng(0).permutation(n)
        cp = dict(c, fc=[a[perm] for a in fc], B=B[perm], scale=sc[perm], batch={k: np.ascontiguousarray(v[perm]) for k, v in batch.items()})
        sp = _run(cp, op=op)
        assert sp["loss"] == pytest.approx(s["loss"], rel=2e-6)
        for k in RENDER_KEYS + GRAD_KEYS:
            assert relerr(sp[k], s[k][perm]) < 2e-6, k
    rperm = np.random.default_rng(1).permutation(R)
    cr = dict(c, batch={k: np.ascontiguousarray(v[:, rperm]) for k, v in batch.items()})
    sr = _run(cr, op=op)
    assert relerr(sr["render_depth"], s["render_depth"][:, rperm]) < 1e-6
    for k in GRAD_KEYS:
        assert relerr(sr[k], s[k]) < 5e-5, k


def test_hipmapper_binds_frames_with_stable_buffers_and_stays_bit_identical():
    """driver.HipMapper.train_frame: the first frame on a set of buffers takes the plain path, the second binds them
    (step.BoundFrame: arguments marshalled once), later frames reuse the binding; a caller that hands over fresh tensors every
    frame stays on the plain path.  Same kernels either way: slabs and losses bit-identical over four frames; a re-stack
    (new object) drops the binding."""
    from vmap_amd.driver import HipMapper
    from vmap_amd.trainer import SimpleConfig, Trainer
    cfg = SimpleConfig(training_device=DEV, n_iter_per_frame=4)
    n, R, S = 3, 24, 10

    def build():
        torch.manual_seed(11)
        m = HipMapper(cfg, device=DEV)
        for _ in range(n):
            m.add_object(Trainer(SimpleConfig(training_device=DEV, hidden_feature_size=32)))
        return m

    keys = ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask")
    frames = [synth.make_batch(n, R * 4, S, seed=50 + i) for i in range(4)]
    a, b = build(), build()
    stable = tuple(torch.from_numpy(frames[0][k]).to(DEV) for k in keys)           # one set of buffers, refilled per frame
    la, lb = [], []
    for i, fr in enumerate(frames):
        for t, k in zip(stable, keys):
            t.copy_(torch.from_numpy(fr[k]).to(DEV))
        la.append(a.train_frame(*stable).loss.clone())
        lb.append(b.train_frame(*(torch.from_numpy(fr[k]).to(DEV) for k in keys)).loss.clone())   # fresh tensors
        assert ("obj" in a._bound) == (i >= 1) and "obj" not in b._bound
    torch.cuda.synchronize()
    assert torch.equal(torch.stack(la), torch.stack(lb)) and torch.equal(a.slab, b.slab)
    assert a.opt.step == b.opt.step == 16
    a.add_object(Trainer(SimpleConfig(training_device=DEV, hidden_feature_size=32)))
    fr = synth.make_batch(n + 1, R * 4, S, seed=60)
    res = a.train_frame(*(torch.from_numpy(fr[k]).to(DEV) for k in keys))
    torch.cuda.synchronize()
    assert "obj" not in a._bound and a.opt.step == 4 and bool(torch.isfinite(res.loss).all())
