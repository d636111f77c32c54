"""GPU tier: ``step_main_s32`` with the SIX-product backward (``tuning.kernel = VMAPSTEP_KERNEL_S32_BWD6``).

The default hidden-32 kernel carries float32 operands as bfloat16 planes and multiplies six plane pairs in the forward (~2^-24,
float32-equivalent) but three in the backward (~2^-16: gradients 6e-6 .. 2e-5 of the reference's, inside the 1e-4 bar).  The
reference is float32 end to end (train.py:64-66, ``AMP = False``); this form is the strictly comparable one: hi.lo + lo.hi +
mid.mid on top of the three in every d-prop and weight-gradient chain.  Same forward code, so loss and renders must be the
default's BITS; gradients are held to the same fixtures at the same bar and must not be further from them than the default's."""
import numpy as np
import pytest
import torch

import cases
from conftest import GRAD_KEYS, RENDER_KEYS, load_golden, relerr
from test_gpu_parity import DEV, TOL, _run, _to_dev
from vmap_amd import _lib, step, synth

pytestmark = pytest.mark.gpu

B6 = {"kernel": _lib.KERNEL_S32_BWD6}
H32_F32_CASES = [n for n, v in cases.CASES.items() if v[3] == 32]


@pytest.mark.parametrize("name", H32_F32_CASES)
def test_six_product_backward_against_the_reference_fixtures(name):
    c = cases.build_case(name)
    g = load_golden(name)
    s6 = _run(c, tuning=B6)
    s3 = _run(c, tuning={"kernel": _lib.KERNEL_AUTO})
    assert s6["loss"] == s3["loss"] and np.array_equal(s6["flags"], s3["flags"])
    for k in RENDER_KEYS + ["var"]:
        assert np.array_equal(s6[k], s3[k]), k                       # the forward is the same code
    rt, gt = TOL.get(name, TOL["default"])
    e6 = {k: relerr(s6[k], g[k]) for k in GRAD_KEYS}
    e3 = {k: relerr(s3[k], g[k]) for k in GRAD_KEYS}
    for k in GRAD_KEYS:
        assert not np.isnan(s6[k]).any(), k
        assert e6[k] < gt, (k, e6[k])
    print(name, "worst gradient tensor vs the reference: six products %.2e, three %.2e" % (max(e6.values()), max(e3.values())))
    if name not in ("saturated", "explode"):                         # (there the reference's own float32 noise floor dominates both)
        assert max(e6.values()) <= max(e3.values()) * 1.25 + 1e-7


def test_six_product_backward_multi_pass_and_repeatability():
    """More objects than compute units' worth of workgroups (the multi-pass instantiation, 50 objects): two launches give the same
    bits; loss against the ATen port; gradients against the DEFAULT kernel's on the same launch plan (same forward, hence the same
    ReLU-kink decisions: what differs is the backward's rounding, a few 1e-5 of a tensor's maximum at most)."""
    from oracle import vmap_oracle_torch as vt
    n, R, S = 50, 120, 10
    fc, B, sc = synth.make_params(n, 32, seed=61)
    batch = synth.make_batch(n, R, S, seed=62)
    c = dict(n=n, R=R, S=S, H=32, fc=fc, B=B, scale=sc, batch=batch)
    op = step.VmapStep(n, R, S, 32, device=DEV, tuning=B6)
    assert op.plan()["kernel"] == "step_main_s32<bwd6>" and op.plan()["workgroups_per_object"] < op.plan()["rounds_per_object"]
    a = _run(c, op=op)
    b = _run(c, op=op)
    for k in GRAD_KEYS + RENDER_KEYS + ["var"]:
        assert np.array_equal(a[k], b[k]), k
    loss_t, _, _ = vt.CpuTrainer(fc, B, sc).step(batch, update=False)
    assert abs(a["loss"] - float(loss_t)) <= 5e-5 * abs(float(loss_t))
    d = _run(c, tuning={"kernel": _lib.KERNEL_AUTO})
    assert d["loss"] == a["loss"]
    worst = max(relerr(a[k], d[k]) for k in GRAD_KEYS)
    print("six-product vs default gradients, 50 objects multi-pass: %.2e" % worst)
    assert worst < 5e-5, worst


def test_six_product_backward_trains_the_reference_frame():
    """The reference's own 20-step frame (fixture cfg2_frame20: train.py:270-326 on the headline shape) through
    vmapstep_train_steps with the fused AdamW: every step's loss within 1e-4 of the reference's."""
    c = cases.build_frame_case("cfg2_frame20")
    g = load_golden("cfg2_frame20")
    n, R, S, H, steps = c["n"], c["R"], c["S"], c["H"], c["n_steps"]
    fr = tuple(torch.from_numpy(c["frame"][k]).to(DEV) for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask"))
    fc = [torch.from_numpy(a).to(DEV) for a in c["fc"]]
    B, sc = torch.from_numpy(c["B"]).to(DEV), torch.from_numpy(c["scale"]).to(DEV)
    op = step.VmapStep(n, R, S, H, device=DEV, max_steps=steps, tuning=B6)
    res = op.train_steps(fc, B, sc, *fr, opt=step.FusedAdamWState(n, H, DEV, lr=1e-3, weight_decay=0.013), n_steps=steps, ray_step=R)
    losses = res.loss.cpu().numpy().astype(np.float64)[:steps]
    err = np.abs(losses - g["losses"]) / np.abs(g["losses"])
    print("per-step loss error vs the reference's frame, six-product backward: max %.2e" % err.max())
    assert err.max() < 1e-4, err


def test_six_product_backward_is_refused_where_it_does_not_exist():
    with pytest.raises(_lib.VmapStepError, match="float32 weights"):
        step.VmapStep(4, 16, 10, 32, device=DEV, weights="bf16", tuning=B6)
    with pytest.raises(_lib.VmapStepError, match="hidden 32"):
        step.VmapStep(1, 16, 14, 128, device=DEV, tuning=B6)


def test_mapper_takes_the_kernel_choice():
    """driver.HipMapper(tuning=...) hands the choice to the object stack's operator ("Precision choices")."""
    from vmap_amd.driver import HipMapper
    from vmap_amd.trainer import SimpleConfig, Trainer
    torch.manual_seed(5)
    n, R, S, steps = 3, 24, 10, 4
    m = HipMapper(SimpleConfig(training_device=DEV, n_iter_per_frame=steps), device=DEV, tuning=B6)
    for _ in range(n):
        m.add_object(Trainer(SimpleConfig(training_device=DEV, hidden_feature_size=32, obj_scale=2.0)))
    fr = synth.make_batch(n, R * steps, S, seed=9)
    batch = tuple(torch.from_numpy(fr[k]).to(DEV) for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask"))
    res = m.train_frame(*batch)
    torch.cuda.synchronize()
    assert m.op.plan()["kernel"] == "step_main_s32<bwd6>"
    assert torch.isfinite(res.loss[:steps]).all()
