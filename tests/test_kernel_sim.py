"""CPU tier: the HIP kernel SOURCE (vmap_amd/csrc/step_kernels.h) executed lane-by-lane on the SIMT executor
of tests/sim and compared with the reference fixtures.  This validates the MFMA operand/accumulator index maps,
LDS addressing, barriers-as-written and the fwd/bwd math without a GPU; the `-m gpu` tests repeat the same
comparisons on the real device through the C ABI."""
import numpy as np
import pytest

import cases
import simlib
from conftest import GRAD_KEYS, RENDER_KEYS, load_golden, relerr
from oracle import vmap_oracle as vo

TOL = {"default": (2e-5, 1e-4), "saturated": (2e-3, 2e-3)}


@pytest.mark.parametrize("name", ["tiny", "ragged", "drop_depth", "drop_opacity", "drop_colour", "saturated", "exact_hit"])
def test_sim_kernel_matches_reference(name):
    c = cases.build_case(name)
    g = load_golden(name)
    s = simlib.sim_step(c)
    rt, gt = TOL.get(name, TOL["default"])
    assert abs(s["loss"] - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for k in RENDER_KEYS:
        assert relerr(s[k], g[k]) < rt, k
    for k in GRAD_KEYS:
        assert relerr(s[k], g[k]) < gt, k
    o = vo.training_step(c["fc"], c["B"], c["scale"], c["batch"], dtype=np.float32)
    assert s["flags"][:3].tolist() == [int(x) for x in o["drop"]]
    assert s["flags"][3] == int(o["explode"])


@pytest.mark.parametrize("nw,G", [(1, None), (2, 5), (3, 7)])
def test_sim_pass_loop_and_group_sizes(nw, G):
    """A workgroup covering several ray groups (NW < NG) and odd group sizes give the same result."""
    c = cases.build_case("ragged")
    g = load_golden("ragged")
    s = simlib.sim_step(c, NW=nw, G=G)
    for k in RENDER_KEYS + GRAD_KEYS:
        assert relerr(s[k], g[k]) < 1e-4, k


def test_sim_forward_only_has_same_renders():
    c = cases.build_case("tiny")
    g = load_golden("tiny")
    s = simlib.sim_step(c, bwd=False)
    assert abs(s["loss"] - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for k in RENDER_KEYS:
        assert relerr(s[k], g[k]) < 2e-5, k


def test_sim_far_point_takes_library_sincos_path():
    c = cases.build_case("tiny")
    c["batch"]["pcs"][1, 3, 4, :] = [3.0e5, -2.0e5, 1.0e5]
    o = vo.training_step(c["fc"], c["B"], c["scale"], c["batch"], dtype=np.float32)
    s = simlib.sim_step(c)
    for k in RENDER_KEYS + GRAD_KEYS:
        assert relerr(s[k], o[k]) < 1e-4, k


def test_sim_fused_adamw_matches_oracle_update():
    c = cases.build_case("tiny")
    o = vo.training_step(c["fc"], c["B"], c["scale"], c["batch"], dtype=np.float32)
    n = c["n"]
    flat = np.concatenate([a.reshape(n, -1) for a in c["fc"]] + [c["B"].reshape(n, -1)], axis=1).astype(np.float32)
    P = flat.shape[1]
    PP = (P + 63) // 64 * 64
    state = dict(p=flat.copy(), m=np.zeros((n, PP), np.float32), v=np.zeros((n, PP), np.float32), step=1)
    s = simlib.sim_step(c, adam=state)
    gflat = s["grads_flat"]
    p_ref, m_ref, v_ref = vo.adamw_update(flat, gflat, np.zeros_like(flat), np.zeros_like(flat), 1)
    assert relerr(state["p"], p_ref) < 1e-6
    assert relerr(state["m"][:, :P], m_ref) < 1e-6
    assert relerr(state["v"][:, :P], v_ref) < 1e-6
    assert relerr(gflat, np.concatenate([o[k].reshape(n, -1) for k in GRAD_KEYS], axis=1)) < 1e-4


@pytest.mark.slow
def test_sim_kernel_full_headline_config():
    """BASELINE configs[1] at full size (20 x 120 x 10, H=32): ~45 s on the simulator."""
    c = cases.build_case("cfg2")
    g = load_golden("cfg2")
    s = simlib.sim_step(c)
    assert abs(s["loss"] - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for k in RENDER_KEYS:
        assert relerr(s[k], g[k]) < 2e-5, k
    for k in GRAD_KEYS:
        assert relerr(s[k], g[k]) < 1e-4, k


@pytest.mark.parametrize("n,R,S,seed", [(1, 120, 10, 2), (2, 40, 3, 5), (3, 9, 14, 4), (2, 5, 20, 7)])
def test_sim_kernel_matches_torch_port_on_seeded_shapes(n, R, S, seed):
    """Same seeds as the GPU tier: the FMA-based PyTorch port is the tight comparator (catches e.g. cancellation in
    the compositing backward that the fixtures' shapes do not excite)."""
    from oracle import vmap_oracle_torch as vt
    from vmap_amd import synth
    fc, B, sc = synth.make_params(n, 32, seed=100 + seed)
    batch = synth.make_batch(n, R, S, seed=200 + seed)
    loss_t, rend_t, grads_t = vt.CpuTrainer(fc, B, sc).step(batch, update=False)
    s = simlib.sim_step(fc, B, sc, batch)
    assert abs(s["loss"] - float(loss_t)) <= 2e-5 * abs(float(loss_t))
    for k in RENDER_KEYS:
        assert relerr(s[k], rend_t[k].detach().numpy()) < 2e-5, k
    for k, g in zip(GRAD_KEYS, grads_t):
        assert relerr(s[k], g.numpy()) < 2e-5, k


@pytest.mark.parametrize("name,nw", [("h64", 0), ("h64", 2), ("bg_h128_s14", 0)])
def test_sim_generic_width_kernel_matches_reference(name, nw):
    """step_main_gen (hidden = 64 / 128, global-memory activations; nw=2: multi-pass accumulate) vs the fixtures."""
    c = cases.build_case(name)
    g = load_golden(name)
    s = simlib.sim_step(c, NW=nw)
    assert abs(s["loss"] - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for k in RENDER_KEYS:
        assert relerr(s[k], g[k]) < 2e-5, k
    for k in GRAD_KEYS:
        assert relerr(s[k], g[k]) < 1e-4, k


@pytest.mark.parametrize("nw", [0, 5])
def test_sim_wide_kernel_matches_reference(nw):
    """step_main_wide (hidden 128: one 32-point tile per workgroup, output blocks split over the four waves;
    nw=5: five passes per workgroup accumulate into its partial buffer) vs the background-shaped fixture."""
    c = cases.build_case("bg_h128_s14")
    g = load_golden("bg_h128_s14")
    s = simlib.sim_step(c, NW=nw, wide=True)
    assert abs(s["loss"] - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for k in RENDER_KEYS:
        assert relerr(s[k], g[k]) < 2e-5, k
    for k in GRAD_KEYS:
        assert relerr(s[k], g[k]) < 1e-4, k
    if nw == 0:
        o = simlib.sim_step(c, NW=0)                  # the general kernel on the same case
        for k in GRAD_KEYS:
            assert relerr(s[k], o[k]) < 2e-5, k


@pytest.mark.parametrize("name,nw,kernel", [("bg_h128_s14", 0, 3), ("bg_h128_s14", 3, 3), ("h64", 0, 3), ("h64", 2, 3),
                                            ("bg_h128_s14", 0, 4), ("bg_h128_s14", 3, 4), ("h64", 0, 4), ("h64", 2, 4)])
def test_sim_ws_kernel_matches_reference(name, nw, kernel):
    """step_main_ws (kernel 3) / step_main_wp (4): hidden 128 / 64 on the bf16 matrix pipe with split operands, a round of
    two 32-point tiles per workgroup, one wave / a pair of waves per output block; nw > 0: several rounds per workgroup
    add into its partial gradients.  Against the fixtures generated by the reference."""
    c = cases.build_case(name)
    g = load_golden(name)
    s = simlib.sim_step(c, NW=nw, wide=kernel)
    assert abs(s["loss"] - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for k in RENDER_KEYS:
        assert relerr(s[k], g[k]) < 2e-5, k
    for k in GRAD_KEYS:
        assert relerr(s[k], g[k]) < 1e-4, k


def test_sim_ws_kernel_bf16_weights():
    """step_main_ws with weight_dtype = bf16 (one weight plane) == the float32 oracle on the rounded weights."""
    from conftest import round_bf16
    c = cases.build_case("bg_h128_s14")
    fc_r = [round_bf16(a) for a in c["fc"]]
    B_r = round_bf16(c["B"])
    s = simlib.sim_step(c, weights_bf16=1, wide=4)
    o = vo.training_step(fc_r, B_r, c["scale"], c["batch"], dtype=np.float32)
    assert abs(s["loss"] - o["loss"]) <= 2e-5 * abs(o["loss"])
    for k in GRAD_KEYS:
        assert relerr(s[k], o[k]) < 1e-4, k


def test_sim_bf16_weights_equal_oracle_on_rounded_weights():
    """weight_dtype = bf16 (BASELINE configs[3]/[4]): the kernels compute from the bfloat16-rounded parameter image with
    fp32 products/sums == the fp32 oracle evaluated on the rounded weights (last bullet)."""
    from conftest import round_bf16
    c = cases.build_case("tiny")
    fc_r = [round_bf16(a) for a in c["fc"]]
    B_r = round_bf16(c["B"])
    assert any(not np.array_equal(a, b) for a, b in zip(fc_r, c["fc"]))
    s = simlib.sim_step(c, weights_bf16=1)
    from oracle import vmap_oracle_torch as vt
    loss_t, rend_t, grads_t = vt.CpuTrainer(fc_r, B_r, c["scale"]).step(c["batch"], update=False)
    assert abs(s["loss"] - float(loss_t)) <= 2e-5 * abs(float(loss_t))
    for k in RENDER_KEYS:
        assert relerr(s[k], rend_t[k].detach().numpy()) < 2e-5, k
    for k, g in zip(GRAD_KEYS, grads_t):
        assert relerr(s[k], g.numpy()) < 1e-4, k
    # and it is NOT the fp32-weight result
    assert relerr(s["render_depth"], load_golden("tiny")["render_depth"]) > 1e-4


# ---- the split-bf16 kernels (split_kernels.h: step_prep_s32 / step_main_s32 / step_finalize_s32) ----------------------
@pytest.mark.parametrize("name", ["tiny", "ragged", "drop_depth", "drop_opacity", "drop_colour", "saturated", "exact_hit"])
def test_sim_split_kernel_matches_reference(name):
    """bf16 matrix instruction with split operands (6 products forward, 3 backward), owner-lane encoding with the
    double-angle octave recurrence, transposing LDS reads, bias columns: same fixtures, same tolerances."""
    c = cases.build_case(name)
    g = load_golden(name)
    s = simlib.sim_step(c, split=True)
    rt, gt = TOL.get(name, TOL["default"])
    assert abs(s["loss"] - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for k in RENDER_KEYS:
        assert relerr(s[k], g[k]) < rt, k
    for k in GRAD_KEYS:
        assert not np.isnan(s[k]).any(), k
        assert relerr(s[k], g[k]) < gt, k
    o = vo.training_step(c["fc"], c["B"], c["scale"], c["batch"], dtype=np.float32)
    assert s["flags"][:3].tolist() == [int(x) for x in o["drop"]]


@pytest.mark.parametrize("name,kw", [("tiny", {}), ("ragged", {}), ("ragged", dict(NW=2, G=5)), ("exact_hit", {})])
def test_sim_split_six_product_backward(name, kw):
    """tuning.kernel = VMAPSTEP_KERNEL_S32_BWD6: hi.lo + lo.hi + mid.mid on top of the default's three products in the d-prop and
    weight-gradient chains (lo planes of deltas, layer inputs and W^T carried through the backward).  Same fixtures and bars - and,
    measured against the oracle evaluated in float64 on the same float32 inputs, gradients several times closer than the default's."""
    c = cases.build_case(name)
    g = load_golden(name)
    s6 = simlib.sim_step(c, split=2, **kw)
    s3 = simlib.sim_step(c, split=1, **kw)
    assert s6["loss"] == s3["loss"]                       # the forward is the same code
    for k in RENDER_KEYS:
        assert np.array_equal(s6[k], s3[k]), k
    for k in GRAD_KEYS:
        assert relerr(s6[k], g[k]) < 1e-4, k
    o = vo.training_step(c["fc"], c["B"], c["scale"], c["batch"], dtype=np.float64)
    e6 = max(relerr(s6[k], o[k]) for k in GRAD_KEYS)
    e3 = max(relerr(s3[k], o[k]) for k in GRAD_KEYS)
    print(name, kw, "max gradient error vs the float64 oracle: six products %.2e, three %.2e" % (e6, e3))
    assert e6 <= e3 * 1.05


@pytest.mark.parametrize("nw,G", [(1, None), (2, 5), (3, 7)])
def test_sim_split_pass_loop_and_group_sizes(nw, G):
    """Several passes per workgroup (the image stays valid across passes: the staging tiles alias the idle transpose tiles)."""
    c = cases.build_case("ragged")
    g = load_golden("ragged")
    s = simlib.sim_step(c, NW=nw, G=G, split=True)
    for k in RENDER_KEYS + GRAD_KEYS:
        assert relerr(s[k], g[k]) < 1e-4, k


def test_sim_split_forward_only_and_far_point():
    c = cases.build_case("tiny")
    g = load_golden("tiny")
    s = simlib.sim_step(c, bwd=False, split=True)
    for k in RENDER_KEYS:
        assert relerr(s[k], g[k]) < 2e-5, k
    c["batch"]["pcs"][1, 3, 4, :] = [3.0e5, -2.0e5, 1.0e5]          # library sincos for octave 0, recurrence above it
    o = vo.training_step(c["fc"], c["B"], c["scale"], c["batch"], dtype=np.float32)
    s = simlib.sim_step(c, split=True)
    for k in RENDER_KEYS + GRAD_KEYS:
        assert relerr(s[k], o[k]) < 1e-4, k


def test_sim_split_bf16_weights_equal_oracle_on_rounded_weights():
    from conftest import round_bf16
    c = cases.build_case("tiny")
    fc_r = [round_bf16(a) for a in c["fc"]]
    B_r = round_bf16(c["B"])
    o = vo.training_step(fc_r, B_r, c["scale"], c["batch"], dtype=np.float32)
    s = simlib.sim_step(c, split=True, weights_bf16=1)
    for k in RENDER_KEYS:
        assert relerr(s[k], o[k]) < 2e-5, k
    for k in GRAD_KEYS:
        assert relerr(s[k], o[k]) < 2e-2, k          # numpy oracle: ReLU-kink noise bound (the ATen port is the tight comparator on the GPU tier)


def test_sim_split_fused_adamw_and_maintained_image():
    """step_finalize_s32: AdamW == oracle update; the planes it rewrites == the planes a fresh pack of the updated
    parameters produces (second step from either gives the same loss)."""
    c = cases.build_case("tiny")
    n = c["n"]
    flat = np.concatenate([a.reshape(n, -1) for a in c["fc"]] + [c["B"].reshape(n, -1)], axis=1).astype(np.float32)
    P = flat.shape[1]
    PP = (P + 63) // 64 * 64
    state = dict(p=flat.copy(), m=np.zeros((n, PP), np.float32), v=np.zeros((n, PP), np.float32), step=1)
    s = simlib.sim_step(c, adam=state, split=True)
    p_ref, m_ref, v_ref = vo.adamw_update(flat, s["grads_flat"], np.zeros_like(flat), np.zeros_like(flat), 1)
    assert relerr(state["p"], p_ref) < 1e-6
    assert relerr(state["m"][:, :P], m_ref) < 1e-6
    assert relerr(state["v"][:, :P], v_ref) < 1e-6


def test_sim_ws_fused_adamw_matches_oracle_update():
    """step_finalize_ws (four threads per parameter quad sum the partial-gradient rows, AdamW, W / W^T image rewrite):
    the update == the oracle's AdamW on the same gradients."""
    c = cases.build_case("bg_h128_s14")
    n = c["n"]
    flat = np.concatenate([a.reshape(n, -1) for a in c["fc"]] + [c["B"].reshape(n, -1)], axis=1).astype(np.float32)
    P = flat.shape[1]
    PP = (P + 63) // 64 * 64
    state = dict(p=flat.copy(), m=np.zeros((n, PP), np.float32), v=np.zeros((n, PP), np.float32), step=1)
    s = simlib.sim_step(c, adam=state, wide=4, NW=5)
    p_ref, m_ref, v_ref = vo.adamw_update(flat, s["grads_flat"], np.zeros_like(flat), np.zeros_like(flat), 1)
    assert relerr(state["p"], p_ref) < 1e-6
    assert relerr(state["m"][:, :P], m_ref) < 1e-6
    assert relerr(state["v"][:, :P], v_ref) < 1e-6


@pytest.mark.parametrize("name,kw", [("h64", dict(wide=4, NW=2)), ("h64", dict(wide=4, NW=1)), ("bg_h128_s14", dict(wide=3, NW=3)),
                                     ("bg_h128_s14", dict(wide=4, NW=5)), ("imap_h256", dict(wide=3, G=2))])
def test_sim_ws_finalize_forms_give_the_same_bits(name, kw):
    """step_finalize_ws with a thread per quad AND row group (few blocks, many rows) and with one thread per quad walking all eight row
    groups (what the library launches for many blocks / few rows - 256 objects x 2 rows): the same ordered sums, so the same gradients,
    loss and AdamW update, bit for bit (also with fewer rows than row groups: empty groups contribute exact zeros)."""
    c = cases.build_case(name)
    n = c["n"]
    flat = np.concatenate([a.reshape(n, -1) for a in c["fc"]] + [c["B"].reshape(n, -1)], axis=1).astype(np.float32)
    PP = (flat.shape[1] + 63) // 64 * 64
    out = []
    for form in (0, 1):
        state = dict(p=flat.copy(), m=np.zeros((n, PP), np.float32), v=np.zeros((n, PP), np.float32), step=1)
        s = simlib.sim_step(c, adam=state, finalize_form=form, **kw)
        out.append((s, state))
    (s0, st0), (s1, st1) = out
    assert s0["loss"] == s1["loss"]
    assert np.array_equal(s0["grads_flat"], s1["grads_flat"])
    for k in ("p", "m", "v"):
        assert np.array_equal(st0[k], st1[k]), k


def test_sim_ws_finalize_xcd_affine_block_map_with_nine_objects():
    """From eight objects on step_finalize_ws deals its blocks to the objects in groups of eight (block b -> object 8 * group + b % 8: all blocks
    of an object on one XCD; a ninth object leaves seven of the second group's eight slots empty).  Both thread forms, against the ATen port."""
    from oracle import vmap_oracle_torch as vt
    from vmap_amd import synth
    n, R, S, H = 9, 6, 10, 64
    fc, B, sc = synth.make_params(n, H, seed=91)
    batch = synth.make_batch(n, R, S, seed=92)
    loss_t, rend_t, grads_t = vt.CpuTrainer(fc, B, sc).step(batch, update=False)
    ref = np.concatenate([g.numpy().reshape(n, -1) for g in grads_t], axis=1)
    outs = []
    for form in (0, 1):
        s = simlib.sim_step(fc, B, sc, batch, wide=4, NW=1, finalize_form=form)
        assert abs(s["loss"] - float(loss_t)) <= 2e-5 * abs(float(loss_t))
        assert relerr(s["grads_flat"][:, :ref.shape[1]], ref) < 1e-4
        outs.append(s)
    assert outs[0]["loss"] == outs[1]["loss"] and np.array_equal(outs[0]["grads_flat"], outs[1]["grads_flat"])


@pytest.mark.parametrize("name,kw", [("ragged", dict(split=True, NW=2, G=5)), ("ragged", dict(split=True, NW=1)),
                                     ("h64", dict(wide=4, NW=2)), ("h64", dict(wide=3, NW=2)), ("bg_h128_s14", dict(wide=3, NW=3))])
def test_sim_bf16_weights_in_the_multi_pass_and_multi_round_forms(name, kw):
    """weight_dtype = bf16 in the instantiations BASELINE configs[3] / [4] actually run: step_main_s32<., MULTI, ., W3 = false>
    (several passes per workgroup: the staging tiles overlay the planes the one-plane form does not have), step_main_wp<2> /
    step_main_ws with several rounds per workgroup.  Comparators: the reference evaluated on bfloat16-rounded parameters
    (fixture <name>_bf16.npz) where it exists, else the ATen port on the rounded weights."""
    from conftest import round_bf16
    from oracle import vmap_oracle_torch as vt
    c = cases.build_case(name)
    fc_r, B_r = [round_bf16(a) for a in c["fc"]], round_bf16(c["B"])
    s = simlib.sim_step(c, weights_bf16=1, **kw)
    if name in cases.BF16_CASES:
        g = load_golden(name + "_bf16")
        ref = {k: g[k] for k in RENDER_KEYS + GRAD_KEYS}
        loss = float(g["loss"])
    else:
        loss_t, rend_t, grads_t = vt.CpuTrainer(fc_r, B_r, c["scale"]).step(c["batch"], update=False)
        ref = {k: rend_t[k].detach().numpy() for k in RENDER_KEYS}
        ref.update({k: gr.numpy() for k, gr in zip(GRAD_KEYS, grads_t)})
        loss = float(loss_t)
    assert abs(s["loss"] - loss) <= 2e-5 * abs(loss)
    for k in RENDER_KEYS:
        assert relerr(s[k], ref[k]) < 2e-5, k
    for k in GRAD_KEYS:
        assert not np.isnan(s[k]).any(), k
        assert relerr(s[k], ref[k]) < 1e-4, k


@pytest.mark.parametrize("name,G,nw,bf16", [("bg_h128_s14", 2, 0, 0), ("bg_h128_s14", 2, 5, 0), ("bg_h128_s14", 1, 0, 1), ("h64", 3, 0, 0), ("h64", 2, 4, 0)])
def test_sim_ws_single_tile_rounds_match_reference(name, G, nw, bf16):
    """step_main_ws<., ., ., ., NT = 1>: rounds of ONE 32-point tile (ray groups of G rays with G S <= 32: 2 x 14, 3 x 10 points;
    fewer rays than fit; several rounds per workgroup adding into its row) - the form the launch plan picks when every tile
    gets a compute unit of its own.  Same fixtures as the two-tile form (bf16: the reference on the rounded weights)."""
    c = cases.build_case(name)
    g = load_golden(name + ("_bf16" if bf16 else ""))
    s = simlib.sim_step(c, wide=3, G=G, NW=nw, weights_bf16=bf16)
    assert abs(s["loss"] - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for k in RENDER_KEYS:
        assert relerr(s[k], g[k]) < 2e-5, k
    for k in GRAD_KEYS:
        assert not np.isnan(s[k]).any(), k
        assert relerr(s[k], g[k]) < 1e-4, k


@pytest.mark.parametrize("G,nw,bf16", [(6, 0, 0), (6, 3, 0), (5, 0, 1), (5, 4, 0)])
def test_sim_ws_three_tile_rounds_match_reference(G, nw, bf16):
    """step_main_ws<4, ., ., ., NT = 3>: rounds of THREE 32-point tiles (hidden 128; ray groups of G rays with 64 < G S <= 96:
    6 x 14 = 84 points, 5 x 14 = 70 with a partly filled third tile and a ragged last round; several rounds per workgroup adding
    into its row) - the form the launch plan picks when two-tile rounds outnumber the compute units and three-tile rounds do
    not (the 1200-ray background batch).  Its LDS map differs from the two-tile form's (heads' partial sums over the layer-input
    images, second-group F images in the scratch); same fixtures."""
    c = cases.build_case("bg_h128_s14")
    g = load_golden("bg_h128_s14" + ("_bf16" if bf16 else ""))
    s = simlib.sim_step(c, wide=3, G=G, NW=nw, weights_bf16=bf16)
    assert abs(s["loss"] - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for k in RENDER_KEYS:
        assert relerr(s[k], g[k]) < 2e-5, k
    for k in GRAD_KEYS:
        assert not np.isnan(s[k]).any(), k
        assert relerr(s[k], g[k]) < 1e-4, k


@pytest.mark.parametrize("G,nw,bf16", [(2, 0, 0), (2, 7, 0), (1, 0, 1)])
def test_sim_ws_hidden256_matches_reference(G, nw, bf16):
    """step_main_ws<8>: hidden 256 (the iMAP field, BASELINE configs[0]) on the bf16 matrix pipe with split operands - EIGHT waves,
    one output block each, single-tile rounds (G S <= 32); one round per workgroup (the single-round specialisation the launch plan
    uses) and several rounds adding into the row; bfloat16 run-time weights against the float32 oracle on the rounded weights."""
    c = cases.build_case("imap_h256")
    if bf16:
        from conftest import round_bf16
        fc_r = [round_bf16(a) for a in c["fc"]]
        B_r = round_bf16(c["B"])
        o = vo.training_step(fc_r, B_r, c["scale"], c["batch"], dtype=np.float32)
        g = {k: o[k] for k in RENDER_KEYS + GRAD_KEYS}
        g["loss"] = o["loss"]
    else:
        g = load_golden("imap_h256")
    s = simlib.sim_step(c, wide=3, G=G, NW=nw, weights_bf16=bf16)
    assert abs(s["loss"] - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for k in RENDER_KEYS:
        assert relerr(s[k], g[k]) < 2e-5, k
    for k in GRAD_KEYS:
        assert not np.isnan(s[k]).any(), k
        assert relerr(s[k], g[k]) < 1e-4, k


@pytest.mark.parametrize("H,wide,split,n,R,S", [(32, False, True, 3, 13, 10), (32, False, False, 2, 7, 10), (64, 4, False, 2, 9, 10), (128, 3, False, 1, 5, 14),
                                                (256, 3, False, 1, 2, 14), (96, False, False, 1, 5, 10)])
def test_ray_handoff_equals_points_on_the_simulator(H, wide, split, n, R, S):
    """ABI v7 on the CPU tier: the kernels' own source, executed lane by lane, given (origin, direction, z) + centres and NO points tensor
    (it is poisoned with NaN) - step_main_s32 / _h32 / _wp / _ws / _ws<8> / _gen rebuild
    (o + d z) - c with one rounding per operation and return the bits of the run on the points tensor formed the same way in numpy."""
    from vmap_amd import synth
    fc, B, sc = synth.make_params(n, H, seed=50 + H)
    batch = synth.make_batch(n, R, S, seed=51 + H)
    rng = np.random.default_rng(52 + H)
    o = rng.uniform(-1, 1, (n, R, 3)).astype(np.float32)
    d = rng.uniform(-1, 1, (n, R, 3)).astype(np.float32)
    c = rng.uniform(-0.5, 0.5, (n, 3)).astype(np.float32)
    z = batch["z"]
    batch = dict(batch, pcs=((o[:, :, None, :] + d[:, :, None, :] * z[..., None]) - c[:, None, None, :]).astype(np.float32))   # float32 numpy: one rounding per operation
    G = 2 if H == 256 else None                    # hidden 256: single-tile rounds (two 14-sample rays)
    a = simlib.sim_step(fc, B, sc, batch, wide=wide, split=split, G=G)
    b = simlib.sim_step(fc, B, sc, batch, wide=wide, split=split, rays=(o, d, c), G=G)
    assert np.isfinite(a["loss"]) and a["loss"] == b["loss"]
    for k in ("render_depth", "render_color", "opacity", "var", "grads_flat"):
        assert np.array_equal(a[k], b[k]), k
    b0 = simlib.sim_step(fc, B, sc, dict(batch, pcs=((o[:, :, None, :] + d[:, :, None, :] * z[..., None])).astype(np.float32)), wide=wide, split=split, rays=(o, d, None), G=G)
    a0 = simlib.sim_step(fc, B, sc, dict(batch, pcs=((o[:, :, None, :] + d[:, :, None, :] * z[..., None])).astype(np.float32)), wide=wide, split=split, G=G)
    assert np.array_equal(a0["grads_flat"], b0["grads_flat"]) and a0["loss"] == b0["loss"]          # centres omitted = zeros


@pytest.mark.parametrize("H", [64, 128, 256])
def test_block_native_gradient_row_holds_every_parameter_exactly_once(H):
    """The row of partial gradients of step_main_ws / _wp is a sequence of 32 x 32 blocks as they leave the matrix pipe + the small
    vectors (RowWs<NB>, wsplit_kernels.h); step_finalize_ws finds the parameter behind a row element through the table step_prep_ws
    writes from ws_row_source.  That table must be a bijection between the live row elements and the flat parameters - a parameter
    missing from it would never be trained, one named twice would take two gradients."""
    import ctypes
    from vmap_amd import layout
    L = simlib.lib()
    L.vmsim_row_table.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    PR = L.vmsim_row_table(H, None)
    tab = np.full(PR, -7, dtype=np.int32)
    assert L.vmsim_row_table(H, tab.ctypes.data_as(ctypes.POINTER(ctypes.c_int))) == PR
    P = layout.param_count(H)
    live = tab[tab >= 0]
    assert PR % 64 == 0 and tab.min() == -1
    assert live.size == P and np.array_equal(np.sort(live), np.arange(P))
    # the blocks come first, 1024 floats each; the padding columns of the encoding blocks are the only holes in front of the small vectors
    nb = H // 32
    blocks = nb * (4 * nb + 8)
    assert PR - 64 < blocks * 1024 + 6 * H + 67 <= PR
    assert (tab[blocks * 1024:blocks * 1024 + 6 * H + 67] >= 0).all() and (tab[blocks * 1024 + 6 * H + 67:] == -1).all()
