"""Inference query kernel: occupancy + colour of one object's field at arbitrary points, what
the reference's Trainer.eval_points (trainer.py:77-95) evaluates for mesh extraction (render_rays.py:98-122).

CPU tier: the kernel SOURCE on the SIMT simulator against the numpy oracle's forward.  GPU tier: through the C-ABI
(Trainer.eval_points -> vmapstep_query_points) against the module's own PyTorch forward and the oracle."""
import numpy as np
import pytest

from oracle import vmap_oracle as vo
from vmap_amd import synth


def _one_object(seed=3, gain=1.0, H=32):
    fc, B, scale = synth.make_params(2, H, scale=1.7, seed=seed, gain=gain)
    k = 1
    return [a[k] for a in fc], B[k], float(np.asarray(scale).reshape(-1)[k])


def _oracle_query(fc_k, B_k, scale_k, pts):
    emb = vo.positional_encoding(pts[None, :, None, :].astype(np.float32), B_k[None], np.array([scale_k], np.float32))[0]
    alpha, color = vo.field_forward(emb, [np.asarray(a)[None] for a in fc_k])[:2]
    occ = 1.0 / (1.0 + np.exp(-alpha.astype(np.float64)))
    return occ.reshape(-1), color.reshape(-1, 3)


@pytest.mark.parametrize("n_pts,grid", [(1, 1), (127, 1), (128, 2), (1000, 3), (1000, 16)])
def test_sim_query_matches_oracle(n_pts, grid):
    import simlib
    fc_k, B_k, sc = _one_object()
    rng = np.random.default_rng(n_pts)
    pts = rng.uniform(-1.2, 1.2, size=(n_pts, 3)).astype(np.float32)
    occ, rgb = simlib.sim_query(fc_k, B_k, sc, pts, grid=grid)
    o_occ, o_rgb = _oracle_query(fc_k, B_k, sc, pts)
    assert np.isfinite(occ).all() and np.isfinite(rgb).all()
    assert np.abs(occ - o_occ).max() < 2e-5
    assert np.abs(rgb - o_rgb).max() < 2e-5


@pytest.mark.parametrize("H,n_pts", [(64, 300), (128, 200), (256, 70)])
def test_sim_query_generic_widths_match_oracle(H, n_pts):
    import simlib
    fc_k, B_k, sc = _one_object(seed=H, H=H)
    pts = np.random.default_rng(H).uniform(-1.2, 1.2, size=(n_pts, 3)).astype(np.float32)
    occ, rgb = simlib.sim_query(fc_k, B_k, sc, pts, grid=2, H=H)
    o_occ, o_rgb = _oracle_query(fc_k, B_k, sc, pts)
    assert np.isfinite(occ).all() and np.isfinite(rgb).all()
    assert np.abs(occ - o_occ).max() < 3e-5 and np.abs(rgb - o_rgb).max() < 3e-5


def test_sim_query_far_point_uses_accurate_path():
    import simlib
    fc_k, B_k, sc = _one_object(seed=5)
    pts = np.array([[0.1, 0.2, 0.3], [4.0e4, -3.0e4, 2.5e4], [0.0, 0.0, 0.0]], np.float32)
    occ, rgb = simlib.sim_query(fc_k, B_k, sc, pts, grid=1)
    assert np.isfinite(occ).all() and np.isfinite(rgb).all()
    o_occ, o_rgb = _oracle_query(fc_k, B_k, sc, pts[[0, 2]])
    assert np.abs(occ[[0, 2]] - o_occ).max() < 2e-5 and np.abs(rgb[[0, 2]] - o_rgb).max() < 2e-5


@pytest.mark.gpu
def test_gpu_eval_points_matches_modules_and_oracle():
    import torch
    from vmap_amd.trainer import Trainer, SimpleConfig
    torch.manual_seed(0)
    cfg = SimpleConfig(training_device="cuda:0", hidden_feature_size=32)
    tr = Trainer(cfg)
    g = torch.Generator().manual_seed(1)
    for n in (1, 77, 128, 100_003):
        pts = ((torch.rand(n, 3, generator=g) * 2 - 1) * 1.5).cuda()
        out = tr.eval_points(pts)
        assert out is not None
        occ, col = out
        with torch.no_grad():
            a, c = tr.fc_occ_map(tr.pe(pts))
        ref_occ = torch.sigmoid(a.squeeze(-1))
        assert occ.shape == (n,) and col.shape == (n, 3)
        assert (occ - ref_occ).abs().max().item() < 2e-5
        assert (col - c).abs().max().item() < 2e-5
    # non-contiguous points (a column-sliced view) go through the stride arguments
    big = ((torch.rand(513, 5, generator=g) * 2 - 1)).cuda()
    occ, col = tr.eval_points(big[:, 1:4])
    occ2, col2 = tr.eval_points(big[:, 1:4].contiguous())
    assert torch.equal(occ, occ2) and torch.equal(col, col2)
    # numpy oracle on a subset
    fc_k = [p.detach().cpu().numpy() for p in tr.fc_occ_map.parameters()]
    pts = ((torch.rand(500, 3, generator=g) * 2 - 1)).cuda()
    occ, col = tr.eval_points(pts)
    o_occ, o_rgb = _oracle_query(fc_k, tr.pe.B_layer.weight.detach().cpu().numpy(), float(tr.pe.scale), pts.cpu().numpy())
    assert np.abs(occ.cpu().numpy() - o_occ).max() < 2e-5 and np.abs(col.cpu().numpy() - o_rgb).max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("H", [64, 96, 128, 256])
def test_gpu_eval_points_generic_widths(H):
    import torch
    from vmap_amd.trainer import Trainer, SimpleConfig
    torch.manual_seed(H)
    tr = Trainer(SimpleConfig(training_device="cuda:0", hidden_feature_size=H))
    g = torch.Generator().manual_seed(2)
    for n in (5, 1000, 20_011):
        pts = ((torch.rand(n, 3, generator=g) * 2 - 1) * 1.5).cuda()
        occ, col = tr.eval_points(pts)
        with torch.no_grad():
            a, c = tr.fc_occ_map(tr.pe(pts))
        assert (occ - torch.sigmoid(a.squeeze(-1))).abs().max().item() < 3e-5
        assert (col - c).abs().max().item() < 3e-5


@pytest.mark.gpu
def test_gpu_query_grid_throughput_smoke():
    """256^3-scale query stays one launch and finishes; prints the rate (evidence is in profiles/)."""
    import torch
    from vmap_amd.trainer import Trainer, SimpleConfig
    tr = Trainer(SimpleConfig(training_device="cuda:0", hidden_feature_size=32))
    pts = (torch.rand(4_000_000, 3, device="cuda") * 2 - 1)
    tr.eval_points(pts[:1000])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = tr.eval_points(pts); e1.record(); torch.cuda.synchronize()
    assert out is not None and torch.isfinite(out[0]).all()
    print(f"query 4M points: {e0.elapsed_time(e1):.3f} ms")
