"""CPU tier: host-side mirror of the reference interface (modules, update_vmap, write-back, sharding)."""
import numpy as np
import torch

import cases
from conftest import relerr
from oracle import vmap_oracle as vo
from vmap_amd import ensemble, fields, layout, trainer


def _cfg(**kw):
    c = trainer.SimpleConfig(training_device="cpu", **kw)
    c.obj_id = 1
    return c


def test_module_parameter_names_and_order_match_reference():
    m = fields.OccupancyMap(hidden_size=32)
    assert [n for n, _ in m.named_parameters()] == list(layout.FC_NAMES)
    assert [tuple(p.shape) for p in m.parameters()] == [tuple(s) for s in layout.fc_shapes(32)]
    pe = fields.UniDirsEmbed(max_deg=5, scale=2.0)
    assert [n for n, _ in pe.named_parameters()] == ["B_layer.weight"]
    assert [n for n, _ in pe.named_buffers()] == ["frequency_bands", "scale"]
    assert list(pe.state_dict().keys()) == ["scale", "B_layer.weight"] or set(pe.state_dict()) == {"scale", "B_layer.weight"}


def test_module_forward_equals_oracle():
    c = cases.build_case("tiny")
    k = 1
    m = fields.OccupancyMap(hidden_size=32)
    pe = fields.UniDirsEmbed(max_deg=5, scale=float(c["scale"][k]))
    with torch.no_grad():
        for p, a in zip(m.parameters(), c["fc"]):
            p.copy_(torch.from_numpy(a[k]))
        pe.B_layer.weight.copy_(torch.from_numpy(c["B"][k]))
        alpha, color = m(pe(torch.from_numpy(c["batch"]["pcs"][k])))
    emb, _, _ = vo.positional_encoding(c["batch"]["pcs"], c["B"], c["scale"])
    a_ref, c_ref, _ = vo.field_forward(emb, c["fc"])
    assert relerr(alpha.squeeze(-1).numpy(), a_ref[k]) < 1e-5
    assert relerr(color.numpy(), c_ref[k]) < 1e-5


def test_update_vmap_stacks_registers_and_writes_back():
    ts = [trainer.Trainer(_cfg()) for _ in range(4)]
    opt = torch.optim.AdamW([torch.autograd.Variable(torch.tensor(0.0))], lr=1e-3, weight_decay=0.013)
    fmodel, params, buffers = ensemble.update_vmap([t.fc_occ_map for t in ts], opt)
    pmodel, pparams, pbuffers = ensemble.update_vmap([t.pe for t in ts], opt)
    assert len(params) == 14 and all(p.requires_grad and p.is_leaf and p.shape[0] == 4 for p in params)
    assert [tuple(b.shape) for b in pbuffers] == [(4, 6), (4,)]
    assert len(opt.param_groups) == 3 and len(opt.param_groups[1]["params"]) == 14
    for k, t in enumerate(ts):
        for p, q in zip(t.fc_occ_map.parameters(), params):
            assert torch.equal(p.detach(), q[k].detach())
    # re-stack (new object): a NEW group is added, the old stacked tensors keep no gradient -> skipped (utils.py:33)
    ts.append(trainer.Trainer(_cfg()))
    _, params2, _ = ensemble.update_vmap([t.fc_occ_map for t in ts], opt)
    assert len(opt.param_groups) == 4 and params2[0].shape[0] == 5
    with torch.no_grad():
        for p in params2:
            p.add_(1.0)
    ensemble.write_back([t.fc_occ_map for t in ts], params2)
    for k, t in enumerate(ts):
        for p, q in zip(t.fc_occ_map.parameters(), params2):
            assert torch.equal(p.detach(), q[k].detach())
    # functional single-object forward (the reference's vmap strategy remains usable on the stacked state)
    x = torch.randn(7, 3)
    e = pmodel([q[0] for q in pparams], [q[0] for q in pbuffers], x)
    a, c = fmodel([q[0] for q in params2], [], e)
    assert e.shape == (7, 129) and a.shape == (7, 1) and c.shape == (7, 3)


def test_shard_objects_partitions_every_object_once():
    for n in (1, 7, 20, 160):
        for ws in (1, 2, 8):
            owned = [ensemble.shard_objects(n, ws, r) for r in range(ws)]
            assert sorted(sum(owned, [])) == list(range(n))
            assert max(map(len, owned)) - min(map(len, owned)) <= 1


import os
import sys

import pytest

_REF = os.environ.get("VMAP_REFERENCE_ROOT", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(_REF), reason="the reference tree exists in the authoring container only")
def test_state_dicts_interchange_with_the_reference_modules_both_ways():
    """fields.OccupancyMap / fields.UniDirsEmbed against the REAL model.OccupancyMap (model.py:16-49) and
    embedding.UniDirsEmbed (embedding.py:43-80): same parameter / buffer names, order and shapes (what utils.update_vmap
    stacks and what a checkpoint stores), strict load_state_dict in both directions, identical forward values."""
    sys.dont_write_bytecode = True
    if _REF not in sys.path:
        sys.path.insert(0, _REF)
    import importlib
    ref_model = importlib.import_module("model")
    ref_emb = importlib.import_module("embedding")
    assert os.path.dirname(ref_model.__file__) == _REF
    for H in (32, 128):
        torch.manual_seed(H)
        theirs = ref_model.OccupancyMap(layout.EMB1, layout.EMB2, hidden_size=H)
        theirs.apply(ref_model.init_weights)
        ours = fields.OccupancyMap(hidden_size=H)
        assert [n for n, _ in ours.named_parameters()] == [n for n, _ in theirs.named_parameters()] == list(layout.FC_NAMES)
        assert [tuple(p.shape) for p in ours.parameters()] == [tuple(p.shape) for p in theirs.parameters()]
        ours.load_state_dict(theirs.state_dict(), strict=True)              # reference checkpoint -> our module
        emb = torch.randn(7, 5, layout.EMB1 + layout.EMB2)
        with torch.no_grad():
            a_t, c_t = theirs(emb)
            a_o, c_o = ours(emb)
        assert torch.equal(a_t, a_o) and torch.equal(c_t, c_o)
        with torch.no_grad():
            for p in ours.parameters():
                p.mul_(1.25)
        theirs.load_state_dict(ours.state_dict(), strict=True)              # our checkpoint -> reference module
        for p, q in zip(theirs.parameters(), ours.parameters()):
            assert torch.equal(p, q)
    for scale in (2.0, 5.0):
        pe_t = ref_emb.UniDirsEmbed(max_deg=5, scale=scale)
        pe_o = fields.UniDirsEmbed(max_deg=5, scale=scale)
        assert list(pe_t.state_dict().keys()) == list(pe_o.state_dict().keys())
        assert [n for n, _ in pe_t.named_parameters()] == [n for n, _ in pe_o.named_parameters()] == ["B_layer.weight"]
        assert [n for n, _ in pe_t.named_buffers()] == [n for n, _ in pe_o.named_buffers()]
        assert torch.equal(pe_t.B_layer.weight, pe_o.B_layer.weight)       # the icosahedron table (embedding.py:51-73)
        assert torch.equal(pe_t.frequency_bands, pe_o.frequency_bands) and float(pe_t.scale) == float(pe_o.scale)
        with torch.no_grad():
            pe_o.B_layer.weight.add_(0.01)
        pe_t.load_state_dict(pe_o.state_dict(), strict=True)
        pe_o.load_state_dict(pe_t.state_dict(), strict=True)
        x = torch.randn(11, 3)
        with torch.no_grad():
            assert torch.equal(pe_t(x), pe_o(x))


def test_bench_traffic_observation_degrades_without_a_gpu():
    """bench.py observes roofline.traffic itself (two rocprofv3 --pmc passes as subprocesses); where that cannot work - no GPU in this
    container - it must say so and let the line fall back to the committed counter file, never raise."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-tier check")
    val, note = bench.observe_traffic("replica_room0_vmap", "f32", timeout_s=60)
    assert val is None and isinstance(note, str) and note


def test_ray_points_container_shapes_and_slices():
    """step.RayPoints (the sampler's hand-off as rays, ABI v7): validation, the (object, ray) slicing train.py:271-272 applies to the points
    tensor, and the host-side reconstruction of the points (one rounding per operation: vmap.py:452-454)."""
    import numpy as np
    import torch
    from vmap_amd import step
    rng = np.random.default_rng(0)
    o, d = torch.from_numpy(rng.standard_normal((4, 30, 3)).astype(np.float32)), torch.from_numpy(rng.standard_normal((4, 30, 3)).astype(np.float32))
    c = torch.from_numpy(rng.standard_normal((4, 3)).astype(np.float32))
    z = torch.from_numpy(rng.uniform(0.1, 4.0, (4, 30, 10)).astype(np.float32))
    r = step.RayPoints(o, d, c)
    assert r.shape[:2] == (4, 30)
    s = r[:, 10:20]
    assert s.shape[:2] == (4, 10) and s.origins.data_ptr() == o[:, 10:20].data_ptr() and s.centers is c or torch.equal(s.centers, c)
    assert r[1:3].centers.shape == (2, 3)
    p = r.points(z)
    ref = (o.numpy()[:, :, None, :] + d.numpy()[:, :, None, :] * z.numpy()[..., None]) - c.numpy()[:, None, None, :]      # float32 numpy: same roundings
    assert p.shape == (4, 30, 10, 3) and np.array_equal(p.numpy(), ref)
    assert torch.equal(step.RayPoints(o, d).points(z), o.unsqueeze(2) + d.unsqueeze(2) * z.unsqueeze(-1))
    import pytest
    with pytest.raises(ValueError):
        step.RayPoints(o, d[:, :5])
    with pytest.raises(ValueError):
        step.RayPoints(o, d, c[:2])
    with pytest.raises(IndexError):
        r[0]


def test_plain_bench_gpus_n_refuses_without_devices():
    """`python bench.py --gpus 2` without a launcher environment becomes its own launcher (bench.launch_ranks); on a box with fewer
    devices than ranks (this container: none) it must leave with status 2 and say why - never run fewer ranks under the label N."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has two devices: the launch itself is covered by the GPU tier")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "VMAP_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-1000:])
    assert "refusing" in r.stderr and "{" not in r.stdout
