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
