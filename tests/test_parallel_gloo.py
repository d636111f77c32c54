"""CPU tier, world_size 2 over gloo: the multi-GPU protocol (object sharding + per-frame flag reduction, shared
background gradient all-reduce) gives the single-process result.  The HIP kernels themselves are covered by the gpu
tier; here the per-shard arithmetic is the oracle, so what is tested is WHAT gets exchanged and when."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from conftest import GRAD_KEYS, relerr
from oracle import vmap_oracle as vo
import bg_twin
from vmap_amd import fields, parallel, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)


def _worker_objects(rank, world, port, ret):
    _init(rank, world, port)
    try:
        c = cases.build_case("drop_depth")       # object 2 has no valid depth -> the switch must reach every rank
        shard = parallel.ObjectShard(c["n"])
        own = shard.owned
        sub = {k: v[own] for k, v in c["batch"].items()}
        _, _, _, _, _, _, local_drop = vo.masks_and_flags(sub["sem"], sub["depth_mask"])
        flags = torch.tensor([[int(x) for x in local_drop] + [0]], dtype=torch.int32)
        shard.reduce_flags(flags)
        o = vo.training_step([a[own] for a in c["fc"]], c["B"][own], c["scale"][own], sub, dtype=np.float32,
                             drop=flags[0, :3].numpy().astype(bool))
        loss = shard.sum_losses(torch.tensor([o["loss"]], dtype=torch.float64))
        ret[rank] = dict(own=own, flags=flags.numpy().copy(), local=local_drop.copy(), loss=float(loss[0]),
                         grads={k: o[k] for k in GRAD_KEYS})
    finally:
        dist.destroy_process_group()


def test_object_sharding_with_flag_reduction_equals_single_process():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_objects, args=(world, port, ret), nprocs=world, join=True)
    c = cases.build_case("drop_depth")
    full = vo.training_step(c["fc"], c["B"], c["scale"], c["batch"], dtype=np.float32)
    assert full["drop"].tolist() == [True, False, False]
    locals_ = [ret[r]["local"].tolist() for r in range(world)]
    assert [True, False, False] in locals_ and [False, False, False] in locals_     # only ONE rank sees it locally
    for r in range(world):
        assert ret[r]["flags"][0, :3].tolist() == [1, 0, 0]
        assert ret[r]["loss"] == pytest.approx(full["loss"], rel=1e-6)
        for k in GRAD_KEYS:
            assert relerr(ret[r]["grads"][k], full[k][ret[r]["own"]]) < 1e-6, k


def _make_bg(seed=3, H=16):
    torch.manual_seed(seed)
    fc = fields.OccupancyMap(hidden_size=H)
    fc.apply(fields.init_weights)
    pe = fields.UniDirsEmbed(max_deg=5, scale=5.0)
    return fc, pe


def _bg_batch(R=64, S=14, steps=3):
    """A whole frame of the background model: `steps` optimisation steps of R rays each ([steps * R, ...])."""
    b = synth.make_batch(1, R * steps, S, seed=9)
    out = {k: torch.from_numpy(v[0]) for k, v in b.items()}
    out["sem"][R:2 * R][::2] = 2            # step 1: every ray of rank 0's shard is 'unknown' ...
    out["sem"][R:2 * R][1::2] = 1           # ... and none of rank 1's: only the per-frame count reduction makes the step consistent
    return out


BG_R, BG_STEPS = 64, 3


def _worker_bg(rank, world, port, ret):
    _init(rank, world, port)
    try:
        fc, pe = _make_bg()
        bg = bg_twin.SharedBackground(fc, pe)
        b = _bg_batch()
        # this rank's rays of every step: rank, rank + world, ... inside each step's R rays
        idx = torch.cat([torch.arange(i * BG_R + rank, (i + 1) * BG_R, world) for i in range(BG_STEPS)])
        loc = {k: v[idx] for k, v in b.items()}
        Rl = BG_R // world
        bg.prepare_frame(loc["sem"], loc["depth_mask"], BG_STEPS)          # ONE count collective per frame
        losses = []
        for i in range(BG_STEPS):
            sl = slice(i * Rl, (i + 1) * Rl)
            losses.append(float(bg.step(loc["pcs"][sl], loc["z"][sl], loc["gt_depth"][sl], loc["gt_rgb"][sl], loc["sem"][sl],
                                        loc["depth_mask"][sl], step_index=i)))   # ONE gradient collective per step
        ret[rank] = dict(losses=losses, params=[p.detach().numpy().copy() for p in bg.params],
                         counts=bg.frame_counts.numpy().copy())
    finally:
        dist.destroy_process_group()


def test_shared_background_allreduce_equals_full_batch_training():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_bg, args=(world, port, ret), nprocs=world, join=True)
    fc, pe = _make_bg()
    bg = bg_twin.SharedBackground(fc, pe)          # world_size 1: plain full-batch training
    b = _bg_batch()
    bg.prepare_frame(b["sem"], b["depth_mask"], BG_STEPS)
    ref_losses = []
    for i in range(BG_STEPS):
        sl = slice(i * BG_R, (i + 1) * BG_R)
        ref_losses.append(float(bg.step(b["pcs"][sl], b["z"][sl], b["gt_depth"][sl], b["gt_rgb"][sl], b["sem"][sl], b["depth_mask"][sl],
                                        step_index=i)))
    for r in range(world):
        assert np.array_equal(ret[r]["counts"], bg.frame_counts.numpy())          # global counts on every rank, for every step
        assert ret[r]["losses"] == pytest.approx(ref_losses, rel=1e-5)
        for p, q in zip(ret[r]["params"], bg.params):
            d = np.abs(p - q.detach().numpy())
            assert d.max() < 3.2e-3 and np.median(d) < 1e-6        # Adam sign flips on ~0 gradients only
    for p, q in zip(ret[0]["params"], ret[1]["params"]):
        assert np.array_equal(p, q)                                 # replicas stay bit-identical
