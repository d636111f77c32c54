"""GPU tier: the PRODUCT's multi-rank code with world_size 2 on the one GPU of the test box.

RCCL refuses two ranks on one device, so the process group is gloo (device tensors travel through the host); everything
else is what an 8-GPU run executes: ``parallel.ObjectShard`` (round-robin object shards, per-frame MAX of the empty-mask
switches between vmapstep_prepare and vmapstep_train_steps_prepared) and ``parallel.SharedBackgroundHip`` (ray-sharded
replicas of the background model: one count all-reduce per frame, ONE [gradients | loss terms] all-reduce per step between
vmapstep_fwd_bwd_prepared and vmapstep_adamw_apply) - both through the C ABI on cuda:0, compared with the world-1 run of the
same classes in this process.  (tests/test_parallel_gloo.py covers the protocol with a PyTorch twin on the CPU tier.)
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import cases
from vmap_amd import synth

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
STEPS = 4
KEYS = ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask")
BG = dict(H=128, R=64, S=14, scale=5.0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _object_frame():
    """4 objects x 24 rays x 4 steps; object 2 has no valid depth in ANY step (cases 'drop_depth'): with round-robin shards
    only rank 0 owns it, so rank 1 learns about the batch-wide switch (render_rays.py:68-73) through the collective only."""
    c = cases.build_case("drop_depth")
    frame = {k: np.ascontiguousarray(np.concatenate([np.roll(v, i, axis=1) for i in range(STEPS)], axis=1)) for k, v in c["batch"].items()}
    return c, frame


def _background():
    from vmap_amd import fields
    torch.manual_seed(4)
    fc = fields.OccupancyMap(hidden_size=BG["H"])
    fc.apply(fields.init_weights)
    pe = fields.UniDirsEmbed(max_deg=5, scale=BG["scale"])
    b = synth.make_batch(1, BG["R"] * STEPS, BG["S"], seed=9)
    fr = {k: v[0].copy() for k, v in b.items()}
    R = BG["R"]
    fr["sem"][R:2 * R][::2] = 2       # step 1: every ray of rank 0's shard is 'unknown', none of rank 1's: only the per-frame
    fr["sem"][R:2 * R][1::2] = 1      # count reduction gives both ranks the same (global) normalisers and switches
    return fc, pe, fr


def _train_objects(shard, c, frame, dev):
    """This rank's object shard through VmapStep.train_steps with the flag reduction; workgroups_per_object is pinned so that
    an object's arithmetic does not depend on how many objects share the GPU (bit-identical to the world-1 run)."""
    from vmap_amd import step
    own = shard.owned
    fc = [torch.from_numpy(a[own]).to(dev) for a in c["fc"]]
    B = torch.from_numpy(c["B"][own]).to(dev)
    sc = torch.from_numpy(c["scale"][own]).to(dev)
    fr = {k: torch.from_numpy(v[own]).to(dev) for k, v in frame.items()}
    op = step.VmapStep(len(own), c["R"], c["S"], c["H"], device=dev, max_steps=STEPS, tuning={"workgroups_per_object": 2})
    opt = step.FusedAdamWState(len(own), c["H"], dev)
    seen = {}

    def reduce(flags):
        seen["local"] = flags.clone()
        return shard.reduce_flags(flags)

    res = op.train_steps(fc, B, sc, *(fr[k] for k in KEYS), opt=opt, n_steps=STEPS, flag_reduce=reduce)
    total = shard.sum_losses(res.loss.clone())
    torch.cuda.synchronize()
    return dict(own=list(own), local_flags=seen["local"].cpu().numpy(), flags=res.flags.cpu().numpy(), loss=res.loss.cpu().numpy(),
                total=total.cpu().numpy(), params=[p.cpu().numpy() for p in fc + [B]])


def _train_background(rank, world, dev):
    from vmap_amd import parallel
    fc, pe, fr = _background()
    Rl = BG["R"] // world
    idx = np.concatenate([np.arange(i * BG["R"] + rank, (i + 1) * BG["R"], world) for i in range(STEPS)])
    loc = tuple(torch.from_numpy(np.ascontiguousarray(fr[k][idx])).to(dev) for k in KEYS)
    bg = parallel.SharedBackgroundHip(fc, pe, Rl, BG["S"], dev, max_steps=STEPS)
    losses = bg.train_frame(*loc, n_steps=STEPS)
    bg.check_flags()
    torch.cuda.synchronize()
    return dict(losses=losses.cpu().numpy(), flags=bg.flags[:STEPS].cpu().numpy(), slab=bg.slab.cpu().numpy(),
                m=bg.opt.exp_avg.cpu().numpy(), steps=bg.opt.step)


def _train_background_owner(rank, world, dev):
    """parallel.OwnerBackgroundHip: rank 0 trains on ALL rays with the plain frame call, one slab broadcast per frame."""
    from vmap_amd import parallel
    fc, pe, fr = _background()
    own = parallel.OwnerBackgroundHip(fc, pe, BG["R"], BG["S"], dev, owner=0, max_steps=STEPS)
    before = own.slab.clone()
    full = tuple(torch.from_numpy(np.ascontiguousarray(fr[k])).to(dev) for k in KEYS) if rank == 0 else (None,) * 6
    res = None
    for _ in range(2):                                  # two frames: the second runs on the bound frame
        res = own.train_frame(*full, n_steps=STEPS)
    torch.cuda.synchronize()
    return dict(slab=own.slab.cpu().numpy(), changed=not torch.equal(before, own.slab), has_op=own.op is not None,
                losses=None if res is None else res.loss[:STEPS].cpu().numpy(), flags=None if res is None else res.flags[:STEPS].cpu().numpy())


def _worker(rank, world, port, ret, backend="gloo"):
    import torch.distributed as dist
    from vmap_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # gloo: both ranks share cuda:0 (RCCL refuses two ranks on one device); nccl: one device per rank, the real thing
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        transport = "RCCL, one device per rank" if backend == "nccl" else "gloo on device tensors"
        try:
            probe = torch.ones(4, device=dev)
            dist.all_reduce(probe)
            assert float(probe[0]) == world
        except Exception as e:                      # a torch build whose gloo cannot take device tensors: stage through the host
            if backend == "nccl":
                raise
            transport = f"gloo through host copies ({type(e).__name__})"
            native, native_b = dist.all_reduce, dist.broadcast

            def staged(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
                h = t.detach().cpu()
                native(h, op=op, group=group)
                t.copy_(h)

            def staged_b(t, src=0, group=None, async_op=False):
                h = t.detach().cpu()
                native_b(h, src=src, group=group)
                t.copy_(h)

            parallel.dist.all_reduce = staged
            parallel.dist.broadcast = staged_b
        c, frame = _object_frame()
        shard = parallel.ObjectShard(c["n"])
        out = dict(transport=transport, device=str(dev), objects=_train_objects(shard, c, frame, dev), background=_train_background(rank, world, dev),
                   owner=_train_background_owner(rank, world, dev))
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_the_single_rank_run():
    _two_ranks_equal_the_single_rank_run("gloo")


def test_two_ranks_on_two_gpus_over_rccl_equal_the_single_rank_run():
    """The same assertions with the process group an 8-GPU run uses: backend "nccl" (= RCCL), one device per rank.  Needs two GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    _two_ranks_equal_the_single_rank_run("nccl")


def _two_ranks_equal_the_single_rank_run(backend):
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret, backend), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1]
    if backend == "nccl":
        assert {ret[0]["device"], ret[1]["device"]} == {"cuda:0", "cuda:1"}
    from vmap_amd import parallel
    dev = torch.device(DEV)

    # ---- objects: world-1 run of the same classes here ----
    c, frame = _object_frame()
    one = _train_objects(parallel.ObjectShard(c["n"], rank=0, world_size=1), c, frame, dev)
    assert one["flags"][:, 0].tolist() == [1] * STEPS                       # the depth term is dropped batch-wide in every step
    r0, r1 = ret[0]["objects"], ret[1]["objects"]
    assert r0["own"] == [0, 2] and r1["own"] == [1, 3]
    assert r0["local_flags"][:, 0].tolist() == [1] * STEPS and r1["local_flags"][:, 0].tolist() == [0] * STEPS   # only rank 0 sees it ...
    for r in (r0, r1):
        assert r["flags"][:, :3].tolist() == one["flags"][:, :3].tolist()      # ... both apply it
        assert int(r["flags"][:, 3].max()) == 0
        np.testing.assert_allclose(r["total"], one["loss"], rtol=1e-5)         # sum of the shards' losses = the batch loss
        for t, p in enumerate(r["params"]):                                    # object k's trajectory does not depend on the sharding
            assert np.array_equal(p, one["params"][t][r["own"]]), t
    assert not np.array_equal(one["params"][2], cases.build_case("drop_depth")["fc"][2])      # and the steps did train

    # ---- shared background: replicas identical across ranks, equal to full-batch training ----
    b0, b1 = ret[0]["background"], ret[1]["background"]
    ref = _train_background(0, 1, dev)
    assert b0["steps"] == b1["steps"] == ref["steps"] == STEPS
    assert np.array_equal(b0["slab"], b1["slab"]) and np.array_equal(b0["m"], b1["m"])          # bit-identical replicas
    assert np.array_equal(b0["losses"], b1["losses"]) and np.array_equal(b0["flags"], b1["flags"])
    np.testing.assert_allclose(b0["losses"], ref["losses"], rtol=1e-5)
    assert b0["flags"].tolist() == ref["flags"].tolist()
    d = np.abs(b0["slab"].astype(np.float64) - ref["slab"])
    assert d.max() <= STEPS * 1.2e-3 and np.median(d) < 1e-6                   # Adam sign flips on ~0 gradients only
    print("transport:", ret[0]["transport"])

    # ---- owner-computes background: the owner = single-GPU training bit for bit, the other rank a bit-identical copy, no operator ----
    o0, o1 = ret[0]["owner"], ret[1]["owner"]
    one = _train_background_owner(0, 1, dev)
    assert o0["has_op"] and not o1["has_op"] and o1["losses"] is None
    assert o0["changed"] and np.array_equal(o0["slab"], o1["slab"]) and np.array_equal(o0["slab"], one["slab"])
    assert np.array_equal(o0["losses"], one["losses"]) and int(o0["flags"][:, 3].max()) == 0


def test_operator_on_a_non_current_device():
    """One process, several GPUs: an operator bound to cuda:1 called while cuda:0 is current runs on cuda:1 (the library
    switches to the device that owns the stream it is given).  Needs a second GPU."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs in one process")
    from conftest import GRAD_KEYS, load_golden, relerr
    from vmap_amd import step
    c = cases.build_case("tiny")
    g = load_golden("tiny")
    d1 = torch.device("cuda:1")
    torch.cuda.set_device(0)
    fc = [torch.from_numpy(a).to(d1) for a in c["fc"]]
    B, sc = torch.from_numpy(c["B"]).to(d1), torch.from_numpy(c["scale"]).to(d1)
    b = {k: torch.from_numpy(v).to(d1) for k, v in c["batch"].items()}
    op = step.VmapStep(c["n"], c["R"], c["S"], c["H"], device=d1)
    gfc = [torch.zeros_like(t) for t in fc]
    gB = torch.zeros_like(B)
    res = op.fwd_bwd(fc, B, sc, *(b[k] for k in KEYS), grads_fc=gfc, grad_B=gB)
    torch.cuda.synchronize(d1)
    assert torch.cuda.current_device() == 0
    assert abs(float(res.loss[0]) - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for k, t in zip(GRAD_KEYS, gfc + [gB]):
        assert relerr(t.cpu().numpy(), g[k]) < 1e-4, k


def test_bench_runs_with_two_ranks_on_one_gpu(tmp_path):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank), with the gloo dry-run backend
    so that both ranks can share cuda:0: the object shards' flag reduction inside the pre-heat / warm-up / timed loops, the
    background replicas on their own group and stream, the max-over-ranks timing and rank 0's JSON line.  (A pre-heat loop
    that ran until each rank's own clock said stop once gave the ranks different collective counts: a hang under RCCL.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VMAP_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "5",
           "--preheat-ms", "30", "--profile-reps", "20", "--no-cpu-baseline", "--no-gpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["steps"] == 40 and j["scaling"] == "weak"
    assert j["value"] > 0 and j["with_background"]["rays_per_step_this_rank"] == 600, j["with_background"]
    wb = j["with_background"]
    assert wb["ray_sharded"]["beside_objects_ms_per_step"] > 0 and wb["ray_sharded"]["allreduce_alone_us"] > 0
    assert wb["owner_computes"]["beside_objects_ms_per_step"] > 0
    assert len(j["world"]["ms_per_step_per_rank"]) == 2 and max(j["world"]["ms_per_step_per_rank"]) <= j["ms_per_step"] * 1.0001
    assert j["world"]["world_size"] == 2 and len(j["world"]["devices"]) == 2 and j["world"]["backend"] == "gloo"


def test_plain_bench_gpus_2_starts_its_own_ranks(tmp_path):
    """plain `python bench.py --gpus 2` - no torch.distributed.run around it, no rank environment - must start
    its two ranks itself and print ONE line with n_gpus == 2 (gloo dry-run backend: both ranks share the box's one device), with the
    region's collective costs and the ten-frame region beside `value`."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VMAP_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "GROUP_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--preheat-ms", "30",
           "--profile-reps", "20", "--no-background"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 20 and j["scaling"] == "weak" and j["value"] > 0
    assert j["world"]["world_size"] == 2 and len(j["world"]["devices"]) == 2 and len(j["world"]["ms_per_step_per_rank"]) == 2
    rc = j["world"]["region_costs"]
    assert rc["flag_allreduce_alone_us_per_frame"] > 0 and rc["barrier_alone_us"] > 0
    assert rc["ten_frame_region"]["steps_per_repeat"] == 200 and rc["ten_frame_region"]["value"] > 0
    assert j["repeats"]["ms_per_step_incl_closing_barrier"] >= j["ms_per_step"]


def test_plain_bench_refuses_more_ranks_than_devices():
    """... and over RCCL it refuses loudly (exit status 2, a message) when the box has fewer devices than --gpus, instead of running
    one rank and printing n_gpus 1 (what round 5's file did)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "VMAP_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(want), "--steps", "20", "--warmup", "5"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-1000:])
    assert "refusing" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
