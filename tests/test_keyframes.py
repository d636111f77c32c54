"""Shared keyframe store.

* ``ObjectKeyframes`` is a reference-counted table over the store (storage only: the reference's keyframe policy, vmap.py:205-262,
  is out of scope and stays with the caller).
* The batched sampler reading the shared store gives bit-identical samples to the sampler reading the reference's
  per-object buffers built from the same frames (simulator here, device kernel in the gpu tier).
"""
import numpy as np
import pytest
import torch

import sampler_cases
import simlib
from vmap_amd.keyframes import FrameStore, ObjectKeyframes

OBJ_ID, OTHER_ID = 7, 3


def test_table_is_storage_only_and_reference_counted():
    """write / note_latest / append: entries hold the frames they were given, replaced frames lose their reference, the latest
    two indices follow note_latest; the stand-in append() overwrites the oldest entry of a full table.  (The reference's keyframe
    POLICY - vmap.py:205-262 - is out of scope and stays with the caller.)"""
    W, H, K = 6, 4, 3
    store = FrameStore(K + 3, W, H, device="cpu")
    z = torch.zeros(W, H, 3, dtype=torch.uint8)
    put = lambda fid: store.put(z, torch.full((W, H), float(fid)), torch.zeros(W, H, dtype=torch.int32), torch.eye(4), fid)
    ok = ObjectKeyframes(store, 1, put(0), [0., 5., 0., 3.], keyframe_buffer_size=K)
    assert ok.n_keyframes == 1 and ok.sampler_entry()["last2"] == (0, 0)
    written = [ok.append(put(f), [float(f), 5., 0., 3.]) for f in range(1, 6)]
    store.collect()
    assert written == [1, 2, 0, 1, 2] and ok.n_keyframes == K
    assert [store.frame_of_slot[s] for s in ok.slots] == [3, 4, 5] and ok.sampler_entry()["last2"] == (1, 2)
    assert [float(store.depth[s, 0, 0]) for s in ok.slots] == [3.0, 4.0, 5.0] and [float(b) for b in ok.bbox[:, 0]] == [3.0, 4.0, 5.0]
    assert {s for s in range(store.capacity) if store.refs[s] > 0} == set(ok.slots)
    ok.write(1, put(9), [9., 5., 0., 3.]); ok.note_latest(1)                        # a caller's own policy: any index, any order
    assert store.frame_of_slot[ok.slots[1]] == 9 and ok.sampler_entry()["last2"] == (2, 1)
    with pytest.raises(IndexError):
        ok.write(K, put(10), [0., 1., 0., 1.])


def test_store_refcounts_shared_between_objects():
    store = FrameStore(4, 6, 4, device="cpu")
    z = torch.zeros(6, 4, 3, dtype=torch.uint8)
    s0 = store.put(z, torch.zeros(6, 4), torch.zeros(6, 4, dtype=torch.int32), torch.eye(4), 0)
    a = ObjectKeyframes(store, 1, s0, [0, 1, 0, 1], keyframe_buffer_size=4)
    b = ObjectKeyframes(store, 2, s0, [0, 1, 0, 1], keyframe_buffer_size=4)
    assert store.refs[s0] == 2
    s1 = store.put(z, torch.zeros(6, 4), torch.zeros(6, 4, dtype=torch.int32), torch.eye(4), 1)
    a.append(s1, [0, 1, 0, 1])
    store.collect()
    assert store.refs[s1] == 1 and store.frame_of_slot[s1] == 1
    s2 = store.put(z, torch.zeros(6, 4), torch.zeros(6, 4, dtype=torch.int32), torch.eye(4), 2)
    store.collect()                                   # nobody kept frame 2
    assert store.frame_of_slot[s2] is None
    with pytest.raises(ValueError):
        FrameStore(256, 6, 4, device="cpu")


def _shared_scene(sc, perm_seed=0):
    """The same frames as scene `sc` in a shared store: slots permuted, state byte replaced by an instance image."""
    K = sc["K"]
    rng = np.random.default_rng(perm_seed)
    C = K + 3
    slots = rng.permutation(C)[:K].astype(np.int32)
    rgbx = rng.integers(0, 256, (C, sc["W"], sc["H"], 4)).astype(np.uint8)      # byte 3 = junk on purpose
    depth = rng.uniform(0.1, 1.0, (C, sc["W"], sc["H"])).astype(np.float32)
    inst = np.full((C, sc["W"], sc["H"]), OTHER_ID, np.int32)
    t_wc = rng.standard_normal((C, 4, 4)).astype(np.float32)
    for k in range(K):
        s = slots[k]
        rgbx[s, :, :, :3] = sc["rgbs"][k, :, :, :3]
        depth[s] = sc["depth"][k]
        t_wc[s] = sc["t_wc"][k]
        state = sc["rgbs"][k, :, :, 3]
        inst[s] = np.where(state == 1, OBJ_ID, np.where(state == 2, -1, OTHER_ID))
    out = dict(sc)
    out.update(store=dict(rgbx=rgbx, depth=depth, inst=inst, t_wc=t_wc), slots=slots, obj_id=OBJ_ID)
    return out


@pytest.mark.parametrize("name", list(sampler_cases.CASES))
def test_sim_sampler_shared_store_equals_per_object_buffers(name):
    sc = sampler_cases.build_scene(name)
    rnd = sampler_cases.draw_randoms(sc)
    own = simlib.sim_sample([sc], [rnd])
    shared = simlib.sim_sample([_shared_scene(sc, 5)], [rnd])
    for k in own:
        assert np.array_equal(own[k], shared[k]), k
    # Philox mode too (same seed, same counters)
    own = simlib.sim_sample([sc], None, seed=11, frame_counter=3)
    shared = simlib.sim_sample([_shared_scene(sc, 6)], None, seed=11, frame_counter=3)
    for k in own:
        assert np.array_equal(own[k], shared[k]), k


@pytest.mark.gpu
def test_gpu_sampler_shared_store_equals_per_object_buffers():
    from vmap_amd import sampler
    dev = "cuda:0"
    names = ["obj", "obj"]          # two objects over ONE store, different keyframe subsets
    scs = [sampler_cases.build_scene(n) for n in names]
    rnds = [sampler_cases.draw_randoms(sc) for sc in scs]
    sc0 = scs[0]
    K, W, H = sc0["K"], sc0["W"], sc0["H"]
    # object 1 sees the same frames with another state image: derive it by relabelling
    scs[1] = dict(scs[1]); scs[1]["rgbs"] = scs[1]["rgbs"].copy()
    st1 = (scs[0]["rgbs"][..., 3].astype(np.int32) + 1) % 3
    scs[1]["rgbs"][..., 3] = st1
    ids = (7, 9)
    store = FrameStore(K + 2, W, H, device=dev)
    oks = []
    for f in range(K):
        s0, s1 = scs[0]["rgbs"][f, :, :, 3], scs[1]["rgbs"][f, :, :, 3]
        # one instance image for both objects: id 7 where object 0's state says 'this', id 9 where object 1's does
        # (the two never coincide: state 1 is a cyclic relabelling of state 0), -1 = unknown elsewhere on a few pixels
        inst = np.where(s0 == 1, 7, np.where(s1 == 1, 9, np.where((np.arange(s0.size).reshape(s0.shape) % 11) == 0, -1, 3))).astype(np.int32)
        slot = store.put(torch.from_numpy(scs[0]["rgbs"][f, :, :, :3].copy()), torch.from_numpy(scs[0]["depth"][f]),
                         torch.from_numpy(inst), torch.from_numpy(scs[0]["t_wc"][f]), f)
        for j in range(2):
            if f == 0:
                oks.append(ObjectKeyframes(store, ids[j], slot, scs[0]["bbox"][0], keyframe_buffer_size=K + 1, center=scs[0]["center"]))
            else:
                oks[j].append(slot, scs[0]["bbox"][f])
    F, P, n1, n2 = sc0["F"], sc0["P"], sc0["n1"], sc0["n2"]
    fx, fy, cx, cy = sc0["intr"]

    def make():
        return sampler.FrameSampler(W, H, F, P, n1, n2, fx, fy, cx, cy, min_depth=sc0["min_bound"], device=dev, seed=5)

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # reference-layout buffers of the two objects (state byte baked from the same instance image, train.py:128-130)
    own_objs = []
    for j in range(2):
        rgbs = scs[0]["rgbs"].copy()
        inst_all = store.inst[[oks[j].slots[k] for k in range(K)]].cpu().numpy()
        rgbs[..., 3] = np.where(inst_all == ids[j], 1, np.where(inst_all == -1, 2, 0))
        own_objs.append(dict(rgbs=t(rgbs), depth=t(scs[0]["depth"]), t_wc=t(scs[0]["t_wc"]), bbox=t(scs[0]["bbox"]),
                             n_keyframes=K, last2=sc0["last2"], center=sc0["center"]))
    a, b = make(), make()
    a.set_objects(own_objs)
    b.set_objects([ok.sampler_entry() for ok in oks])
    assert [ok.n_keyframes for ok in oks] == [K, K] and oks[0].sampler_entry()["last2"] == tuple(sc0["last2"])
    rnd = {k: t(np.stack([r[k] for r in rnds]).astype(np.int32 if k == "kf_ids" else np.float32)) for k in rnds[0]}
    for test in (rnd, None):
        oa, ob = a.sample(test), b.sample(test)
        for k in oa:
            assert torch.equal(oa[k], ob[k]), k
    assert set(torch.unique(ob["sem"]).tolist()) <= {0, 1, 2} and (ob["sem"][0] == 1).any() and (ob["sem"][1] == 1).any()
