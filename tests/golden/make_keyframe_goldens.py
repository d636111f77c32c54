"""Generate tests/golden/keyframes_policy.json by running the REAL reference bookkeeping (vmap.py:205-262
``sceneObject.append_keyframe`` / ``prune_keyframe``) on CPU.  Authoring container only (needs /root/reference).

``vmap.py`` imports open3d / trimesh / cv2 / imgviz / skimage (stubbed, unused here) and ``bidict`` (pinned
bidict==0.22.0, environment.yml:77, not installed and not vendored).  The policy depends on bidict's item ORDER
(``list(kf_id_dict.items())[:-2]``), so the stand-in below restates the part of bidict 0.22.0 the reference uses -
an insertion-ordered forward dict plus ``.inv[value] = key`` with the default ``on_dup`` (``BidictBase._write``: always
``fwdm[newkey] = newval; invm[newval] = newkey``; on key duplication in the view being written delete the stale
entry of the OTHER dict) - i.e. writing through ``.inv`` drops the old forward item and appends the new one LAST.
Every frame's depth image is filled with its frame id, so the fixture also records which frame each keyframe buffer
entry holds after every call.
"""
from __future__ import annotations

import json
import os
import random
import sys
import types
from unittest import mock

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True


class _Inv:
    def __init__(self, owner):
        self.o = owner

    def __setitem__(self, val, key):            # b.inv[val] = key
        o = self.o
        if key in o and dict.__getitem__(o, key) == val:
            return                              # same item: no-op
        if key in o:
            raise ValueError("ValueDuplicationError")      # default on_dup.val = RAISE
        for k0 in [k0 for k0, v0 in o.items() if v0 == val]:
            dict.__delitem__(o, k0)             # "just key duplication" in the inverse view: drop the stale forward item
        dict.__setitem__(o, key, val)           # ... and the new forward item is the newest one

    def __getitem__(self, val):
        return next(k for k, v in self.o.items() if v == val)


class bidict(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.inv = _Inv(self)

    def __setitem__(self, key, val):
        for k0 in [k0 for k0, v0 in self.items() if v0 == val and k0 != key]:
            raise ValueError("ValueDuplicationError")
        dict.__setitem__(self, key, val)


def import_reference_vmap():
    for name in ("open3d", "trimesh", "cv2", "imgviz", "skimage", "skimage.measure", "scipy.spatial"):
        sys.modules.setdefault(name, mock.MagicMock())
    sys.modules["bidict"] = types.SimpleNamespace(bidict=bidict)
    sys.path.insert(0, REF)
    import vmap as ref_vmap
    assert os.path.dirname(ref_vmap.__file__) == REF
    return ref_vmap


def main():
    ref_vmap = import_reference_vmap()
    from cfg import Config
    cases = []
    for buf, step, n_frames, seed in ((6, 3, 40, 1), (5, 1, 25, 2), (8, 4, 60, 3), (20, 25, 120, 4)):
        cfg = Config(os.path.join(REF, "configs/Replica/config_replica_room0_vMAP.json"))
        cfg.data_device = cfg.training_device = "cpu"
        cfg.W, cfg.H = 6, 4
        cfg.keyframe_buffer_size, cfg.keyframe_step = buf, step
        W, H = 6, 4
        rgb = torch.zeros(W, H, 3, dtype=torch.uint8)
        mask = torch.ones(W, H, dtype=torch.uint8)
        t_wc = torch.eye(4)

        def dep(fid):
            return torch.full((W, H), float(fid), dtype=torch.float32)

        random.seed(seed)
        first = 100
        with mock.patch("builtins.print", lambda *a, **k: None):
            obj = ref_vmap.sceneObject(cfg, 1, rgb, dep(first), mask, torch.tensor([0., 5., 0., 3.]), t_wc, first)
            trace = []
            for i in range(n_frames):
                fid = first + 1 + i
                obj.append_keyframe(rgb, dep(fid), mask, torch.tensor([0., 5., 0., 3.]) + i % 2, t_wc, fid)
                trace.append(dict(n_keyframes=int(obj.n_keyframes), kf_pointer=None if obj.kf_pointer is None else int(obj.kf_pointer),
                                  latest=[int(v) for v in obj.lastest_kf_queue],
                                  items=[[int(k), int(v)] for k, v in obj.kf_id_dict.items()],
                                  frame_in_entry=[int(obj.depth_batch[k, 0, 0]) for k in range(buf - 1)] if obj.kf_buffer_full
                                  else [int(obj.depth_batch[k, 0, 0]) for k in range(obj.n_keyframes)],
                                  bbox0=[float(obj.bbox[k, 0]) for k in range(obj.n_keyframes)]))
        cases.append(dict(keyframe_buffer_size=buf, keyframe_step=step, seed=seed, first_frame=first, trace=trace))
        print(f"buffer {buf} step {step}: {n_frames} frames, final items {trace[-1]['items']}")
    with open(os.path.join(HERE, "keyframes_policy.json"), "w") as fh:
        json.dump(cases, fh)


if __name__ == "__main__":
    main()
