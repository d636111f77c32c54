"""Shared keyframe store + the reference's per-object keyframe policy on top of it.

The reference gives every object its own copy of every keyframe it keeps: ``rgbs_batch [K,W,H,4]`` u8 (RGB + pixel
state), ``depth_batch [K,W,H]`` f32, ``t_wc_batch [K,4,4]`` (vmap.py:143-176) - 130 MB per object at 20 x 1200 x 680,
2.6 GB for 20 objects that mostly hold the SAME frames.  Here a frame is stored once:

``FrameStore``       device ring of C frames: ``rgbx`` u8 [C,W,H,4] (byte 3 unused), ``depth`` f32 [C,W,H],
                     ``inst`` i32 [C,W,H] (instance ids, -1 = unknown), ``t_wc`` f32 [C,4,4]; reference-counted slots.
``ObjectKeyframes``  per object: the table keyframe index -> store slot, the 2-D boxes, and exactly the bookkeeping of
                     ``sceneObject.append_keyframe`` / ``prune_keyframe`` (vmap.py:205-262): every ``keyframe_step``-th
                     frame becomes a keyframe, other frames overwrite the newest entry, a full buffer overwrites
                     ``kf_pointer`` and prunes a random keyframe other than the latest two (``random.choice``, so the
                     decisions are reproducible against the reference under the same ``random.seed``).

The pixel state the reference bakes into byte 3 per object (train.py:128-130: 1 = this object, 2 = unknown, 0 = other) is
derived by the sampler kernel from ``inst`` and the object's id at gather time (``vmapstep_sample_object.slots/inst``),
so the batched sampler reads the shared store directly: ``FrameSampler.set_objects([ok.sampler_entry() for ok in ...])``.
"""
from __future__ import annotations

import random
from typing import Dict, List, Optional

import torch


class FrameStore:
    MAX_SLOTS = 255                      # the sampler kernel packs the slot into 8 bits next to the pixel coordinates

    def __init__(self, capacity: int, width: int, height: int, device="cuda:0"):
        if not 1 <= capacity <= self.MAX_SLOTS:
            raise ValueError(f"capacity must be in [1, {self.MAX_SLOTS}]")
        self.capacity, self.W, self.H = int(capacity), int(width), int(height)
        self.device = torch.device(device)
        self.rgbx = torch.zeros(capacity, width, height, 4, dtype=torch.uint8, device=self.device)
        self.depth = torch.zeros(capacity, width, height, dtype=torch.float32, device=self.device)
        self.inst = torch.full((capacity, width, height), -1, dtype=torch.int32, device=self.device)
        self.t_wc = torch.zeros(capacity, 4, 4, dtype=torch.float32, device=self.device)
        self.refs = [0] * capacity
        self.frame_of_slot: List[Optional[int]] = [None] * capacity

    def put(self, rgb: torch.Tensor, depth: torch.Tensor, inst: torch.Tensor, t_wc: torch.Tensor, frame_id: int) -> int:
        """Store one frame (rgb u8 [W,H,3], depth f32 [W,H], inst int [W,H], t_wc [4,4]) in a free slot; returns the
        slot with a reference count of 0 - objects that keep the frame ``retain`` it, ``collect()`` frees the rest."""
        try:
            slot = self.refs.index(0, 0)
            while self.frame_of_slot[slot] is not None:          # occupied by a frame nobody retained yet
                slot = self.refs.index(0, slot + 1)
        except ValueError:
            raise RuntimeError("FrameStore is full: raise capacity or collect() unreferenced frames") from None
        self.rgbx[slot, :, :, :3] = rgb.to(self.device)
        self.depth[slot] = depth.to(self.device)
        self.inst[slot] = inst.to(self.device, torch.int32)
        self.t_wc[slot] = t_wc.to(self.device)
        self.frame_of_slot[slot] = int(frame_id)
        return slot

    def retain(self, slot: int):
        self.refs[slot] += 1

    def release(self, slot: int):
        assert self.refs[slot] > 0
        self.refs[slot] -= 1
        if self.refs[slot] == 0:
            self.frame_of_slot[slot] = None

    def collect(self):
        """Free the slots of frames no object kept (call once per frame after every object saw it)."""
        for s in range(self.capacity):
            if self.refs[s] == 0:
                self.frame_of_slot[s] = None

    def bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (self.rgbx, self.depth, self.inst, self.t_wc))


class ObjectKeyframes:
    """Keyframe bookkeeping of ONE object over a shared FrameStore; same decisions as vmap.py:205-262."""

    def __init__(self, store: FrameStore, obj_id: int, first_slot: int, bbox_2d, frame_id: int = 0,
                 keyframe_buffer_size: int = 20, keyframe_step: int = 25, center=(0.0, 0.0, 0.0)):
        self.store, self.obj_id = store, int(obj_id)
        self.keyframe_buffer_size, self.keyframe_step = int(keyframe_buffer_size), int(keyframe_step)
        self.n_keyframes = 1                                  # vmap.py:128
        self.kf_pointer: Optional[int] = None
        self.kf_id_dict: Dict[int, int] = {int(frame_id): 0}  # frame id -> keyframe index (insertion ordered, like bidict)
        self.kf_buffer_full = False
        self.frame_cnt = 0
        self.lastest_kf_queue: List[int] = []
        self.slots = [-1] * self.keyframe_buffer_size        # keyframe index -> store slot
        self.bbox = torch.zeros(self.keyframe_buffer_size, 4, dtype=torch.float32, device=store.device)
        self.center = tuple(float(c) for c in center)
        self._slots_dev = torch.zeros(self.keyframe_buffer_size, dtype=torch.int32, device=store.device)
        self._set(0, first_slot, bbox_2d)

    # ---- storage of one entry -------------------------------------------------------------------------------
    def _set(self, k: int, slot: int, bbox_2d):
        if self.slots[k] >= 0:
            self.store.release(self.slots[k])
        self.store.retain(slot)
        self.slots[k] = int(slot)
        self._slots_dev[k] = int(slot)
        self.bbox[k] = torch.as_tensor(bbox_2d, dtype=torch.float32)

    def _inv_set(self, k: int, frame_id: int):
        """``kf_id_dict.inv[k] = frame_id`` of the reference's ``bidict`` (bidict==0.22.0, environment.yml:77; restated from
        its published ``BidictBase._write``): the forward item whose value is k is dropped and ``frame_id -> k`` is
        inserted as the NEWEST item - the forward mapping is an insertion-ordered dict and ``prune_keyframe`` protects
        its last two items.  Re-assigning the same pair is a no-op; a frame id that already maps to another keyframe
        raises, as bidict's default ``on_dup`` does."""
        frame_id = int(frame_id)
        if self.kf_id_dict.get(frame_id, None) == k:
            return
        if frame_id in self.kf_id_dict:
            raise ValueError(f"frame id {frame_id} already names keyframe {self.kf_id_dict[frame_id]}")
        for f, v in list(self.kf_id_dict.items()):
            if v == k:
                del self.kf_id_dict[f]
        self.kf_id_dict[frame_id] = k

    # ---- vmap.py:205-257 ------------------------------------------------------------------------------------
    def append_keyframe(self, slot: int, bbox_2d, frame_id: int = 1):
        assert self.n_keyframes <= self.keyframe_buffer_size - 1
        is_kf = (self.frame_cnt % self.keyframe_step == 0) or self.n_keyframes == 1
        if self.n_keyframes == self.keyframe_buffer_size - 1:          # buffer full: overwrite kf_pointer, maybe prune
            self.kf_buffer_full = True
            if self.kf_pointer is None:
                self.kf_pointer = self.n_keyframes
            self._set(self.kf_pointer, slot, bbox_2d)
            self._inv_set(self.kf_pointer, frame_id)
            if is_kf:
                self.lastest_kf_queue.append(self.kf_pointer)
                _, pruned_kf_id = self.prune_keyframe()
                self.kf_pointer = pruned_kf_id
        else:
            if not is_kf:                                              # not a keyframe: replace the newest entry
                self._set(self.n_keyframes - 1, slot, bbox_2d)
                self._inv_set(self.n_keyframes - 1, frame_id)
            else:                                                      # new keyframe
                self.kf_id_dict[int(frame_id)] = self.n_keyframes
                self._set(self.n_keyframes, slot, bbox_2d)
                self.lastest_kf_queue.append(self.n_keyframes)
                self.n_keyframes += 1
        self.frame_cnt += 1
        if len(self.lastest_kf_queue) > 2:
            self.lastest_kf_queue = self.lastest_kf_queue[-2:]

    def prune_keyframe(self):
        return random.choice(list(self.kf_id_dict.items())[:-2])       # vmap.py:259-262: never the latest two

    # ---- what the batched sampler needs (vmapstep_sample_object in shared-store mode) ------------------------
    def sampler_entry(self) -> dict:
        last2 = (self.lastest_kf_queue + [0, 0])[:2] if len(self.lastest_kf_queue) < 2 else self.lastest_kf_queue[-2:]
        return dict(store=self.store, slots=self._slots_dev, bbox=self.bbox, n_keyframes=self.n_keyframes,
                    last2=tuple(int(v) for v in last2), center=self.center, obj_id=self.obj_id)
