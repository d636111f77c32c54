"""Shared keyframe store + the per-object table the batched sampler reads.  STORAGE ONLY.

The reference gives every object its own copy of every keyframe it keeps: ``rgbs_batch [K,W,H,4]`` u8 (RGB + pixel
state), ``depth_batch [K,W,H]`` f32, ``t_wc_batch [K,4,4]`` (vmap.py:143-176) - 130 MB per object at 20 x 1200 x 680,
2.6 GB for 20 objects that mostly hold the SAME frames.  Here a frame is stored once:

``FrameStore``       device ring of C frames: ``rgbx`` u8 [C,W,H,4] (byte 3 unused), ``depth`` f32 [C,W,H],
                     ``inst`` i32 [C,W,H] (instance ids, -1 = unknown), ``t_wc`` f32 [C,4,4]; reference-counted slots.
``ObjectKeyframes``  per object: the table keyframe index -> store slot, the 2-D boxes, the two latest keyframe indices - what
                     ``vmapstep_sample_object`` needs, nothing else.  WHICH frame becomes a keyframe and which entry a full
                     buffer overwrites or prunes (``sceneObject.append_keyframe`` / ``prune_keyframe``, vmap.py:205-262) is
                     keyframe POLICY: out of scope and left with the caller, who calls ``write(k, ...)``
                     where the reference writes ``rgbs_batch[k] / depth_batch[k] / t_wc_batch[k] / bbox[k]`` and
                     ``note_latest(k)`` where it appends to its latest-keyframe queue.  ``append`` is this package's own
                     minimal stand-in for tests and demos (every frame a keyframe, oldest entry overwritten when full).

The pixel state the reference bakes into byte 3 per object (train.py:128-130: 1 = this object, 2 = unknown, 0 = other) is
derived by the sampler kernel from ``inst`` and the object's id at gather time (``vmapstep_sample_object.slots/inst``),
so the batched sampler reads the shared store directly: ``FrameSampler.set_objects([ok.sampler_entry() for ok in ...])``.
"""
from __future__ import annotations

from typing import List

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
    """The keyframe table of ONE object over a shared FrameStore (storage only; the keyframe policy is the caller's)."""

    def __init__(self, store: FrameStore, obj_id: int, first_slot: int, bbox_2d, keyframe_buffer_size: int = 20, center=(0.0, 0.0, 0.0)):
        self.store, self.obj_id = store, int(obj_id)
        self.keyframe_buffer_size = int(keyframe_buffer_size)
        self.n_keyframes = 0
        self.latest: List[int] = []                           # the (up to) two newest keyframe indices, oldest first
        self.slots = [-1] * self.keyframe_buffer_size        # keyframe index -> store slot
        self.bbox = torch.zeros(self.keyframe_buffer_size, 4, dtype=torch.float32, device=store.device)
        self.center = tuple(float(c) for c in center)
        self._slots_dev = torch.zeros(self.keyframe_buffer_size, dtype=torch.int32, device=store.device)
        self._next = 0                                        # append()'s ring position once the table is full
        self.write(0, first_slot, bbox_2d)
        self.note_latest(0)

    def write(self, k: int, slot: int, bbox_2d):
        """Entry k := frame in store slot `slot` with its 2-D box (the old entry's frame loses a reference)."""
        if not 0 <= k < self.keyframe_buffer_size:
            raise IndexError(f"keyframe index {k} outside the table of {self.keyframe_buffer_size}")
        if self.slots[k] >= 0:
            self.store.release(self.slots[k])
        self.store.retain(slot)
        self.slots[k] = int(slot)
        self._slots_dev[k] = int(slot)
        self.bbox[k] = torch.as_tensor(bbox_2d, dtype=torch.float32)
        self.n_keyframes = max(self.n_keyframes, k + 1)

    def note_latest(self, k: int):
        """k is now the newest keyframe (the sampler draws a fixed share of its rays from the latest two)."""
        self.latest = [i for i in self.latest if i != k][-1:] + [int(k)]

    def append(self, slot: int, bbox_2d) -> int:
        """This package's stand-in policy (NOT the reference's): every frame becomes a keyframe; a full table overwrites its oldest
        entry.  Returns the index written."""
        if self.n_keyframes < self.keyframe_buffer_size:
            k = self.n_keyframes
        else:
            k = self._next
            self._next = (self._next + 1) % self.keyframe_buffer_size
        self.write(k, slot, bbox_2d)
        self.note_latest(k)
        return k

    # ---- what the batched sampler needs (vmapstep_sample_object in shared-store mode) ------------------------
    def sampler_entry(self) -> dict:
        last2 = (self.latest + [0, 0])[:2] if len(self.latest) < 2 else self.latest[-2:]
        return dict(store=self.store, slots=self._slots_dev, bbox=self.bbox, n_keyframes=self.n_keyframes,
                    last2=tuple(int(v) for v in last2), center=self.center, obj_id=self.obj_id)
