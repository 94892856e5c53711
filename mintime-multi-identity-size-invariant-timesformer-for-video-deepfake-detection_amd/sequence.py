"""Input-sequence builder: from "which faces of which identities exist for this video" to the side
inputs SizeInvariantTimeSformer.forward takes, for a whole batch, built on the device.

Mirrors the reference's data-preparation rules (it builds Python lists per sample on DataLoader workers):
  * identity order and truncation            deepfakes_dataset.py:149-158      (predict.py:196-199)
  * slots per identity                       deepfakes_dataset.py:50-53,160-190 (predict.py:203-243)
  * which faces fill an identity's slots     deepfakes_dataset.py:236-242      (predict.py:277-279)
  * face/frame area ratio                    deepfakes_dataset.py:246-262      (predict.py:285-296)
  * buckets, padding, mask, identities_mask, temporal positions: on the device, csrc/sequence.hip (mt_build_clip_inputs)

Host work here is per-video bookkeeping on a handful of integers (it depends on directory listings, exactly like the
reference's); everything that scales with B*F*49 happens in one kernel launch and never exists on the host.
"""
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import lib as L

MAX_IDENTITIES = 4      # the reference's slot table has rows for 1..4 identities (deepfakes_dataset.py:50-53)


@dataclass
class Identity:
    """One clustered identity of a video: its faces as (video frame number, crop height, crop width), any order."""
    name: str
    faces: List[Tuple[int, int, int]]
    slots: int = 0                      # filled in by plan_clip

    @property
    def mean_side(self):
        """Mean crop WIDTH: the reference takes groups()[0] of libmagic's "W x H" string (deepfakes_dataset.py:113); if any
        file of the identity fails to parse there, the whole identity's mean falls back to 0 (:114-115) -- the caller models
        that by handing a face with width <= 0, which zeroes the mean here as well."""
        if not self.faces or any(f[2] <= 0 for f in self.faces):
            return 0.0
        return float(np.mean([f[2] for f in self.faces]))


@dataclass
class ClipPlan:
    identities: List[Identity]
    video_wh: Tuple[int, int]
    variant: str = "dataset"
    chosen: List[List[Tuple[int, int, int]]] = field(default_factory=list)   # per identity: the faces that fill its slots


def slot_caps(num_frames: int, n_identities: int) -> List[int]:
    if not 1 <= n_identities <= MAX_IDENTITIES:
        raise KeyError(f"{n_identities} identities: the slot table covers 1..{MAX_IDENTITIES} (deepfakes_dataset.py:50-53)")
    f = num_frames
    return {1: [f], 2: [f // 2, f // 2], 3: [f // 3, f // 3, f // 4], 4: [f // 3, f // 3, f // 8, f // 8]}[n_identities]


def assign_slots(face_counts: Sequence[int], num_frames: int) -> List[int]:
    """Slots per identity (identities already ordered and truncated).  Under-full identities hand their unused quota to the
    NEXT identity's face count, over-full ones are capped and remember their surplus; a short sequence is topped up from the
    surpluses in order and finally by padding the last identity."""
    n = len(face_counts)
    faces = [int(c) for c in face_counts]
    surplus = [0] * n
    if n == 1:
        faces[0] = num_frames
    else:
        caps = slot_caps(num_frames, n)
        for i in range(n):
            if faces[i] < caps[i] and i < n - 1:
                faces[i + 1] += caps[i] - faces[i]
            elif faces[i] > caps[i]:
                surplus[i] = faces[i] - caps[i]
                faces[i] = caps[i]
    short = num_frames - sum(faces)
    for i in range(n):
        if short <= 0:
            break
        add = min(surplus[i], short)
        faces[i] += add
        short -= add
    if short > 0:
        faces[-1] += short
    return faces


def plan_clip(identities: Sequence[Identity], num_frames: int, video_wh, max_identities: int = 3, ordering: int = 0,
              sample_index: int = 1, variant: str = "dataset") -> ClipPlan:
    """Order identities (0: by mean face side, 1: by number of faces; descending, ties keep input order), keep the first
    max_identities, give each its slots and pick the faces that fill them (temporal order, uniform subsampling whose phase
    alternates with the sample index in the dataset and is fixed in predict.py)."""
    if ordering == 0:
        ids = sorted(identities, key=lambda i: i.mean_side, reverse=True)
    elif ordering == 1:
        ids = sorted(identities, key=lambda i: len(i.faces), reverse=True)
    else:
        raise ValueError("identities_ordering 2 (random shuffle) is not reproducible; shuffle before calling")
    ids = [Identity(i.name, list(i.faces)) for i in ids[:max_identities]]
    for ident, s in zip(ids, assign_slots([len(i.faces) for i in ids], num_frames)):
        ident.slots = s
    plan = ClipPlan(ids, tuple(video_wh), variant)
    for ident in ids:
        faces = sorted(ident.faces, key=lambda f: f[0])
        k = len(faces)
        if k > ident.slots:
            if variant == "predict" or sample_index % 2:
                idx = np.round(np.linspace(0, k - 2, ident.slots)).astype(int)
            else:
                idx = np.round(np.linspace(1, k - 1, ident.slots)).astype(int)
            faces = [faces[i] for i in idx]
        plan.chosen.append(faces)
    return plan


def area_ratio(face_h: int, face_w: int, video_w: float, video_h: float, variant: str = "dataset") -> int:
    video_area = video_w * video_h / 2
    face_area = face_h * face_w / 2 if variant == "dataset" else face_h * face_w
    r = int(face_area * 100 / video_area)
    if not 0 <= r <= 100:
        raise IndexError(f"face/frame area ratio {r} is outside the 20 size buckets (deepfakes_dataset.py:30-31,261-262)")
    return r


def build_batch(plans: Sequence[ClipPlan], num_frames: int, num_patches: int = 49, device="cuda", size_embedding_on_host=True):
    """-> dict(mask [B,F] bool, identities_mask [B,F,F] bool, size_embedding [B,F] int32, positions [B,1+F*49] int64), the
    collated equivalents of the tuple members at deepfakes_dataset.py:339.  size_embedding is returned on the host by default
    because that is where the reference's callers leave it (train.py:355); the device copy is `size_embedding_dev`."""
    B, F = len(plans), num_frames
    slots = np.zeros((B, MAX_IDENTITIES), np.int32)
    valid = np.zeros((B, MAX_IDENTITIES), np.int32)
    frames = np.zeros((B, F), np.int32)
    ratio = np.zeros((B, F), np.int32)
    modes = {p.variant for p in plans}
    if len(modes) != 1:
        raise ValueError("one batch, one variant")
    for b, plan in enumerate(plans):
        s = 0
        if sum(i.slots for i in plan.identities) != F:
            raise ValueError(f"clip {b}: slots {[i.slots for i in plan.identities]} do not add up to num_frames={F}")
        for i, (ident, faces) in enumerate(zip(plan.identities, plan.chosen)):
            slots[b, i], valid[b, i] = ident.slots, min(len(faces), ident.slots)
            for k, (fr, h, w) in enumerate(faces[:ident.slots]):
                frames[b, s + k] = fr
                ratio[b, s + k] = area_ratio(h, w, plan.video_wh[0], plan.video_wh[1], plan.variant)
            s += ident.slots
    dev = torch.device(device)
    packed = torch.from_numpy(np.concatenate([slots.ravel(), valid.ravel(), frames.ravel(), ratio.ravel()])).pin_memory()
    d = packed.to(dev, non_blocking=True)
    o = [0, slots.size, 2 * slots.size, 2 * slots.size + frames.size]
    mask = torch.empty(B, F, dtype=torch.bool, device=dev)
    ident = torch.empty(B, F, F, dtype=torch.bool, device=dev)
    sizes = torch.empty(B, F, dtype=torch.int32, device=dev)
    positions = torch.empty(B, 1 + F * num_patches, dtype=torch.int64, device=dev)
    L.check(L.get().mt_build_clip_inputs(L.ptr(d[o[0]:]), L.ptr(d[o[1]:]), L.ptr(d[o[2]:]), L.ptr(d[o[3]:]), L.ptr(mask), L.ptr(ident),
                                         L.ptr(sizes), L.ptr(positions), B, F, num_patches, MAX_IDENTITIES,
                                         1 if modes.pop() == "predict" else 0, L.stream_ptr()), "mt_build_clip_inputs")
    out = dict(mask=mask, identities_mask=ident, positions=positions, size_embedding_dev=sizes)
    out["size_embedding"] = sizes.cpu() if size_embedding_on_host else sizes
    return out
