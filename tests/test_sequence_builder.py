"""Next-row f1, the input-sequence builder.

CPU: the oracle's restatement and the product's host logic against the reference's own get_sorted_identities
(tests/golden/slots.json: deepfakes_dataset.py:130-186 run on throw-away directory trees by tools/make_golden.py), plus
hand-derived cases for the per-clip tensors (PARITY UNPINNED there: __getitem__ needs cv2/albumentations, which cannot be
imported or faked here).  GPU: mt_build_clip_inputs bit-exact against the oracle on random clips."""
import json
import os

import numpy as np
import pytest
import torch

import mintime_amd
from mintime_amd import sequence as S
from oracle import mintime_oracle as O
from tests.util import GOLDEN


def _slot_cases():
    with open(os.path.join(GOLDEN, "slots.json")) as fh:
        return json.load(fh)


def test_size_embedding_constants_match_reference():
    """RANGE_SIZE / SIZE_EMB_DICT of the imported deepfakes_dataset module (tools/make_golden.py slots_case) against the oracle's and
    the product's tables: 20 buckets, (0..5), (6..10), ... (96..100)."""
    with open(os.path.join(GOLDEN, "f1_constants.json")) as fh:
        ref = json.load(fh)
    assert [list(t) for t in O.SIZE_EMB_DICT] == ref["SIZE_EMB_DICT"] and O._RANGE_SIZE == ref["RANGE_SIZE"]
    assert len(ref["SIZE_EMB_DICT"]) == 20 and ref["SIZE_EMB_DICT"][0] == [0, 5] and ref["SIZE_EMB_DICT"][-1] == [96, 100]
    # the oracle's bucket rule agrees with the reference's table on every integer ratio 0..100 (the device builder is compared with
    # the oracle bit for bit in the GPU test below)
    for ratio in range(101):
        want = next(i for i, (a, b) in enumerate(ref["SIZE_EMB_DICT"]) if a <= ratio <= b) + 1
        assert O.size_bucket(ratio, 1, 200, 1, "predict") == want, ratio      # face area = ratio, video area = 200 * 1 / 2 = 100


def test_oracle_slot_assignment_matches_reference():
    for r in _slot_cases():
        s = O.sort_identities([(i, 0, c) for i, c in enumerate(r["counts"])], 1, r["max_identities"])
        assert [x[0] for x in s] == r["order"], r
        assert O.assign_slots([x[2] for x in s], r["num_frames"]) == r["slots"], r


def test_product_slot_assignment_matches_reference():
    for r in _slot_cases():
        ids = [S.Identity(str(i), [(3 * k, 10, 10) for k in range(c)]) for i, c in enumerate(r["counts"])]
        plan = S.plan_clip(ids, r["num_frames"], (1280, 720), max_identities=r["max_identities"], ordering=1)
        assert [int(i.name) for i in plan.identities] == r["order"], r
        assert [i.slots for i in plan.identities] == r["slots"], r
        assert sum(i.slots for i in plan.identities) == r["num_frames"]
        for ident, faces in zip(plan.identities, plan.chosen):
            assert len(faces) == min(len(ident.faces), ident.slots)
            assert [f[0] for f in faces] == sorted(f[0] for f in faces)


def test_hand_derived_clip_tensors():
    # 8 slots, 2 identities [4,4]; identity A has 4 faces at video frames 0,10,20,30; identity B only 3 (frames 10,20,40)
    ids = [dict(slots=4, faces=[(0, 200, 200), (10, 200, 200), (20, 200, 200), (30, 200, 200)]),
           dict(slots=4, faces=[(10, 100, 100), (20, 100, 100), (40, 100, 100)])]
    size, mask, im, pos = O.build_clip_tensors(ids, 8, 49, video_wh=(1000, 800), variant="predict")
    # predict.py:291: face area 200*200 = 40000, video area 1000*800/2 = 400000 -> ratio 10 -> bucket (6..10) = 2;
    # 100*100 -> ratio 2 -> bucket 1; padded slot -> 0
    assert size.tolist() == [2, 2, 2, 2, 1, 1, 1, 0] and size.dtype == torch.int32
    assert mask.tolist() == [True] * 7 + [False]
    blk = torch.zeros(8, 8, dtype=torch.bool)
    blk[:4, :4] = True
    blk[4:, 4:] = True
    assert torch.equal(im, blk)
    # distinct frames {0,10,20,30,40} -> ranks 1..5; the padded slot takes max(frames so far) = 40 -> rank 5
    ranks = [1, 2, 3, 4, 2, 3, 5, 5]
    want = [0] + [(p - 1) * 49 + 1 + j for p in ranks for j in range(49)]
    assert pos.tolist() == want and pos.dtype == torch.int64
    # dataset variant: face area halved (deepfakes_dataset.py:259) and -- faithful quirk -- the mask stays all ones (:281)
    size_d, mask_d, _, _ = O.build_clip_tensors(ids, 8, 49, video_wh=(1000, 800), variant="dataset")
    assert size_d.tolist() == [1, 1, 1, 1, 1, 1, 1, 0]          # ratios 5 and 1
    assert mask_d.tolist() == [True] * 8
    # bucket edges: ratio 0..5 -> 1, 6 -> 2, 100 -> 20, 101 -> IndexError like the reference's np.where(...)[0][0]
    assert [O.size_bucket(r * 10, 10, 100, 200, "predict") for r in (0, 5, 6, 10, 11, 100)] == [1, 1, 2, 2, 3, 20]
    with pytest.raises(IndexError):
        O.size_bucket(1010, 10, 100, 200, "predict")


def test_select_faces_rule():
    assert O.select_faces(3, 4) == [0, 1, 2]
    assert O.select_faces(10, 4, index=1) == [0, 3, 5, 8]          # round(linspace(0, 8, 4)) = 0, 2.67, 5.33, 8
    assert O.select_faces(10, 4, index=2) == [1, 4, 6, 9]          # round(linspace(1, 9, 4))
    assert O.select_faces(10, 4, index=2, variant="predict") == [0, 3, 5, 8]


def _random_plans(rng, B, F, variant):
    plans, oracle_ids = [], []
    for b in range(B):
        n_id = int(rng.integers(1, 5))
        counts = rng.choice(np.arange(1, 2 * F), size=n_id, replace=False)
        ids = []
        for i, c in enumerate(counts):
            frs = rng.choice(np.arange(0, 300), size=int(c), replace=False)
            ids.append(S.Identity(f"id{i}", [(int(fr), int(rng.integers(20, 300)), int(rng.integers(20, 300))) for fr in frs]))
        plan = S.plan_clip(ids, F, (1280, 720), max_identities=int(rng.integers(1, 5)), ordering=int(rng.integers(0, 2)),
                           sample_index=b, variant=variant)
        plans.append(plan)
        oracle_ids.append([dict(slots=i.slots, faces=ch) for i, ch in zip(plan.identities, plan.chosen)])
    return plans, oracle_ids


@pytest.mark.gpu
@pytest.mark.parametrize("F,variant", [(8, "dataset"), (16, "predict"), (32, "predict"), (8, "predict")])
def test_device_builder_is_bit_exact_vs_oracle(F, variant):
    rng = np.random.default_rng(F * 7 + len(variant))
    B = 37
    plans, oracle_ids = _random_plans(rng, B, F, variant)
    out = S.build_batch(plans, F, 49, device="cuda")
    assert out["size_embedding"].device.type == "cpu" and out["size_embedding"].dtype == torch.int32
    for b in range(B):
        size, mask, im, pos = O.build_clip_tensors(oracle_ids[b], F, 49, video_wh=(1280, 720), variant=variant)
        assert torch.equal(out["size_embedding"][b], size), b
        assert torch.equal(out["mask"][b].cpu(), mask), b
        assert torch.equal(out["identities_mask"][b].cpu(), im), b
        assert torch.equal(out["positions"][b].cpu(), pos), b
    assert out["mask"].dtype == torch.bool and out["positions"].dtype == torch.int64


@pytest.mark.gpu
def test_built_inputs_drive_the_model():
    """The builder's outputs are directly consumable by SizeInvariantTimeSformer.forward (same dtypes/devices as the loader's)."""
    from mintime_amd import arch, synth, SizeInvariantTimeSformer
    rng = np.random.default_rng(5)
    F, B = 8, 3
    plans, oracle_ids = _random_plans(rng, B, F, "predict")
    out = S.build_batch(plans, F, 49, device="cuda")
    cfg = arch.default_tsf_config(1280, F)
    model = SizeInvariantTimeSformer(config=cfg)
    sd = synth.tsf_state(cfg, 0)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    feats = synth.features(B, F, 1280, 0)
    with torch.no_grad():
        y = model(feats.cuda(), mask=out["mask"], identities_mask=out["identities_mask"], size_embedding=out["size_embedding"],
                  positions=out["positions"])
        yo = O.tsf_forward(sd, cfg, feats, out["mask"].cpu(), out["identities_mask"].cpu(), out["size_embedding"],
                           out["positions"].cpu())
    assert float((y.cpu() - yo).abs().max()) <= 1e-3 * float(yo.abs().max())


def test_ordering_by_face_size_uses_the_crop_width_and_the_all_or_nothing_fallback():
    """identities_ordering = 0 (deepfakes_dataset.py:113,142-143): identities sort by the mean of groups()[0] of libmagic's
    "W x H" string, i.e. the crop WIDTH, and an identity with any unparsable file counts as 0.  Hand-derived (python-magic is
    not installed here and may not be faked, so this rule is restated, not pinned): faces are (frame, height, width)."""
    tall = S.Identity("tall", [(0, 300, 100), (3, 300, 100)])          # mean width 100, mean height 300
    wide = S.Identity("wide", [(0, 120, 200), (3, 120, 200)])          # mean width 200, mean height 120
    assert tall.mean_side == 100.0 and wide.mean_side == 200.0
    plan = S.plan_clip([tall, wide], 8, (1280, 720), max_identities=2, ordering=0)
    assert [i.name for i in plan.identities] == ["wide", "tall"]      # by width; by height the order would be the opposite
    plan = S.plan_clip([tall, wide], 8, (1280, 720), max_identities=1, ordering=0)
    assert [i.name for i in plan.identities] == ["wide"]              # ... and truncation follows that order
    broken = S.Identity("broken", [(0, 500, 500), (3, 500, -1)])       # one file failed to parse -> the whole identity is 0
    assert broken.mean_side == 0.0
    plan = S.plan_clip([broken, tall], 8, (1280, 720), max_identities=2, ordering=0)
    assert [i.name for i in plan.identities] == ["tall", "broken"]
