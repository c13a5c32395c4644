"""CPU tests: the oracle (oracle/dadet_oracle.c, oracle/model_ref.py) against
  (a) the reference's own known-answer vectors (tests/golden/reference_known_answers.json),
  (b) outputs of the reference's compiled CPU operators (tests/golden/ref_ops.npz; live oracle/_ref when present),
  (c) loss dictionaries / intermediates of the imported Python reference (tests/golden/da_*.npz).
"""
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def test_nms_reference_known_answers():
    from oracle import ops as O

    ka = json.load(open(os.path.join(GOLD, "reference_known_answers.json")))
    assert len(ka["nms"]) == 6
    for case in ka["nms"]:
        keep = O.nms(np.array(case["boxes"], np.float32), np.array(case["scores"], np.float32), case["thresh"], 0)
        assert np.array_equal(np.sort(keep), np.array(case["keep"])), case["thresh"]


def test_box_decode_reference_known_answer():
    from oracle import model_ref

    ka = json.load(open(os.path.join(GOLD, "reference_known_answers.json")))
    for case in ka["box_decode"]:
        got = model_ref.decode(torch.tensor(case["deltas"]), torch.tensor(case["boxes"]), case["weights"]).numpy()
        np.testing.assert_allclose(got, np.array(case["expected"], np.float32), atol=1e-4)


def test_ops_match_reference_build_fixture():
    from oracle import ops as O

    z = np.load(os.path.join(GOLD, "ref_ops.npz"))
    for ph, sr in ((7, 0), (14, 0), (7, 2)):
        got = O.roi_align_forward(z["roi/input"], z["roi/rois"], 1 / 16.0, ph, ph, sr)
        assert np.array_equal(got, z["roi/out_%d_%d" % (ph, sr)])  # bit exact
    for thr in (0.3, 0.5, 0.7):
        assert np.array_equal(O.nms(z["nms/boxes"], z["nms/scores"], thr, 0), z["nms/keep_%.1f" % thr])


def test_ops_match_live_reference_build():
    from oracle import build_ref
    from oracle import ops as O

    ref = build_ref.load()
    if ref is None:
        pytest.skip("oracle/_ref/ref_C.so not built")
    rng = np.random.default_rng(7)
    x = rng.standard_normal((2, 6, 17, 23)).astype(np.float32)
    rois = np.concatenate([rng.integers(0, 2, (30, 1)), rng.uniform(-10, 200, (30, 2)),
                           rng.uniform(150, 420, (30, 2))], 1).astype(np.float32)
    want = ref.roi_align_forward(torch.from_numpy(x), torch.from_numpy(rois), 1 / 16.0, 7, 7, 0).numpy()
    assert np.array_equal(O.roi_align_forward(x, rois, 1 / 16.0, 7, 7, 0), want)
    boxes = np.concatenate([rng.uniform(0, 300, (800, 2)), rng.uniform(300, 500, (800, 2))], 1).astype(np.float32)
    scores = (rng.permutation(800) / 800.0).astype(np.float32)
    assert np.array_equal(O.nms(boxes, scores, 0.6, 0),
                          ref.nms(torch.from_numpy(boxes), torch.from_numpy(scores), 0.6).numpy())


def test_roi_align_backward_is_adjoint_of_forward():
    """<ROIAlign(x), g> == <x, ROIAlign^T(g)> — the reference has no CPU backward, so the restated backward is
    validated as the exact adjoint of the (reference-pinned) forward, in float64-accumulated dot products."""
    from oracle import ops as O

    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 5, 14, 19)).astype(np.float32)
    rois = np.concatenate([rng.integers(0, 2, (12, 1)), rng.uniform(-5, 150, (12, 2)),
                           rng.uniform(100, 330, (12, 2))], 1).astype(np.float32)
    g = rng.standard_normal((12, 5, 7, 7)).astype(np.float32)
    y = O.roi_align_forward(x, rois, 1 / 16.0, 7, 7, 0)
    gx = O.roi_align_backward(g, rois, 1 / 16.0, 7, 7, 2, 5, 14, 19, 0)
    lhs = float((y.astype(np.float64) * g).sum())
    rhs = float((x.astype(np.float64) * gx).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)


@pytest.mark.parametrize("case", ["da_plain", "da_triplet_aligned"])
def test_model_ref_reproduces_reference_losses(case):
    """oracle/model_ref.py on the seeded inputs of the fixture == the imported reference's loss dict."""
    from da_detect_amd.config import cfg
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.modeling.detector import build_detection_model
    from golden.fill import fill_state_dict
    from golden.cases import case_cfg
    from oracle import model_ref

    z = np.load(os.path.join(GOLD, case + ".npz"))
    c = case_cfg(case)
    seed, H, W, nimg = int(z["seed"]), int(z["H"]), int(z["W"]), int(z["nimg"])
    sd = fill_state_dict(build_detection_model(c).state_dict(), seed)
    images, targets = make_batch(c, nimg, H, W, seed=seed, device=torch.device("cpu"))
    inter = {}
    torch.manual_seed(seed)
    # proposal selection is fed the fixture's RPN maps: this host's CPU GEMMs may round differently from the
    # authoring host's and flip near-tied scores; the maps themselves are compared below
    losses = model_ref.training_losses(sd, c, images.tensors, model_ref.targets_to_dicts(targets), state={},
                                       intermediates=inter,
                                       selection_maps=(torch.from_numpy(z["objectness"]), torch.from_numpy(z["deltas"])))
    want = {k[5:]: float(z[k]) for k in z.files if k.startswith("loss/")}
    assert set(losses) == set(want)
    for k, v in want.items():
        assert abs(float(losses[k]) - v) <= 1e-4 * max(abs(v), 1.0), (k, float(losses[k]), v)
    np.testing.assert_allclose(inter["objectness"].numpy(), z["objectness"], rtol=1e-5, atol=1e-5)
    nprop = sum(1 for k in z.files if k.startswith("proposals/") and k.endswith("/boxes"))
    assert nprop == min(nimg, 2)  # the box head sees [source, target] only, also in triplet mode
    for i in range(nprop):
        np.testing.assert_allclose(inter["proposals"][i][0].numpy(), z["proposals/%d/boxes" % i], atol=1e-3)


def test_roi_pool_oracle_properties():
    """the ROIPool restatement has no CPU reference to run against (ROIPool.h:20-22): pin it by properties"""
    from oracle import ops as O

    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 6, 12, 20)).astype(np.float32)
    rois = np.array([[0, 0, 0, 19 * 16, 11 * 16],      # the whole map, exactly divisible by 2x2 bins
                     [1, 48, 32, 48, 32],               # a single cell
                     [1, -400, -400, -300, -300]], dtype=np.float32)   # outside: empty
    out, arg = O.roi_pool_forward(x, rois, 1 / 16.0, 2, 2)
    want = torch.nn.functional.adaptive_max_pool2d(torch.from_numpy(x[0]), 2).numpy()
    assert np.array_equal(out[0], want)
    assert np.array_equal(out[1], np.broadcast_to(x[1, :, 2, 3][:, None, None], (6, 2, 2)))
    assert np.all(arg[1] == 2 * 20 + 3)
    assert np.all(out[2] == 0) and np.all(arg[2] == -1)
    g = np.ones_like(out)
    gin = O.roi_pool_backward(g, arg, rois, 2, 6, 12, 20)
    assert gin.sum() == (arg >= 0).sum() and gin[1, :, 2, 3].tolist() == [4.0] * 6


def test_model_ref_inference_reproduces_reference_detections():
    """eval path: oracle/model_ref.inference vs the imported reference's detections (tests/golden/eval_da_plain.npz)"""
    from da_detect_amd.data.synthetic import make_batch
    from golden.cases import case_cfg
    from golden.fill import fill_state_dict
    from oracle import model_ref

    z = np.load(os.path.join(GOLD, "eval_da_plain.npz"))
    ref_keys = json.load(open(os.path.join(GOLD, "reference_state_dict_keys.json")))["da_plain"]
    c = case_cfg("da_plain")
    sd = fill_state_dict({k: torch.empty(v) for k, v in ref_keys.items()}, int(z["seed"]))
    images, _ = make_batch(c, int(z["nimg"]), int(z["H"]), int(z["W"]), seed=int(z["seed"]),
                           device=torch.device("cpu"))
    inter = {}
    dets = model_ref.inference(sd, c, images.tensors, inter)
    np.testing.assert_allclose(inter["class_logits"].numpy(), z["class_logits"], rtol=1e-4, atol=1e-5)
    for i, d in enumerate(dets):
        assert np.array_equal(d["labels"].numpy(), z["det/%d/labels" % i])
        np.testing.assert_allclose(d["boxes"].numpy(), z["det/%d/boxes" % i], atol=2e-3)
        np.testing.assert_allclose(d["scores"].numpy(), z["det/%d/scores" % i], atol=1e-6)
        np.testing.assert_allclose(inter["proposals"][i][0].numpy(), z["proposals/%d/boxes" % i], atol=1e-3)


def test_model_ref_inference_fpn_reproduces_reference_detections():
    """FPN eval path (FPN, 5-level RPN, LevelMapper pooling, FPN2MLP head) vs tests/golden/eval_fpn.npz"""
    from da_detect_amd.data.synthetic import make_batch
    from golden.cases import case_cfg
    from golden.fill import fill_state_dict
    from oracle import model_ref

    z = np.load(os.path.join(GOLD, "eval_fpn.npz"))
    ref_keys = json.load(open(os.path.join(GOLD, "reference_state_dict_keys.json")))["fpn"]
    c = case_cfg("fpn")
    sd = fill_state_dict({k: torch.empty(v) for k, v in ref_keys.items()}, int(z["seed"]))
    images, _ = make_batch(c, int(z["nimg"]), int(z["H"]), int(z["W"]), seed=int(z["seed"]),
                           device=torch.device("cpu"))
    inter = {}
    dets = model_ref.inference_fpn(sd, c, images.tensors, inter)
    assert len(torch.unique(inter["levels"])) >= 3          # the LevelMapper spreads ROIs over the pyramid
    for l in range(5):
        np.testing.assert_allclose(inter["objectness"][l].numpy(), z["objectness/%d" % l], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(inter["class_logits"].numpy(), z["class_logits"], rtol=1e-4, atol=1e-5)
    for i, d in enumerate(dets):
        assert np.array_equal(d["labels"].numpy(), z["det/%d/labels" % i])
        np.testing.assert_allclose(d["boxes"].numpy(), z["det/%d/boxes" % i], atol=2e-3)
        np.testing.assert_allclose(d["scores"].numpy(), z["det/%d/scores" % i], atol=1e-6)


def test_device_draw_restatement_follows_the_sampler_rule():
    """oracle/model_ref.sample_pos_neg_device restates the PRODUCT's random draw (csrc/sampling.hip) so that the default
    GPU path can be compared with the oracle on the same sample (tests/test_default_path_gpu.py).  Here: the key is the
    scalar splitmix64 finaliser, and the subsets obey the reference sampler's rule
    (balanced_positive_negative_sampler.py:27-76) — counts per class, ignored rows never taken, smallest keys win, ties
    to the lower index."""
    from oracle import model_ref

    def key(seed, i):
        m = (1 << 64) - 1
        z = (seed + (i + 1) * 0x9E3779B97F4A7C15) & m
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
        return (z ^ (z >> 31)) >> 32

    seed = 0xDEADBEEFCAFEBABE
    keys = model_ref.device_sample_keys(seed, 500)
    assert all(int(keys[i]) == key(seed, i) for i in range(500))
    g = torch.Generator().manual_seed(4)
    labels = torch.randint(-1, 4, (5000,), generator=g)
    pm, nm = model_ref.sample_pos_neg_device(labels, 256, 0.25, seed)
    assert int(pm.sum()) == 64 and int(nm.sum()) == 192
    assert bool((labels[pm] >= 1).all()) and bool((labels[nm] == 0).all()) and not bool((pm & nm).any())
    all_keys = model_ref.device_sample_keys(seed, 5000).astype(np.int64)
    pos = torch.nonzero(labels >= 1).squeeze(1).numpy()
    assert all_keys[pm.numpy()].max() <= np.sort(all_keys[pos])[64 - 1]           # the 64 smallest keys of the class
    few = torch.tensor([1, 0, -1, 0, 1, 0])                                       # fewer candidates than the quota
    pm, nm = model_ref.sample_pos_neg_device(few, 256, 0.25, 7)
    assert pm.tolist() == [True, False, False, False, True, False] and nm.tolist() == [False, True, False, True, False, True]


def test_model_ref_accepts_proposal_lists():
    """training_losses(selection_proposals=...): the proposal lists handed in replace the restatement's own selection (what
    tests/test_default_path_gpu.py uses to compare the GPU path on IDENTICAL lists: sigmoid-tied neighbours may swap
    between devices).  Own lists handed back in -> same losses; two neighbours swapped -> still the same SET, another
    order, and the sampled boxes follow the list."""
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.modeling.detector import build_detection_model
    from golden.cases import case_cfg
    from golden.fill import fill_state_dict
    from oracle import model_ref

    case = "da_plain"
    z = np.load(os.path.join(GOLD, case + ".npz"))
    c = case_cfg(case)
    seed, H, W, nimg = int(z["seed"]), int(z["H"]), int(z["W"]), int(z["nimg"])
    sd = fill_state_dict(build_detection_model(c).state_dict(), seed)
    images, targets = make_batch(c, nimg, H, W, seed=seed, device=torch.device("cpu"))
    gts = model_ref.targets_to_dicts(targets)
    maps = (torch.from_numpy(z["objectness"]), torch.from_numpy(z["deltas"]))

    def run(props):
        inter = {}
        torch.manual_seed(seed)
        with torch.no_grad():
            losses = model_ref.training_losses(sd, c, images.tensors, gts, state={}, intermediates=inter,
                                               selection_maps=maps, selection_proposals=props)
        return {k: float(v) for k, v in losses.items()}, inter

    base, inter = run(None)
    own = [(b.clone(), s.clone()) for b, s in inter["proposals"]]
    again, _ = run(own)
    assert again == base
    only_first, _ = run(own[:1])                       # a shorter list covers the leading images
    assert only_first == base
    swapped = [(b.clone(), s.clone()) for b, s in own]
    b1, s1 = swapped[1]
    b1[[3, 4]] = b1[[4, 3]]
    s1[[3, 4]] = s1[[4, 3]]
    other, inter2 = run(swapped)
    assert torch.equal(inter2["proposals"][1][0][3], own[1][0][4]) and torch.equal(inter2["proposals"][0][0], own[0][0])
    assert set(other) == set(base)
