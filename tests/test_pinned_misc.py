"""Fixtures recorded from the imported reference by tests/golden/make_golden_misc.py:
focal loss (layers/sigmoid_focal_loss.py:40-52), training-mode multi-level proposal selection + five-level RPN losses
(rpn/inference.py:124-181, rpn/loss.py:101-143), adaptive triplet margins (da_heads/loss.py:180-222)."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


# ---------------------------------------------------------------------------------------------------- focal loss
def test_oracle_focal_loss_matches_the_reference_python_implementation():
    """the C oracle restates the CUDA kernel (SigmoidFocalLoss_cuda.cu:21-101); on |x| <= 4 its FLT_MIN clamp is inactive
    and the reference's plain log(1 - p) is still accurate, so both must agree with `sigmoid_focal_loss_cpu`"""
    from oracle import ops as O

    z = np.load(os.path.join(GOLD, "focal_ref.npz"))
    for i in range(3):
        gamma, alpha = float(z["case%d/gamma" % i]), float(z["case%d/alpha" % i])
        got = O.sigmoid_focal_loss_forward(z["logits"], z["targets"], gamma, alpha)
        np.testing.assert_allclose(got, z["case%d/loss" % i], rtol=2e-5, atol=3e-6)
        gotb = O.sigmoid_focal_loss_backward(z["logits"], z["targets"], z["case%d/d_losses" % i], gamma, alpha)
        np.testing.assert_allclose(gotb, z["case%d/d_logits" % i], rtol=2e-5, atol=3e-6)
    assert (z["targets"] == -1).any() and (z["targets"] == 0).any() and (z["targets"] == 8).any()


@pytest.mark.gpu
def test_hip_focal_loss_matches_the_reference_python_implementation(device):
    from da_detect_amd import _C

    z = np.load(os.path.join(GOLD, "focal_ref.npz"))
    logits = torch.from_numpy(z["logits"]).to(device)
    targets = torch.from_numpy(z["targets"]).to(device)
    for i in range(3):
        gamma, alpha = float(z["case%d/gamma" % i]), float(z["case%d/alpha" % i])
        got = _C.sigmoid_focalloss_forward(logits, targets, logits.shape[1], gamma, alpha)
        np.testing.assert_allclose(got.cpu().numpy(), z["case%d/loss" % i], rtol=2e-5, atol=3e-6)
        d = torch.from_numpy(z["case%d/d_losses" % i]).to(device)
        gotb = _C.sigmoid_focalloss_backward(logits, targets, d, logits.shape[1], gamma, alpha)
        np.testing.assert_allclose(gotb.cpu().numpy(), z["case%d/d_logits" % i], rtol=2e-5, atol=3e-6)


# -------------------------------------------------------------------------- FPN, training-mode selection + RPN losses
def _fpn_train_inputs(z, device):
    from da_detect_amd.data.synthetic import make_batch
    from golden.cases import case_cfg

    c = case_cfg("fpn")
    images, targets = make_batch(c, int(z["nimg"]), int(z["H"]), int(z["W"]), seed=int(z["seed"]), device=device)
    for t in targets:
        t.add_field("is_source", torch.ones_like(t.get_field("is_source")))
    return c, images, targets


def test_oracle_fpn_training_selection_and_rpn_losses():
    from oracle import model_ref

    z = np.load(os.path.join(GOLD, "fpn_train_rpn.npz"))
    c, images, targets = _fpn_train_inputs(z, torch.device("cpu"))
    obj = [torch.from_numpy(z["objectness/%d" % l]) for l in range(5)]
    dlt = [torch.from_numpy(z["deltas/%d" % l]) for l in range(5)]
    gts = model_ref.targets_to_dicts(targets)
    sizes = [(int(z["H"]), int(z["W"]))] * int(z["nimg"])
    props = model_ref.rpn_proposals_fpn_train(obj, dlt, sizes, gts, c)
    total = 0
    for i, (b, s) in enumerate(props):
        assert tuple(b.shape) == z["proposals/%d/boxes" % i].shape
        np.testing.assert_allclose(b.numpy(), z["proposals/%d/boxes" % i], atol=1e-3)
        np.testing.assert_allclose(s.numpy(), z["proposals/%d/objectness" % i], rtol=1e-6, atol=1e-7)
        total += len(b) - len(gts[i]["boxes"])
    assert total == c.MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN, "the batch-wide top-k must have cut the proposal set"
    torch.manual_seed(int(z["seed"]))
    lo, lb = model_ref.rpn_losses(obj, dlt, torch.cat(model_ref.fpn_anchors(obj, c), 0), sizes, gts, c)
    assert abs(float(lo) - float(z["loss/loss_objectness"])) <= 1e-5
    assert abs(float(lb) - float(z["loss/loss_rpn_box_reg"])) <= 1e-5


@pytest.mark.gpu
def test_hip_fpn_training_selection_and_rpn_losses(device):
    """RPNModule in training mode over the pyramid: proposals of select_over_all_levels (batch-wide top-k) + GT boxes
    given the reference's per-level maps, and the five-level RPN losses on this model's own maps"""
    from da_detect_amd.modeling.detector import build_detection_model
    from da_detect_amd.utils import rng
    from golden.fill import fill_state_dict

    z = np.load(os.path.join(GOLD, "fpn_train_rpn.npz"))
    c, images, targets = _fpn_train_inputs(z, device)
    model = build_detection_model(c)
    model.load_state_dict(fill_state_dict(model.state_dict(), int(z["seed"])))
    model = model.to(device).train()
    selector = model.rpn.box_selector_train
    orig = selector.forward
    gold_obj = [torch.from_numpy(z["objectness/%d" % l]).to(device) for l in range(5)]
    gold_del = [torch.from_numpy(z["deltas/%d" % l]).to(device) for l in range(5)]
    selector.forward = lambda anchors, objectness, box_regression, tg=None: orig(anchors, gold_obj, gold_del, tg)
    captured = {}
    model.rpn.head.register_forward_hook(lambda m, i, o: captured.update(objectness=[t.detach() for t in o[0]]))
    rng.use_cpu_stream(True)
    try:
        torch.manual_seed(int(z["seed"]))
        with torch.no_grad():
            feats = model.backbone(images.tensors)
            proposals, losses = model.rpn(images, feats, targets)
    finally:
        rng.use_cpu_stream(False)
        selector.forward = orig
    for l in range(5):
        np.testing.assert_allclose(captured["objectness"][l].cpu().numpy(), z["objectness/%d" % l], rtol=1e-4, atol=1e-4)
    for i, p in enumerate(proposals):
        assert len(p) == len(z["proposals/%d/boxes" % i]), (i, len(p))
        np.testing.assert_allclose(p.bbox.cpu().numpy(), z["proposals/%d/boxes" % i], atol=2e-4)
        np.testing.assert_allclose(p.get_field("objectness").cpu().numpy(), z["proposals/%d/objectness" % i],
                                   rtol=2e-6, atol=1e-7)
    for k in ("loss_objectness", "loss_rpn_box_reg"):
        want = float(z["loss/" + k])
        assert abs(float(losses[k]) - want) <= 1e-4 * max(1.0, abs(want)), (k, float(losses[k]), want)


# ---------------------------------------------------------------------------------------- adaptive triplet margins
@pytest.mark.gpu
def test_adaptive_triplet_margin_trajectory_matches_the_reference(device):
    """TripletMargins: the image margin grows by lr after every exactly-zero loss (until int(margin) == int(max)), stays
    after a non-zero one; the instance margin is fixed (adaptive=False)"""
    from da_detect_amd.modeling.da_heads.loss import TripletMargins

    z = np.load(os.path.join(GOLD, "triplet_margin.npz"))
    ev = TripletMargins()
    a, p = (torch.from_numpy(z[k]).to(device) for k in ("a", "p"))
    negs = [torch.from_numpy(z["n"]).to(device), torch.from_numpy(z["n_hard"]).to(device)]
    prev = 1
    assert (z["img"][1:4, 1] > z["img"][0:3, 1]).all() and z["img"][4, 1] == z["img"][3, 1]   # grows, pauses after 0.97
    for it, which in enumerate(z["order"]):
        loss = ev.triplet_img_loss(a, p, negs[int(which)], prev, adaptive=True, lr=0.001, max_margin=3.0, margin=1.0)
        want_prev, want_margin, want_loss = z["img"][it]
        assert float(prev) == pytest.approx(want_prev, rel=1e-5, abs=1e-7)
        assert ev.margin_img == pytest.approx(want_margin, rel=0, abs=1e-12), (it, ev.margin_img, want_margin)
        assert float(loss) == pytest.approx(want_loss, rel=1e-5, abs=1e-7)
        prev = loss.detach().cpu()
    ia, ip, ineg = (torch.from_numpy(z[k]).to(device) for k in ("ia", "ip", "ineg"))
    prev = 1
    for it in range(3):
        loss = ev.triplet_ins_loss(ia, ip, ineg, prev, adaptive=False, lr=0.001, max_margin=3.0, margin=0.7)
        assert ev.margin_ins == pytest.approx(z["ins"][it][0]) and float(loss) == pytest.approx(z["ins"][it][1], rel=1e-5)
        prev = loss.detach().cpu()
