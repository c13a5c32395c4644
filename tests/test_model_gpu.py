"""Parity of the full HIP-backed model against the golden fixtures made from the imported reference
(tests/golden/da_*.npz), plus gradient parity against the CPU oracle and optimizer / property checks.

Tolerances: losses 1e-4 relative (north-star bar: "fp32 losses within 1e-4"); proposal boxes 1e-3 px
(device expf vs libm); feature / logit tensors 1e-4 of their scale (fp32 MFMA accumulation order differs from the
CPU GEMMs)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _build(case, device):
    from da_detect_amd.modeling.detector import build_detection_model
    from golden.cases import case_cfg
    from golden.fill import fill_state_dict

    z = np.load(os.path.join(GOLD, case + ".npz"))
    c = case_cfg(case)
    model = build_detection_model(c)
    sd = fill_state_dict(model.state_dict(), int(z["seed"]))
    model.load_state_dict(sd)
    return z, c, model.to(device).train(), sd


def _run_with_golden_rpn_selection(model, z, images, targets, seed, device, inject=True):
    """training-mode forward with the CPU random stream; optionally the proposal SELECTION (sort / decode / NMS —
    the discontinuous part of the path) is fed the reference's RPN maps from the fixture, so that index-valued
    results can be compared exactly instead of through fp32 noise on near-tied scores.  Everything continuous
    (backbone, RPN head, RPN losses, box head, DA heads) still runs on this model's own tensors."""
    from da_detect_amd.utils import rng

    captured = {}
    model.backbone.register_forward_hook(lambda m, i, o: captured.__setitem__("feat", o[0].detach()))
    model.rpn.head.register_forward_hook(
        lambda m, i, o: captured.update(objectness=o[0][0].detach(), deltas=o[1][0].detach()))
    selector = model.rpn.box_selector_train
    orig_sel = selector.forward
    if inject:
        gold_obj = [torch.from_numpy(z["objectness"]).to(device)]
        gold_del = [torch.from_numpy(z["deltas"]).to(device)]
        selector.forward = lambda anchors, objectness, box_regression, tg=None: orig_sel(anchors, gold_obj, gold_del, tg)
    evaluator = model.roi_heads.box.loss_evaluator
    orig_sub = evaluator.subsample

    def spy(proposals, tg):
        captured.setdefault("proposals", [(p.bbox.clone(), p.get_field("objectness").clone()) for p in proposals])
        out = orig_sub(proposals, tg)
        captured.setdefault("sampled", [p.bbox.clone() for p in out])
        return out

    evaluator.subsample = spy
    rng.use_cpu_stream(True)
    try:
        torch.manual_seed(seed)
        losses = model(images, targets)
    finally:
        rng.use_cpu_stream(False)
        selector.forward = orig_sel
        evaluator.subsample = orig_sub
    return losses, captured


@pytest.mark.parametrize("case", ["da_plain", "da_img_only", "da_triplet", "da_triplet_aligned"])
def test_losses_match_reference_golden(device, case):
    from da_detect_amd.data.synthetic import make_batch

    z, c, model, _ = _build(case, device)
    seed, H, W, nimg = int(z["seed"]), int(z["H"]), int(z["W"]), int(z["nimg"])
    images, targets = make_batch(c, nimg, H, W, seed=seed, device=device)
    losses, captured = _run_with_golden_rpn_selection(model, z, images, targets, seed, device, inject=True)
    want = {k[5:]: float(z[k]) for k in z.files if k.startswith("loss/")}
    assert set(losses) == set(want), (sorted(losses), sorted(want))
    # continuous stages: backbone and RPN head against the reference's tensors
    feat = captured["feat"].cpu()
    np.testing.assert_allclose(feat[:, ::64, ::3, ::3].numpy(), z["feat_sample"], rtol=1e-4,
                               atol=1e-4 * float(z["feat_absmean"]))
    np.testing.assert_allclose(captured["objectness"].cpu().numpy(), z["objectness"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(captured["deltas"].cpu().numpy(), z["deltas"], rtol=1e-4, atol=1e-4)
    # index-valued stages given identical RPN maps: same proposals in the same order, same sampled ROIs
    for i, (b, s) in enumerate(captured["proposals"]):
        assert tuple(b.shape) == z["proposals/%d/boxes" % i].shape, "proposal count differs for image %d" % i
        # device sigmoid may differ from libm by an ulp; the ORDER (hence every index) must be the same
        np.testing.assert_allclose(s.cpu().numpy(), z["proposals/%d/objectness" % i], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(b.cpu().numpy(), z["proposals/%d/boxes" % i], atol=2e-4)  # device expf ulp
    for i, b in enumerate(captured["sampled"]):
        np.testing.assert_allclose(b.cpu().numpy(), z["sampled_boxes/%d" % i], atol=2e-4)
    for k, v in want.items():
        got = float(losses[k])
        assert abs(got - v) <= 1e-4 * max(abs(v), 1.0), (case, k, got, v)


def test_end_to_end_without_injection_is_statistically_close(device):
    """no injection: proposal ranking may flip between near-tied scores (fp32 noise), so only aggregate
    agreement is asserted"""
    from da_detect_amd.data.synthetic import make_batch

    z, c, model, _ = _build("da_plain", device)
    seed, H, W, nimg = int(z["seed"]), int(z["H"]), int(z["W"]), int(z["nimg"])
    images, targets = make_batch(c, nimg, H, W, seed=seed, device=device)
    losses, captured = _run_with_golden_rpn_selection(model, z, images, targets, seed, device, inject=False)
    for i, (b, _) in enumerate(captured["proposals"]):
        assert abs(b.shape[0] - z["proposals/%d/boxes" % i].shape[0]) <= 0.02 * z["proposals/%d/boxes" % i].shape[0] + 2
    for k in ("loss_objectness", "loss_rpn_box_reg", "loss_da_image"):  # independent of the proposal set
        v = float(z["loss/" + k])
        assert abs(float(losses[k]) - v) <= 1e-4 * max(abs(v), 1.0), (k, float(losses[k]), v)
    for k in ("loss_classifier", "loss_box_reg", "loss_da_instance"):
        v = float(z["loss/" + k])
        assert abs(float(losses[k]) - v) <= 0.1 * max(abs(v), 1.0), (k, float(losses[k]), v)


def test_gradients_match_cpu_oracle(device):
    """backward through every HIP kernel (conv dgrad / wgrad, ROIAlign backward, fused DA heads) against torch
    autograd on the oracle (oracle/model_ref.py) in float64, same weights / inputs / random stream, with the ATen sampling
    chain and the reference's random stream instead of the device sampler (30 s of oracle; in the default run since round 6).
    Tolerances.  At 192 x 320 a ReLU whose pre-activation lies within fp32 rounding of zero fires on one side and not on
    the other, and ONE such unit moves its layer's weight gradient by ~1e-3 of its norm and every layer BELOW it by ~1e-4
    (the maps are small: few units share a gradient).  tools/probes/slow_grad_probe.py on this very case
    (profiles/r06_slow_grad_probe_modes_4_0_3.txt): the exact-fp32 MFMA kernels (mode 0) and the default contraction
    (mode 4) flip the SAME units — instance head 8e-4, res4 block 1 2e-4, everything below it 6e-5 .. 8e-5, identical to
    three digits in both modes — while the six-term bf16 mode happens to land on the oracle's side everywhere (max 4e-5).
    It is the decision of an fp32 ReLU against a float64 one, not the contraction.  So: every tensor under the flip bound
    (a wrong tile, a dropped row block, a missing term moves a tensor by O(1)), the MEDIAN tensor at rounding level, and at
    least half of all tensors at rounding level (a flip high in the network taints every tensor below it: a share of
    tensors says nothing about the number of flips)."""
    from da_detect_amd.data.synthetic import make_batch
    from oracle import model_ref

    z, c, model, sd = _build("da_plain", device)
    seed, H, W, nimg = int(z["seed"]), int(z["H"]), int(z["W"]), int(z["nimg"])
    images, targets = make_batch(c, nimg, H, W, seed=seed, device=device)
    losses, _ = _run_with_golden_rpn_selection(model, z, images, targets, seed, device, inject=True)
    sum(losses.values()).backward()
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    # the oracle in FLOAT64: against it the HIP path is at rounding level on every tensor that no flipped ReLU touches
    # (test_default_path_gpu._check_gradients: every tensor < 4e-3, at least 90% of them < 5e-5); the earlier form of
    # this test compared with the fp32 oracle and had to allow 1e-3 in L2 and 2e-2 in the max norm
    osd = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    for n in names:
        osd[n].requires_grad_(True)
    cpu_images, cpu_targets = make_batch(c, nimg, H, W, seed=seed, device=torch.device("cpu"))
    torch.manual_seed(seed)
    olosses = model_ref.training_losses(osd, c, cpu_images.tensors.double(), model_ref.targets_to_dicts(cpu_targets),
                                        selection_maps=(torch.from_numpy(z["objectness"]), torch.from_numpy(z["deltas"])))
    for k in olosses:  # same sampled ROIs on both sides -> same losses
        assert abs(float(olosses[k].detach()) - float(losses[k].detach())) <= 1e-4 * max(abs(float(olosses[k].detach())), 1.0), k
    sum(olosses.values()).backward()
    params = dict(model.named_parameters())
    from test_default_path_gpu import _check_gradients

    got = {n: params[n].grad.detach().cpu() for n in names}
    worst, above = _check_gradients(got, {n: osd[n].grad for n in names}, flipped_share=0.5)
    errs = sorted(float((got[n].double() - osd[n].grad).norm()) / (float(osd[n].grad.norm()) + 1e-30) for n in names)
    assert errs[len(errs) // 2] < 2.5e-5, "median relative L2 gradient error %.2e" % errs[len(errs) // 2]
    print("worst relative L2 gradient error vs the fp64 oracle: %.3e over %d tensors; above rounding level: %s" % (
        worst, len(names), above))


def test_fused_sgd_matches_torch_sgd(device):
    from da_detect_amd.parallel.reducer import BucketedGradReducer
    from da_detect_amd.solver import FusedSGD

    torch.manual_seed(0)
    shapes = [(64, 32, 3, 3), (64,), (10, 7), (5,), (1, 3, 1, 1)]
    ps = [torch.nn.Parameter(torch.randn(s, device=device)) for s in shapes]
    ps[0].data = ps[0].data.contiguous(memory_format=torch.channels_last)
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    groups = lambda xs: [{"params": [x], "lr": 0.01 * (1 + i % 2), "weight_decay": 5e-4 * (i % 2 == 0)}  # noqa: E731
                         for i, x in enumerate(xs)]
    fused = FusedSGD(groups(ps), 0.01, momentum=0.9)
    fused.attach_reducer(BucketedGradReducer(ps, bucket_bytes=4096))
    ref = torch.optim.SGD(groups(qs), 0.01, momentum=0.9)
    for step in range(4):
        fused.zero_grad()
        ref.zero_grad()
        gs = [torch.randn(s, device=device) for s in shapes]
        for i, (p, q, g) in enumerate(zip(ps, qs, gs)):
            if i == 3 or (i == 2 and step < 2):
                continue      # parameter 3 never gets a gradient, parameter 2 only from the third step on:
                              # torch skips `grad is None` parameters (no weight decay, no momentum)
            (p * g).sum().backward()
            (q * g).sum().backward()
        fused.step()
        ref.step()
    for p, q in zip(ps, qs):
        torch.testing.assert_close(p.detach(), q.detach(), rtol=1e-6, atol=1e-7)
    assert torch.equal(ps[3].detach(), qs[3].detach())


# ---- size-independent properties at the BASELINE sizes ------------------------------------------------------
def test_nms_idempotent_and_sorted_at_full_size(device):
    from da_detect_amd import _C

    g = torch.Generator().manual_seed(5)
    n = 12000
    xy = torch.rand((n, 2), generator=g) * torch.tensor([1900.0, 950.0])
    boxes = torch.cat([xy, xy + torch.rand((n, 2), generator=g) * 250 + 8], 1).to(device)
    scores = torch.rand(n, generator=g).to(device)
    keep, cnt = _C.nms_with_count(boxes, scores, 0.7)
    keep = keep[: int(cnt)]
    assert bool((keep[1:] > keep[:-1]).all()), "kept indices must be ascending"
    k2, c2 = _C.nms_with_count(boxes[keep].contiguous(), scores[keep].contiguous(), 0.7)
    assert int(c2) == keep.numel() and torch.equal(k2[: int(c2)], torch.arange(keep.numel(), device=device))
    # no two survivors overlap by >= thr
    b = boxes[keep][:1500]
    area = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    wh = (torch.min(b[:, None, 2:], b[:, 2:]) - torch.max(b[:, None, :2], b[:, :2]) + 1).clamp(min=0)
    iou = wh[..., 0] * wh[..., 1] / (area[:, None] + area - wh[..., 0] * wh[..., 1])
    iou.fill_diagonal_(0)
    assert float(iou.max()) < 0.7


def test_roi_align_constant_and_linearity_at_full_size(device):
    from da_detect_amd import _C

    g = torch.Generator().manual_seed(6)
    R = 512
    xy = torch.rand((R, 2), generator=g) * torch.tensor([1700.0, 680.0])  # ROIs stay inside the 2048x1024 image
    rois = torch.cat([(torch.arange(R) % 2).float().view(-1, 1), xy, xy + torch.rand((R, 2), generator=g) * 300 + 20], 1).to(device)
    const = torch.full((2, 1024, 64, 128), 3.25, device=device)
    out = _C.roi_align_forward(const, rois, 1 / 16.0, 14, 14, 0)
    assert out.shape == (R, 1024, 14, 14)
    torch.testing.assert_close(out, torch.full_like(out, 3.25), rtol=2e-6, atol=2e-6)
    a = torch.randn((2, 256, 64, 128), device=device)
    b = torch.randn((2, 256, 64, 128), device=device)
    ya, yb, yab = [_C.roi_align_forward(t, rois, 1 / 16.0, 14, 14, 0) for t in (a, b, a + b)]
    torch.testing.assert_close(yab, ya + yb, rtol=1e-4, atol=1e-4)
    # backward is the adjoint of forward: <RA(a), g> == <a, RA^T(g)>
    gout = torch.randn_like(ya)
    gin = _C.roi_align_backward(gout, rois, 1 / 16.0, 14, 14, 2, 256, 64, 128, 0)
    lhs, rhs = float((ya.double() * gout.double()).sum()), float((a.double() * gin.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)


def test_conv_linearity_and_adjoint_at_full_size(device):
    """RPN 3x3 conv at the C4 resolution of 1024x2048 inputs: linear in x, and dgrad is its adjoint."""
    from da_detect_amd import _C

    CL = torch.channels_last
    x = torch.randn((2, 1024, 64, 128), device=device).contiguous(memory_format=CL)
    x2 = torch.randn_like(x)
    w = (torch.randn((1024, 1024, 3, 3), device=device) * 0.01).contiguous(memory_format=CL)
    y, y2, y12 = [_C.conv_forward(t, w, pad=1) for t in (x, x2, x + x2)]
    torch.testing.assert_close(y12, y + y2, rtol=1e-3, atol=1e-3)
    g = torch.randn_like(y)
    dx = _C.conv_forward(g, _C.conv_weight_transpose(w), pad=1)
    lhs, rhs = float((y.double() * g.double()).sum()), float((x.double() * dx.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)
    dw = _C.conv_wgrad(x, g, tuple(w.shape), 1, 1)
    lhs2 = float((dw.double() * w.double()).sum())
    assert abs(lhs - lhs2) <= 1e-4 * max(abs(lhs), 1.0)  # <conv(x,w), g> == <w, wgrad(x,g)>


# ---------------------------------------------------------------------------------------------- evaluation path
def _eval_model(device):
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.modeling.detector import build_detection_model
    from golden.cases import case_cfg
    from golden.fill import fill_state_dict

    z = np.load(os.path.join(GOLD, "eval_da_plain.npz"))
    c = case_cfg("da_plain")
    model = build_detection_model(c)
    model.load_state_dict(fill_state_dict(model.state_dict(), int(z["seed"])))
    model = model.to(device).eval()
    images, _ = make_batch(c, int(z["nimg"]), int(z["H"]), int(z["W"]), seed=int(z["seed"]), device=device)
    return z, model, images


def test_eval_detections_match_reference_golden(device):
    """model.eval() forward (generalized_rcnn.py:61-70,145-156 -> box_head.py eval branch -> PostProcessor):
    detections vs the imported reference's (tests/golden/eval_da_plain.npz).  The proposal SELECTION is fed the
    fixture's RPN maps (see _run_with_golden_rpn_selection); box head, softmax, decode, per-class NMS and the
    top-100 cut run on this model's own tensors."""
    z, model, images = _eval_model(device)
    captured = {}
    model.rpn.head.register_forward_hook(
        lambda m, i, o: captured.update(objectness=o[0][0].detach(), deltas=o[1][0].detach()))
    model.roi_heads.box.predictor.register_forward_hook(
        lambda m, i, o: captured.update(class_logits=o[0].detach(), box_regression=o[1].detach()))
    selector = model.rpn.box_selector_test
    orig = selector.forward
    gold_obj = [torch.from_numpy(z["objectness"]).to(device)]
    gold_del = [torch.from_numpy(z["deltas"]).to(device)]
    selector.forward = lambda anchors, objectness, box_regression, tg=None: orig(anchors, gold_obj, gold_del, tg)
    try:
        with torch.no_grad():
            dets = model(images)
    finally:
        selector.forward = orig
    scale = float(np.abs(z["objectness"]).mean())
    assert float((captured["objectness"].cpu() - torch.from_numpy(z["objectness"])).abs().max()) < 1e-4 * max(scale, 1)
    np.testing.assert_allclose(captured["class_logits"].cpu().numpy(), z["class_logits"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(captured["box_regression"].cpu().numpy(), z["box_regression"], rtol=1e-4, atol=2e-5)
    assert len(dets) == int(z["nimg"])
    for i, d in enumerate(dets):
        assert d.mode == "xyxy" and d.size == (int(z["W"]), int(z["H"]))
        labels = d.get_field("labels").cpu().numpy()
        assert labels.dtype == np.int64 and np.array_equal(labels, z["det/%d/labels" % i])
        np.testing.assert_allclose(d.bbox.cpu().numpy(), z["det/%d/boxes" % i], atol=5e-3)
        np.testing.assert_allclose(d.get_field("scores").cpu().numpy(), z["det/%d/scores" % i], atol=2e-5)  # softmax of logits good to 2e-5


@pytest.mark.parametrize("fixture", ["eval_c4_plain_small", "eval_c4_plain_full"])
def test_config1_plain_c4_eval_matches_reference_golden(device, fixture):
    """BASELINE.json configs[0]: e2e_faster_rcnn_R_50_C4_1x.yaml (81 classes, no DA heads, SIZE_DIVISIBILITY 0) on
    2 x 200 x 333 and on its own 2 x 800 x 1333 inputs — nothing is a multiple of 32 (C4 maps 13 x 21 / 50 x 84, every
    stage has an odd extent).  Eval-mode detections of the imported reference (generalized_rcnn.py:134-136;
    tests/golden/make_golden_config1.py); the proposal selection is fed the fixture's RPN maps."""
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.modeling.detector import build_detection_model
    from golden.cases import case_cfg
    from golden.fill import fill_state_dict

    z = np.load(os.path.join(GOLD, fixture + ".npz"))
    c = case_cfg("c4_plain")
    assert c.DATALOADER.SIZE_DIVISIBILITY == 0 and c.MODEL.ROI_BOX_HEAD.NUM_CLASSES == 81
    model = build_detection_model(c)
    assert not model.da_heads
    model.load_state_dict(fill_state_dict(model.state_dict(), int(z["seed"])))
    model = model.to(device).eval()
    H, W = int(z["H"]), int(z["W"])
    images, _ = make_batch(c, int(z["nimg"]), H, W, seed=int(z["seed"]), device=device)
    assert tuple(images.tensors.shape[-2:]) == (H, W), "SIZE_DIVISIBILITY 0: no padding"
    captured = {}
    model.rpn.head.register_forward_hook(
        lambda m, i, o: captured.update(objectness=o[0][0].detach(), deltas=o[1][0].detach()))
    model.roi_heads.box.predictor.register_forward_hook(
        lambda m, i, o: captured.update(class_logits=o[0].detach(), box_regression=o[1].detach()))
    selector = model.rpn.box_selector_test
    orig = selector.forward
    gold_obj = [torch.from_numpy(z["objectness"]).to(device)]
    gold_del = [torch.from_numpy(z["deltas"]).to(device)]
    selector.forward = lambda anchors, objectness, box_regression, tg=None: orig(anchors, gold_obj, gold_del, tg)
    try:
        with torch.no_grad():
            dets = model(images)
    finally:
        selector.forward = orig
    assert tuple(captured["objectness"].shape) == z["objectness"].shape
    np.testing.assert_allclose(captured["objectness"].cpu().numpy(), z["objectness"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(captured["deltas"].cpu().numpy(), z["deltas"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(captured["class_logits"].cpu().numpy(), z["class_logits"], rtol=1e-4, atol=5e-5)
    rows = int(z["box_regression_rows"])
    np.testing.assert_allclose(captured["box_regression"].cpu().numpy()[::rows], z["box_regression"], rtol=1e-4,
                               atol=5e-5)
    assert len(dets) == int(z["nimg"])
    for i, d in enumerate(dets):
        assert d.size == (W, H)
        assert np.array_equal(d.get_field("labels").cpu().numpy(), z["det/%d/labels" % i])
        np.testing.assert_allclose(d.bbox.cpu().numpy(), z["det/%d/boxes" % i], atol=5e-3)
        np.testing.assert_allclose(d.get_field("scores").cpu().numpy(), z["det/%d/scores" % i], atol=2e-5)


def test_config1_plain_c4_training_step_at_800x1333(device):
    """the same configuration in training mode at its own size.  The reference cannot run it (UnboundLocalError at
    generalized_rcnn.py:150) so there is no golden: losses finite, every trainable parameter gets a
    finite gradient, and the step is reproducible."""
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.engine.trainer import train_step
    from da_detect_amd.modeling.detector import build_detection_model
    from da_detect_amd.parallel.reducer import BucketedGradReducer
    from da_detect_amd.solver import make_optimizer
    from golden.cases import case_cfg
    from golden.fill import fill_state_dict

    c = case_cfg("c4_plain")
    model = build_detection_model(c)
    model.load_state_dict(fill_state_dict(model.state_dict(), 1))
    model = model.to(device).train()
    images, targets = make_batch(c, 2, 800, 1333, seed=1, device=device)
    for t in targets:     # plain training: every image is labelled
        t.add_field("is_source", torch.ones_like(t.get_field("is_source")))
        t.add_field("labels", t.get_field("labels") % 80 + 1)
    opt = make_optimizer(c, model)
    opt.attach_reducer(BucketedGradReducer([p for p in model.parameters() if p.requires_grad]))
    torch.manual_seed(1)
    losses = train_step(model, opt, images, targets)
    assert set(losses) == {"loss_classifier", "loss_box_reg", "loss_objectness", "loss_rpn_box_reg"}
    assert all(bool(torch.isfinite(v)) for v in losses.values()), losses
    for n, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().sum()) > 0, n


def test_eval_without_injection_and_rpn_only(device):
    """no injection: same number of detections per image and score multiset within fp32 noise; RPN_ONLY eval
    returns the proposals (generalized_rcnn.py:141-143)"""
    z, model, images = _eval_model(device)
    with torch.no_grad():
        dets = model(images)
    for i, d in enumerate(dets):
        want = np.sort(z["det/%d/scores" % i])
        got = np.sort(d.get_field("scores").cpu().numpy())
        assert abs(len(got) - len(want)) <= 2
        k = min(len(got), len(want))
        np.testing.assert_allclose(got[-k:], want[-k:], atol=1e-4)
    roi_heads, model.roi_heads = model.roi_heads, None
    try:
        with torch.no_grad():
            props = model(images)
    finally:
        model.roi_heads = roi_heads
    for i, p in enumerate(props):
        assert abs(len(p) - len(z["proposals/%d/boxes" % i])) <= 3 and p.has_field("objectness")


def test_fpn_eval_detections_match_reference_golden(device):
    """north-star item a20: FPN.forward, 5-level RPN + select_over_all_levels, LevelMapper pooling, FPN2MLP head and
    FPNPredictor in eval mode vs the imported reference (tests/golden/eval_fpn.npz); proposal selection is fed the
    fixture's per-level RPN maps."""
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.modeling.detector import build_detection_model
    from golden.cases import case_cfg
    from golden.fill import fill_state_dict

    z = np.load(os.path.join(GOLD, "eval_fpn.npz"))
    c = case_cfg("fpn")
    model = build_detection_model(c)
    model.load_state_dict(fill_state_dict(model.state_dict(), int(z["seed"])))
    model = model.to(device).eval()
    images, _ = make_batch(c, int(z["nimg"]), int(z["H"]), int(z["W"]), seed=int(z["seed"]), device=device)
    captured = {}
    model.rpn.head.register_forward_hook(lambda m, i, o: captured.update(objectness=[t.detach() for t in o[0]]))
    model.roi_heads.box.predictor.register_forward_hook(
        lambda m, i, o: captured.update(class_logits=o[0].detach(), box_regression=o[1].detach()))
    selector = model.rpn.box_selector_test
    orig = selector.forward
    gold_obj = [torch.from_numpy(z["objectness/%d" % l]).to(device) for l in range(5)]
    gold_del = [torch.from_numpy(z["deltas/%d" % l]).to(device) for l in range(5)]
    selector.forward = lambda anchors, objectness, box_regression, tg=None: orig(anchors, gold_obj, gold_del, tg)
    try:
        with torch.no_grad():
            dets = model(images)
    finally:
        selector.forward = orig
    for l in range(5):
        np.testing.assert_allclose(captured["objectness"][l].cpu().numpy(), z["objectness/%d" % l], rtol=1e-4,
                                   atol=1e-4)
    np.testing.assert_allclose(captured["class_logits"].cpu().numpy(), z["class_logits"], rtol=1e-4, atol=5e-5)
    rows = int(z["box_regression_rows"])
    np.testing.assert_allclose(captured["box_regression"].cpu().numpy()[::rows], z["box_regression"], rtol=1e-4,
                               atol=5e-5)
    for i, d in enumerate(dets):
        assert np.array_equal(d.get_field("labels").cpu().numpy(), z["det/%d/labels" % i])
        np.testing.assert_allclose(d.bbox.cpu().numpy(), z["det/%d/boxes" % i], atol=5e-3)
        np.testing.assert_allclose(d.get_field("scores").cpu().numpy(), z["det/%d/scores" % i], atol=2e-5)


def test_fpn_training_step_runs_and_grads_flow(device):
    """plain (non-DA) FPN training: the reference itself cannot run it (generalized_rcnn.py:150 UnboundLocalError,
    SURVEY.md fact 5), so there is no golden; check the losses are finite and every trainable parameter of the
    pyramid / MLP head gets a finite gradient through the hand-written backward."""
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.modeling.detector import build_detection_model
    from golden.cases import case_cfg
    from golden.fill import fill_state_dict

    c = case_cfg("fpn")
    c.merge_from_list(["MODEL.DA_HEADS.DA_IMG_LOSS_WEIGHT", 0.0, "MODEL.DOMAIN_ADAPTATION_ON", False])
    model = build_detection_model(c)
    model.load_state_dict(fill_state_dict(model.state_dict(), 1))
    model = model.to(device).train()
    images, targets = make_batch(c, 2, 192, 320, seed=1, device=device)
    for t in targets:
        t.add_field("is_source", torch.ones_like(t.get_field("is_source")))
    losses = model(images, targets)
    assert set(losses) >= {"loss_classifier", "loss_box_reg", "loss_objectness", "loss_rpn_box_reg"}
    total = sum(losses.values())
    assert torch.isfinite(total)
    total.backward()
    for name, p in model.named_parameters():
        if p.requires_grad and ("fpn" in name or "fc6" in name or "fc7" in name or "rpn.head" in name
                                or "predictor" in name or "layer4" in name):
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
            # at 192x320 no ROI is large enough for P4/P5 (LevelMapper) and few anchors are sampled there: the
            # output convs of the two coarsest levels may legitimately see a zero gradient
            if "fpn_layer3" not in name and "fpn_layer4" not in name:
                assert p.grad.abs().sum() > 0, name


@pytest.mark.parametrize("case", ["da_plain", "da_triplet_aligned"])
def test_overlapped_rpn_backward_gives_the_same_gradients(device, case):
    """RPNModule.early_backward queues the RPN branch's backward during the forward pass and bridges its feature
    gradient back into the main graph: every parameter gradient must equal the single-pass one (same sums, the
    feature-gradient additions merely associate differently)."""
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward
    from da_detect_amd.utils import rng

    z, c, model, sd = _build(case, device)
    images, targets = make_batch(c, int(z["nimg"]), int(z["H"]), int(z["W"]), seed=int(z["seed"]), device=device)
    grads = []
    for early in (False, True):
        enable_overlapped_rpn_backward(model, early)
        model.zero_grad(set_to_none=True)
        rng.use_cpu_stream(True)
        try:
            torch.manual_seed(3)
            losses = model(images, targets)
            if early:
                assert not losses["loss_objectness"].requires_grad and losses["loss_classifier"].requires_grad
            sum(losses.values()).backward()
        finally:
            rng.use_cpu_stream(False)
        grads.append({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    enable_overlapped_rpn_backward(model, False)
    assert set(grads[0]) == set(grads[1]) and any(n.startswith("rpn.head") for n in grads[0])
    for n in grads[0]:
        a, b = grads[0][n], grads[1][n]
        assert float((a - b).norm()) <= 1e-5 * float(a.norm()) + 1e-10, n


def test_fpn_dcn_da_training_step(device):
    """BASELINE configs[4] (R-101-FPN + DCN + DA heads).  The reference cannot run this combination (da_heads loss
    concatenates per-level logits along dim 0; instance head sized for C4), so there is no golden: check (1) the step
    runs through DFConv2d bottlenecks, FPN, per-level image heads, the FPN2MLP instance features, with finite losses
    and gradients everywhere; (2) the multi-level image loss equals torch's BCE over the elements of ALL levels and
    the consistency term equals the reference's formula applied per level (consistency_loss.py:13-27)."""
    import torch.nn.functional as F

    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.modeling.detector import build_detection_model
    from golden.cases import fpn_dcn_da_cfg
    from golden.fill import fill_state_dict

    c = fpn_dcn_da_cfg()
    model = build_detection_model(c)
    sd = fill_state_dict(model.state_dict(), 2)
    for k in sd:
        if ".conv2.offset." in k:
            sd[k] = sd[k] * 0.05      # small but non-zero sampling offsets
    model.load_state_dict(sd)
    model = model.to(device).train()
    images, targets = make_batch(c, 2, 192, 320, seed=2, device=device)
    captured = {}
    model.backbone.register_forward_hook(lambda m, i, o: captured.__setitem__("feats", [t.detach() for t in o]))
    losses = model(images, targets)
    assert set(losses) == {"loss_classifier", "loss_box_reg", "loss_objectness", "loss_rpn_box_reg", "loss_da_image",
                           "loss_da_instance", "loss_da_consistency"}
    total = sum(losses.values())
    assert torch.isfinite(total)
    total.backward()
    seen = set()
    for name, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
            if p.grad.abs().sum() > 0:
                seen.add(name.split(".")[0] + "." + name.split(".")[1])
    assert {"backbone.body", "backbone.fpn", "rpn.head", "roi_heads.box", "da_heads.imghead",
            "da_heads.inshead"} <= seen, seen
    assert any(p.grad.abs().sum() > 0 for n, p in model.named_parameters() if ".conv2.offset.weight" in n)
    # (2) image-level loss over the pyramid
    head = model.da_heads.imghead
    with torch.no_grad():
        logits = head(captured["feats"])
        labels = torch.tensor([1.0, 0.0], device=device)
        flat = torch.cat([l.reshape(2, -1) for l in logits], dim=1)
        want = F.binary_cross_entropy_with_logits(flat, labels[:, None].expand_as(flat))
    got = losses["loss_da_image"].detach() / c.MODEL.DA_HEADS.DA_IMG_LOSS_WEIGHT
    assert abs(float(got) - float(want)) <= 1e-4 * abs(float(want)), (float(got), float(want))


def test_fused_sampler_path_trains_and_agrees_with_the_index_loss(device):
    """default GPU path (device-side random keys, dadet_sample_rois + per-row loss kernel): the sample obeys the
    reference's counts (box_head/loss.py:95-130), the step is reproducible under torch.manual_seed, and on the SAME
    sample the per-row loss equals the index-list loss of the parity path"""
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.layers.misc import fast_rcnn_loss_fused
    from da_detect_amd.utils import rng

    z, c, model, _ = _build("da_plain", device)
    seed, H, W, nimg = int(z["seed"]), int(z["H"]), int(z["W"]), int(z["nimg"])
    images, targets = make_batch(c, nimg, H, W, seed=seed, device=device)
    assert not rng.cpu_stream_enabled()
    evaluator = model.roi_heads.box.loss_evaluator
    captured = {}
    orig_call = evaluator.__class__.__call__

    def spy(self, class_logits, box_regression):
        captured["logits"], captured["reg"] = class_logits[0].detach(), box_regression[0].detach()
        return orig_call(self, class_logits, box_regression)

    evaluator.__class__.__call__ = spy
    try:
        torch.manual_seed(seed)
        losses = model(images, targets)
        sum(losses.values()).backward()
        first = {k: float(v.detach()) for k, v in losses.items()}
        prep = evaluator._loss_prep
        assert prep.get("rows") is True
        cap = evaluator.fg_bg_sampler.batch_size_per_image
        for p, (n_pos, n_neg), src in zip(evaluator._proposals, evaluator.fg_bg_sampler.last_counts,
                                          evaluator._is_source):
            lab = p.get_field("labels")
            assert len(p) == n_pos + n_neg <= cap and int((lab > 0).sum()) == n_pos
            assert n_pos <= int(cap * evaluator.fg_bg_sampler.positive_fraction)
            assert bool((p.get_field("domain_labels") == src).all())
            assert src or n_pos == 0
        # the same sample through the index-list kernel
        ll, rt = prep["loss_labels"], prep["regression_targets"]
        srcr = torch.nonzero(ll >= 0).squeeze(1)
        labels_src = ll[srcr]
        pos = torch.nonzero(labels_src > 0).squeeze(1)
        map_inds = 4 * labels_src[pos][:, None] + torch.arange(4, device=ll.device)
        k0, k1 = fast_rcnn_loss_fused(captured["logits"], captured["reg"], srcr, labels_src, srcr[pos], map_inds,
                                      rt[srcr[pos]])
        assert abs(float(k0) - first["loss_classifier"]) <= 1e-6 * max(1.0, abs(float(k0)))
        assert abs(float(k1) - first["loss_box_reg"]) <= 1e-6 * max(1.0, abs(float(k1)))
        assert all(np.isfinite(v) for v in first.values())
        grads = [p.grad for p in model.roi_heads.box.parameters() if p.grad is not None]
        assert grads and all(bool(torch.isfinite(g).all()) for g in grads)
        model.zero_grad()
        torch.manual_seed(seed)
        again = {k: float(v.detach()) for k, v in model(images, targets).items()}
        for k in first:     # same seed -> same sample -> the box-head losses repeat bit for bit; the image-level DA
            tol = 0.0 if k in ("loss_classifier", "loss_box_reg") else 1e-5   # loss sums with atomics
            assert abs(again[k] - first[k]) <= tol * max(1.0, abs(first[k])), (k, first[k], again[k])
    finally:
        evaluator.__class__.__call__ = orig_call


def test_direct_weight_gradient_accumulation_matches_the_autograd_path(device):
    """weight gradients added straight into the reducer's flat buckets on the lane (utils.streams.enable_direct_wgrad)
    vs the same step with autograd accumulating them: identical parameters after one optimizer step"""
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.engine.trainer import train_step
    from da_detect_amd.parallel.reducer import BucketedGradReducer
    from da_detect_amd.solver import make_optimizer
    from da_detect_amd.utils import rng, streams

    results = []
    for direct in (False, True):
        z, c, model, _ = _build("da_plain", device)
        seed, H, W, nimg = int(z["seed"]), int(z["H"]), int(z["W"]), int(z["nimg"])
        images, targets = make_batch(c, nimg, H, W, seed=seed, device=device)
        opt = make_optimizer(c, model)
        reducer = BucketedGradReducer([p for p in model.parameters() if p.requires_grad])
        opt.attach_reducer(reducer)
        streams.enable_direct_wgrad(direct)
        rng.use_cpu_stream(True)
        try:
            torch.manual_seed(seed)
            train_step(model, opt, images, targets)
            grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.requires_grad}
            torch.manual_seed(seed + 1)
            train_step(model, opt, images, targets)
        finally:
            rng.use_cpu_stream(False)
            streams.enable_direct_wgrad(False)
        torch.cuda.synchronize()
        results.append((grads, {n: p.detach().clone() for n, p in model.named_parameters()}, set(reducer.touched)))
    (g0, p0, t0), (g1, p1, t1) = results
    assert len(t0) == len(t1) > 50
    # (the direct path issues a block's weight gradients as ONE grouped launch — another cut of the reduction over the rows
    # than the per-layer launches of the autograd path: the same sums in another order, so elements far below a tensor's
    # largest agree to fp32 resolution of THAT, not of themselves)
    for n in g0:
        top = float(g0[n].abs().max())
        torch.testing.assert_close(g1[n], g0[n], rtol=1e-5, atol=max(1e-7, 2e-6 * top), msg=lambda m, n=n: "%s: %s" % (n, m))
    for n in p0:
        torch.testing.assert_close(p1[n], p0[n], rtol=1e-5, atol=1e-7, msg=lambda m, n=n: "%s: %s" % (n, m))


def test_a_block_used_twice_keeps_both_weight_gradient_contributions(device):
    """one Bottleneck applied twice in a graph with direct weight-gradient accumulation: both uses add split partial sums
    into the SAME gradient buffers, and the merged reduction launch must not see the two items at once (it is a plain
    read-modify-write per buffer).  Compared with autograd accumulating the two contributions."""
    from da_detect_amd.modeling.backbone.resnet import BottleneckWithFixedBatchNorm
    from da_detect_amd.parallel.reducer import BucketedGradReducer
    from da_detect_amd.utils import streams

    torch.manual_seed(3)
    block = BottleneckWithFixedBatchNorm(256, 64, 256, stride=1).to(device)
    x = torch.randn((2, 256, 64, 96), device=device).contiguous(memory_format=torch.channels_last)
    params = [p for p in block.parameters() if p.requires_grad]
    grads = []
    for direct in (False, True):
        red = BucketedGradReducer(params)
        streams.enable_direct_wgrad(direct)
        try:
            red.zero_grad()
            y = block(block(x, in_relu=False, out_private=False), in_relu=True, out_private=False)
            y.square().mean().backward()
            if direct:
                assert len(streams._PENDING_REDUCES) > 0, "the case needs split weight gradients queued for the merged pass"
            red.finalize()
            torch.cuda.synchronize()
            grads.append([p.grad.clone() for p in params])
        finally:
            streams.enable_direct_wgrad(False)
            for h in red._hooks:
                h.remove()
    for a, b in zip(*grads):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=1e-7)


def test_early_image_level_da_backward_gives_the_same_gradients(device):
    """without a consistency term the image-level DA loss and its backward are queued in front of the box head
    (DomainAdaptationModule.early_image_level) and the instance-head passes run on a side stream: losses and every
    parameter gradient equal the reference-order schedule's"""
    from golden.cases import case_cfg
    from golden.fill import fill_state_dict

    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.engine.trainer import enable_overlapped_rpn_backward
    from da_detect_amd.modeling.detector import build_detection_model
    from da_detect_amd.utils import rng

    z = np.load(os.path.join(GOLD, "da_plain.npz"))
    c = case_cfg("da_plain")
    c.merge_from_list(["MODEL.DA_HEADS.DA_CST_LOSS_WEIGHT", 0.0])
    model = build_detection_model(c)
    model.load_state_dict(fill_state_dict(model.state_dict(), int(z["seed"])))
    model = model.to(device).train()
    images, targets = make_batch(c, int(z["nimg"]), int(z["H"]), int(z["W"]), seed=int(z["seed"]), device=device)
    runs = []
    for early in (False, True):
        enable_overlapped_rpn_backward(model, early)
        model.zero_grad(set_to_none=True)
        rng.use_cpu_stream(True)
        try:
            torch.manual_seed(3)
            losses = model(images, targets)
            assert "loss_da_image" in losses and "loss_da_instance" in losses and "loss_da_consistency" not in losses
            assert losses["loss_da_image"].requires_grad != early
            sum(losses.values()).backward()
        finally:
            rng.use_cpu_stream(False)
        torch.cuda.synchronize()
        runs.append(({k: float(v.detach()) for k, v in losses.items()},
                     {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    enable_overlapped_rpn_backward(model, False)
    (l0, g0), (l1, g1) = runs
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 1e-5 * max(1.0, abs(l0[k])), (k, l0[k], l1[k])
    assert set(g0) == set(g1) and any(n.startswith("da_heads.imghead") for n in g0)
    for n in g0:
        assert float((g0[n] - g1[n]).norm()) <= 1e-5 * float(g0[n].norm()) + 1e-10, n


def test_fpn_device_side_selection_equals_the_host_chain(device, monkeypatch):
    """multi-level training selection with every count left on the device (rpn/inference.py
    `_select_over_all_levels_device`: fixed-capacity buffers, batch-wide top-k as a mask, PendingProposals) against the
    host chain of the reference (per-level count round trips, `select_over_all_levels`, `add_gt_proposals`): the same
    boxes with the same scores in the same order, per image"""
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.modeling.detector import build_detection_model
    from da_detect_amd.modeling.rpn import inference as inf
    from da_detect_amd.structures.image_list import to_image_list
    from golden.cases import fpn_dcn_da_cfg
    from golden.fill import fill_state_dict

    c = fpn_dcn_da_cfg()
    model = build_detection_model(c)
    model.load_state_dict(fill_state_dict(model.state_dict(), 5))
    model = model.to(device).train()
    images, targets = make_batch(c, 2, 192, 320, seed=5, device=device)
    with torch.no_grad():
        il = to_image_list(images)
        feats = model.backbone(il.tensors)
        obj, reg = model.rpn.head(feats)
        anchors = model.rpn.anchor_generator(il, feats)
        sel = model.rpn.box_selector_train
        monkeypatch.setattr(inf, "_DEVICE_SELECT", True)
        sel.defer = True
        dev_out = sel(anchors, obj, reg, targets)
        assert all(getattr(type(p), "is_pending_proposals", False) for p in dev_out)
        monkeypatch.setattr(inf, "_DEVICE_SELECT", False)
        sel.defer = False
        host_out = sel(anchors, obj, reg, targets)
    assert len(dev_out) == len(host_out) == 2
    for d, h in zip(dev_out, host_out):
        assert not getattr(type(h), "is_pending_proposals", False)
        assert len(d) == len(h) and len(h) > 50
        assert torch.equal(d.bbox, h.bbox)
        assert torch.equal(d.get_field("objectness"), h.get_field("objectness"))


@pytest.mark.parametrize("post_nms", [None, 40])
def test_box_head_queued_before_the_sampled_counts_gives_the_same_step(device, post_nms, monkeypatch):
    """ROIBoxHead.forward queues the pooler + res5 head on the assumption that every image fills its BATCH_SIZE_PER_IMAGE
    rows, before the sampler's counts have reached the host, and keeps the result only when the counts confirm it.
    Same losses and gradients, bit for bit, as with the counts first (DADET_ROI_SPECULATE=0's order) — also when the
    assumption FAILS (post_nms = 40: fewer proposals than rows to fill, the queued result is dropped)."""
    from da_detect_amd.data.synthetic import make_batch
    from da_detect_amd.modeling.roi_heads.box_head import box_head

    z, c, model, _ = _build("da_plain", device)
    seed, H, W, nimg = int(z["seed"]), int(z["H"]), int(z["W"]), int(z["nimg"])
    images, targets = make_batch(c, nimg, H, W, seed=seed, device=device)
    if post_nms is not None:
        model.rpn.box_selector_train.post_nms_top_n = post_nms
        model.rpn.box_selector_train.fpn_post_nms_top_n = post_nms
    evaluator = model.roi_heads.box.loss_evaluator
    seen = {}
    orig_finish = evaluator.subsample_finish

    def finish(state):
        out = orig_finish(state)
        seen["exact"], seen["queued"] = state["exact"], state["speculative"] is not None
        return out

    evaluator.subsample_finish = finish
    extractor = model.roi_heads.box.feature_extractor
    calls = []
    extractor.register_forward_hook(lambda m, i, o: calls.append(1))
    results = {}
    for spec in (True, False):
        monkeypatch.setattr(box_head, "_SPECULATE", spec)
        seen.clear()
        del calls[:]
        model.zero_grad()
        torch.manual_seed(seed)
        losses = model(images, targets)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        results[spec] = ({k: float(v.detach()) for k, v in losses.items()},
                         {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
        if spec:
            assert seen["queued"] and seen["exact"] == (post_nms is None), seen
            cap = evaluator.fg_bg_sampler.batch_size_per_image
            assert all((len(p) == cap) == (post_nms is None) for p in evaluator._proposals)
            assert len(calls) == (1 if post_nms is None else 2)      # a dropped result is pooled again from the exact lists
            # ... and counted: a configuration that keeps dropping switches the early queue off by itself
            st = model.roi_heads.box.speculation
            assert (st["kept"], st["dropped"]) == ((1, 0) if post_nms is None else (0, 1)), st
        else:
            assert len(calls) == 1
    if post_nms is not None:
        head = model.roi_heads.box
        monkeypatch.setattr(box_head, "_SPECULATE", True)
        for _ in range(15):
            assert head.speculate
            head._note_speculation(False)
        assert not head.speculate and head.speculation["dropped"] == 16
        del calls[:]
        torch.manual_seed(seed)
        model(images, targets)
        assert len(calls) == 1 and head.speculation["dropped"] == 16      # nothing queued early any more
    (la, ga), (lb, gb) = results[True], results[False]
    for k in la:      # (the DA losses sum with atomics: not bit-reproducible from one call to the next)
        tol = 1e-5 if k.startswith("loss_da") else 0.0
        assert abs(la[k] - lb[k]) <= tol * max(1.0, abs(lb[k])), (k, la[k], lb[k])
    box = [n for n in ga if n.startswith("roi_heads.")]
    assert box and set(ga) == set(gb)
    for n in box:     # (the res5 head also carries the instance-level DA branch's gradient: same sums, atomics' order)
        if "predictor" in n:
            assert torch.equal(ga[n], gb[n]), n
        else:
            torch.testing.assert_close(ga[n], gb[n], rtol=1e-4, atol=1e-6 * float(gb[n].abs().max()), msg=n)
