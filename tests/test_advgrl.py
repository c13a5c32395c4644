"""The ACTIVE AdvGRL branch (reference da_heads.py:173-195; called at :128-139 for the image head and :153-160 for the
instance head) against tests/golden/advgrl.npz, generated from the imported reference by
tests/golden/make_golden_advgrl.py.  The reversal weight only acts in backward, so the comparison is on gradients:
d loss_da_image / d C4 features and d loss_da_instance / d ROI features, plus the weights themselves.

Cases: `active` (weight = -advGRL_WEIGHT / loss, both heads), `clamped` (min(threshold, 1/loss) clamps, both heads),
`instance_dormant` (image head active, instance head above the 0.6288 gate -> fixed -GRL_WEIGHT)."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
Z = os.path.join(HERE, "golden", "advgrl.npz")
CASES = ["active", "clamped", "instance_dormant"]


@pytest.mark.parametrize("case", CASES)
def test_adv_grl_weight_rule_matches_reference(case):
    """DomainAdaptationModule_triplet.adv_grl_weight (host logic, runs on the CPU) on the reference's current losses"""
    from da_detect_amd.modeling.da_heads.da_heads import DomainAdaptationModule_triplet
    from golden.cases import ADVGRL_CASES, case_cfg

    z = np.load(Z)
    c = case_cfg("da_triplet")
    c.merge_from_list(list(ADVGRL_CASES[case][0]))
    mod = DomainAdaptationModule_triplet(c)
    assert abs(mod.adv_gate - 0.628771) < 1e-5
    da = c.MODEL.DA_HEADS
    for branch, base, adv in (("img", da.DA_IMG_GRL_WEIGHT, da.DA_IMG_advGRL_WEIGHT),
                              ("ins", da.DA_INS_GRL_WEIGHT, da.DA_INS_advGRL_WEIGHT)):
        cur = torch.tensor(float(z["%s/current_%s" % (case, branch)]))
        w = float(mod.adv_grl_weight(cur, base, adv))
        want = float(z["%s/weight_%s" % (case, branch)])
        assert abs(w - want) <= 1e-6 * max(1.0, abs(want)), (case, branch, w, want)
        assert bool(z["%s/active_%s" % (case, branch)]) == (float(cur) <= mod.adv_gate)
    if case == "clamped":
        assert float(z["clamped/weight_img"]) == pytest.approx(-da.DA_IMG_advGRL_WEIGHT * da.DA_ADV_GRL_THRESHOLD)
    if case == "instance_dormant":
        assert float(z["instance_dormant/weight_ins"]) == pytest.approx(-da.DA_INS_GRL_WEIGHT)


def test_oracle_reproduces_the_reference_gradients_through_the_active_branch():
    """oracle/model_ref.py (adv_weight + _GRL) on the `clamped` case: losses and both gradients of the fixture"""
    from golden.cases import advgrl_setup
    from oracle import model_ref

    z = np.load(Z)
    case = "clamped"
    c, model, sd, images, targets = advgrl_setup(z, case, torch.device("cpu"))
    for n, p in model.named_parameters():
        if p.requires_grad:
            sd[n].requires_grad_(True)
    inter = {}
    torch.manual_seed(int(z["seed"]))
    losses = model_ref.training_losses(sd, c, images.tensors, model_ref.targets_to_dicts(targets), state={},
                                       intermediates=inter, grad_probe=True)
    for k in losses:
        want = float(z["%s/loss/%s" % (case, k)])
        assert abs(float(losses[k].detach()) - want) <= 1e-5 * max(abs(want), 1.0), (k, float(losses[k].detach()), want)
    g_feat, = torch.autograd.grad(losses["loss_da_image"], inter["feat_graph"], retain_graph=True)
    g_ins, = torch.autograd.grad(losses["loss_da_instance"], inter["ins_feat_graph"])
    np.testing.assert_allclose(g_feat[:2, ::16].numpy(), z[case + "/g_feat"], rtol=1e-4,
                               atol=1e-5 * float(z[case + "/g_feat_absmax"]))
    np.testing.assert_allclose(g_ins[:, ::64, 0, 0].numpy(), z[case + "/g_ins"], rtol=1e-4,
                               atol=1e-5 * float(z[case + "/g_ins_absmax"]))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_path_through_the_active_advgrl_branch(device, case):
    """the fused image head (weight resolved on the device from its own forward loss, da_heads.hip) and the instance
    head's gradient_scalar with a tensor weight, against the reference's gradients"""
    from da_detect_amd.utils import rng
    from golden.cases import advgrl_setup

    z = np.load(Z)
    c, model, sd, images, targets = advgrl_setup(z, case, device)
    model.load_state_dict(sd)
    model = model.to(device).train()
    got = {}
    model.backbone.register_forward_hook(lambda m, i, o: got.__setitem__("feat", o[0]))
    model.da_heads_triplet.register_forward_pre_hook(lambda m, args: got.__setitem__("ins_feat", args[1]))
    selector = model.rpn.box_selector_train
    orig_sel = selector.forward
    gold_obj = [torch.from_numpy(z["objectness"]).to(device)]
    gold_del = [torch.from_numpy(z["deltas"]).to(device)]
    selector.forward = lambda anchors, objectness, box_regression, tg=None: orig_sel(anchors, gold_obj, gold_del, tg)
    rng.use_cpu_stream(True)        # the reference's randperm / dropout stream
    try:
        torch.manual_seed(int(z["seed"]))
        losses = model(images, targets)
    finally:
        rng.use_cpu_stream(False)
        selector.forward = orig_sel
    want = {k[len(case) + 6:]: float(z[k]) for k in z.files if k.startswith(case + "/loss/")}
    assert set(losses) == set(want), (sorted(losses), sorted(want))
    for k, v in want.items():
        assert abs(float(losses[k].detach()) - v) <= 1e-4 * max(abs(v), 1.0), (case, k, float(losses[k].detach()), v)
    assert float(losses["loss_da_image"].detach()) < 0.6288, "the image branch must be on the active side of the gate"
    g_feat, = torch.autograd.grad(losses["loss_da_image"], got["feat"], retain_graph=True)
    g_ins, = torch.autograd.grad(losses["loss_da_instance"], got["ins_feat"], retain_graph=True)
    assert float(g_feat[2].abs().max()) == 0.0
    np.testing.assert_allclose(g_feat[:2, ::16].cpu().numpy(), z[case + "/g_feat"], rtol=1e-3,
                               atol=1e-4 * float(z[case + "/g_feat_absmax"]))
    np.testing.assert_allclose(g_ins[:, ::64, 0, 0].cpu().numpy(), z[case + "/g_ins"], rtol=1e-3,
                               atol=1e-4 * float(z[case + "/g_ins_absmax"]))
    # the scale of the gradient IS the reversal weight: a dormant (-0.1) weight on the image branch would be off by
    # the factor weight / -0.1 (4.6x in `active`, 2x in `clamped`)
    ratio = float(g_feat[:2, ::16].abs().sum()) / float(np.abs(z[case + "/g_feat"]).sum())
    assert abs(ratio - 1.0) < 1e-3, ratio
