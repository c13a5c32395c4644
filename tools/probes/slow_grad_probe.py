"""Per-tensor relative L2 error of the HIP path's parameter gradients against the float64 oracle on the da_plain golden case
(the harness of tests/test_model_gpu.py::test_gradients_match_cpu_oracle), once per contraction mode: separates what a mode's
arithmetic contributes from what a ReLU flipped by rounding contributes (the latter shows in every mode, on different units).
  python tools/probes/slow_grad_probe.py [modes, default 4,0]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(mode, device):
    from da_detect_amd import _C
    from da_detect_amd.data.synthetic import make_batch
    from oracle import model_ref
    import test_model_gpu as T

    _C.set_gemm_mode(mode)
    z, c, model, sd = T._build("da_plain", device)
    seed, H, W, nimg = int(z["seed"]), int(z["H"]), int(z["W"]), int(z["nimg"])
    images, targets = make_batch(c, nimg, H, W, seed=seed, device=device)
    losses, _ = T._run_with_golden_rpn_selection(model, z, images, targets, seed, device, inject=True)
    sum(losses.values()).backward()
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    got = {n: p.grad.detach().cpu().double() for n, p in model.named_parameters() if p.requires_grad}
    if run.want is None:
        osd = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        for n in names:
            osd[n].requires_grad_(True)
        cpu_images, cpu_targets = make_batch(c, nimg, H, W, seed=seed, device=torch.device("cpu"))
        torch.manual_seed(seed)
        ol = model_ref.training_losses(osd, c, cpu_images.tensors.double(), model_ref.targets_to_dicts(cpu_targets),
                                       selection_maps=(torch.from_numpy(z["objectness"]), torch.from_numpy(z["deltas"])))
        sum(ol.values()).backward()
        run.want = {n: osd[n].grad.detach() for n in names}
        print("image size %d x %d, %d images" % (H, W, nimg))
    return {n: float((got[n] - run.want[n]).norm()) / (float(run.want[n].norm()) + 1e-30) for n in names}


run.want = None

if __name__ == "__main__":
    modes = [int(m) for m in (sys.argv[1] if len(sys.argv) > 1 else "4,0").split(",")]
    dev = torch.device("cuda:0")
    res = {m: run(m, dev) for m in modes}
    print("%-55s " % "parameter" + " ".join("mode %d   " % m for m in modes))
    for n in res[modes[0]]:
        print("%-55s " % n + " ".join("%.2e" % res[m][n] for m in modes))
    for m in modes:
        v = sorted(res[m].values())
        print("mode %d: median %.2e, above 5e-5: %d of %d, max %.2e" % (m, v[len(v) // 2], sum(x >= 5e-5 for x in v), len(v), v[-1]))
