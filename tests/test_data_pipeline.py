"""Data side: transforms, samplers, datasets on the CPU; the device-side batch
preparation on the GPU.  Pins: Pillow itself for the resize arithmetic, index streams recorded from the reference's
sampler modules (tests/golden/reference_samplers.json), the host transform chain for the device path."""
import json
import os
import random

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402

RESIZE_CASES = [(64, 128, 38, 75), (100, 60, 100, 33), (37, 53, 80, 101), (128, 256, 75, 150), (17, 19, 17, 19),
                (50, 50, 7, 93)]


def _image(rng, H, W):
    return rng.integers(0, 256, (H, W, 3), dtype=np.uint8)


@pytest.mark.parametrize("H,W,oh,ow", RESIZE_CASES)
def test_resize_restatement_equals_pillow(H, W, oh, ow):
    """oracle/image_ref.py vs Pillow's own Image.resize(BILINEAR) — what torchvision's F.resize calls for PIL images"""
    from oracle.image_ref import pil_bilinear_resize

    img = _image(np.random.default_rng(H + ow), H, W)
    want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(pil_bilinear_resize(img, oh, ow), want)


def test_vectorised_tables_equal_the_scalar_restatement():
    from da_detect_amd.data.device_prep import resample_tables
    from oracle.image_ref import resample_coeffs

    for a, b in [(2048, 1200), (1024, 600), (53, 101), (60, 33), (50, 7), (50, 93), (1914, 1052), (9, 9)]:
        b1, c1 = resample_tables(a, b)
        b2, c2 = resample_coeffs(a, b)
        assert np.array_equal(b1, b2) and np.array_equal(c1, c2), (a, b)


def test_get_size_rules():
    """transforms.py:41-62: shorter side to min_size unless the longer side would exceed max_size"""
    from da_detect_amd.data.transforms import Resize

    r = Resize(600, 1200)
    assert r.get_size((2048, 1024)) == (600, 1200)          # Cityscapes: the DA yamls' 600 / 1200
    assert r.get_size((1024, 2048)) == (1200, 600)
    assert r.get_size((500, 375)) == (600, 800)
    assert Resize(800, 1333).get_size((1000, 300)) == (400, 1333)   # capped by max_size: int(round(1333 * .3)) = 400
    assert Resize(800, 1333).get_size((640, 480)) == (800, 1066)
    assert Resize((600,), 1200).get_size((1200, 600)) == (600, 1200)   # already there: returned as is


def test_host_transform_chain_equals_the_oracle():
    """build_transforms(cfg) (PIL resize / flip, ToTensor, Normalize to_bgr255) vs oracle/image_ref.preprocess, same
    random decisions; boxes follow the image"""
    from da_detect_amd.config import cfg
    from da_detect_amd.data.transforms import build_transforms
    from da_detect_amd.structures.bounding_box import BoxList
    from oracle.image_ref import preprocess

    c = cfg.clone()
    c.merge_from_list(["INPUT.MIN_SIZE_TRAIN", (40,), "INPUT.MAX_SIZE_TRAIN", 70])
    tf = build_transforms(c, True)
    rng = np.random.default_rng(3)
    for seed in range(4):
        img = _image(rng, 48, 96)
        boxes = BoxList(torch.tensor([[4.0, 6.0, 40.0, 30.0], [50.0, 10.0, 95.0, 47.0]]), (96, 48), mode="xyxy")
        random.seed(seed)
        out, tgt = tf(Image.fromarray(img), boxes)
        random.seed(seed)
        oh, ow = tf.transforms[0].get_size((96, 48))
        flip = tf.transforms[1].toss()
        want = preprocess(img, oh, ow, flip, c.INPUT.PIXEL_MEAN, c.INPUT.PIXEL_STD, c.INPUT.TO_BGR255)
        assert (oh, ow) == (35, 70) and tuple(out.shape) == (3, 35, 70)
        assert np.array_equal(out.numpy(), want)
        ref = boxes.resize((ow, oh))
        if flip:
            ref = ref.transpose(0)
        assert torch.equal(tgt.bbox, ref.bbox) and tgt.size == (ow, oh)


def test_samplers_reproduce_the_reference_streams():
    from da_detect_amd.data import samplers as S

    gold = json.load(open(os.path.join(GOLD, "reference_samplers.json")))
    for case in gold["distributed"]:
        for rank, want in enumerate(case["indices"]):
            s = S.DistributedSampler(list(range(case["n"])), num_replicas=case["world"], rank=rank, shuffle=True)
            s.set_epoch(case["epoch"])
            assert list(s) == want and len(s) == len(want)
    for case in gold["grouped"]:
        base = S.DistributedSampler(list(range(case["n"])), num_replicas=1, rank=0, shuffle=True)
        base.set_epoch(case["epoch"])
        b = S.GroupedBatchSampler(base, case["groups"], case["batch_size"], drop_uneven=case["drop_uneven"])
        assert len(b) == len(case["batches"]) and [list(x) for x in b] == case["batches"]
    for case in gold["iteration"]:
        base = S.DistributedSampler(list(range(case["n"])), num_replicas=1, rank=0, shuffle=True)
        bs = torch.utils.data.sampler.BatchSampler(base, case["batch_size"], drop_last=False)
        it = S.IterationBasedBatchSampler(bs, case["num_iterations"], case["start_iter"])
        assert [list(x) for x in it] == case["batches"]


def _write_coco(tmp, name, n_images, rng, sizes=None):
    root = os.path.join(tmp, name)
    os.makedirs(root)
    images, annos = [], []
    aid = 1
    for i in range(n_images):
        H, W = sizes[i] if sizes else (40, 80)
        Image.fromarray(_image(rng, H, W)).save(os.path.join(root, "im%d.png" % i))
        images.append({"id": 100 + i, "file_name": "im%d.png" % i, "height": H, "width": W})
        for k in range(0 if i == 1 else 2):                  # image 1 has no annotation
            annos.append({"id": aid, "image_id": 100 + i, "category_id": 24 + 2 * k, "iscrowd": 0,
                          "bbox": [5 + k, 6, 20, 15], "area": 300})
            aid += 1
    ann = os.path.join(tmp, name + ".json")
    json.dump({"images": images, "annotations": annos,
               "categories": [{"id": 24, "name": "person"}, {"id": 26, "name": "car"}]}, open(ann, "w"))
    return ann, root


def test_datasets_and_loaders(tmp_path):
    from da_detect_amd.config import cfg
    from da_detect_amd.data.build import make_da_data_loaders, make_triplet_data_loader
    from da_detect_amd.data.datasets import COCODataset, TripletDataset

    rng = np.random.default_rng(0)
    specs = {k: _write_coco(str(tmp_path), k, 4, rng) for k in ("source", "target", "auxiliary")}
    ds = COCODataset(*specs["source"], remove_images_without_annotations=True, is_source=True, decode_to_tensor=True)
    assert len(ds) == 3 and ds.id_to_img_map == {0: 100, 1: 102, 2: 103}          # image 101 has no boxes
    img, tgt, idx = ds[0]
    assert img.dtype == torch.uint8 and tuple(img.shape) == (40, 80, 3) and idx == 0
    assert tgt.mode == "xyxy" and tgt.bbox.tolist() == [[5.0, 6.0, 24.0, 20.0], [6.0, 6.0, 25.0, 20.0]]
    assert tgt.get_field("labels").tolist() == [1, 2] and tgt.get_field("is_source").tolist() == [True, True]
    assert ds.contiguous_category_id_to_json_id == {1: 24, 2: 26} and ds.get_img_info(1)["file_name"] == "im2.png"
    tds = TripletDataset([COCODataset(*specs[k], remove_images_without_annotations=True, is_source=(k == "source"),
                                      decode_to_tensor=True) for k in ("source", "target", "auxiliary")])
    s = tds[1]
    assert torch.equal(s[3].bbox, s[1].bbox) and s[3].get_field("is_source").tolist() == [False, False]
    assert s[5].get_field("is_source").tolist() == [False, False] and s[1].get_field("is_source").all()
    c = cfg.clone()
    c.merge_from_list(["SOLVER.IMS_PER_BATCH", 2, "SOLVER.MAX_ITER", 3, "DATALOADER.NUM_WORKERS", 0,
                       "DATALOADER.SIZE_DIVISIBILITY", 32, "INPUT.MIN_SIZE_TRAIN", (32,), "INPUT.MAX_SIZE_TRAIN", 64,
                       "MODEL.DOMAIN_ADAPTATION_ON", True])
    loaders = make_da_data_loaders(c, {k: specs[k] for k in ("source", "target")})
    batches = [list(l) for l in loaders]
    assert [len(b) for b in batches] == [3, 3]
    images, targets, ids = batches[0][0]
    assert tuple(images.tensors.shape) == (1, 3, 32, 64) and len(targets) == 1 and targets[0].get_field("is_source").all()
    assert not batches[1][0][1][0].get_field("is_source").any()
    trip = list(make_triplet_data_loader(c, specs))
    assert len(trip) == 3 and len(trip[0]) == 9 and tuple(trip[0][2].tensors.shape) == (1, 3, 32, 64)


# ----------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_device_batch_preparation_is_bit_exact():
    """DeviceBatchPreparer (csrc/image.hip) vs the host chain restated in oracle/image_ref.py (itself equal to Pillow):
    resized / mirrored / normalised pixels bit for bit, zero padding to SIZE_DIVISIBILITY, targets transformed alike"""
    from da_detect_amd.config import cfg
    from da_detect_amd.data.device_prep import DeviceBatchPreparer
    from da_detect_amd.structures.bounding_box import BoxList
    from oracle.image_ref import preprocess

    c = cfg.clone()
    c.merge_from_list(["DATALOADER.SIZE_DIVISIBILITY", 32])
    prep = DeviceBatchPreparer(c, is_train=True)
    rng = np.random.default_rng(7)
    imgs = [_image(rng, 64, 128), _image(rng, 90, 100), _image(rng, 50, 120)]
    decisions = [((38, 75), True), ((90, 100), False), ((77, 50), True)]    # down-scale + flip, identity, mixed
    targets = [BoxList(torch.tensor([[3.0, 4.0, 60.0, 40.0]]), (im.shape[1], im.shape[0]), mode="xyxy") for im in imgs]
    for t in targets:
        t.add_field("labels", torch.tensor([1]))
    batch, out_t = prep([torch.from_numpy(i).cuda() for i in imgs], [t.to("cuda") for t in targets], decisions)
    assert tuple(batch.tensors.shape) == (3, 3, 96, 128) and batch.image_sizes == [(38, 75), (90, 100), (77, 50)]
    got = batch.tensors.cpu().numpy()
    for i, (im, ((oh, ow), flip)) in enumerate(zip(imgs, decisions)):
        want = preprocess(im, oh, ow, flip, c.INPUT.PIXEL_MEAN, c.INPUT.PIXEL_STD, c.INPUT.TO_BGR255)
        assert np.array_equal(got[i, :, :oh, :ow], want), i
        assert not got[i, :, oh:, :].any() and not got[i, :, :, ow:].any()
        ref = targets[i].resize((ow, oh))
        if flip:
            ref = ref.transpose(0)
        assert torch.equal(out_t[i].bbox.cpu(), ref.bbox) and out_t[i].size == (ow, oh)


@pytest.mark.gpu
def test_device_batch_preparation_at_cityscapes_size_and_model_input():
    """1024x2048 -> 600x1200 (the DA yamls' training size) against Pillow directly; the result feeds the model"""
    from da_detect_amd.config import cfg
    from da_detect_amd.data.device_prep import DeviceBatchPreparer

    c = cfg.clone()
    c.merge_from_list(["INPUT.MIN_SIZE_TRAIN", (600,), "INPUT.MAX_SIZE_TRAIN", 1200, "DATALOADER.SIZE_DIVISIBILITY", 32])
    prep = DeviceBatchPreparer(c, is_train=False)
    img = _image(np.random.default_rng(1), 1024, 2048)
    batch, _ = prep([torch.from_numpy(img).cuda()], None, [((600, 1200), False)])
    assert tuple(batch.tensors.shape) == (1, 3, 608, 1216)
    pil = np.asarray(Image.fromarray(img).resize((1200, 600), Image.BILINEAR)).astype(np.float32)
    want = (pil / np.float32(255.0))[:, :, ::-1] * np.float32(255.0) - np.asarray(c.INPUT.PIXEL_MEAN, np.float32)
    got = batch.tensors[0, :, :600, :1200].permute(1, 2, 0).cpu().numpy()
    assert np.array_equal(got, want.astype(np.float32))


@pytest.mark.gpu
def test_training_entry_point_on_a_tiny_dataset(tmp_path):
    """tools/train_net_da.py end to end: json datasets -> host transforms -> loaders -> do_da_train -> checkpoints"""
    import subprocess
    import sys

    rng = np.random.default_rng(0)
    # ragged sizes: source and target images of a step are zero-padded to a common (divisible-by-32) size
    specs = {"source": _write_coco(str(tmp_path), "source", 4, rng, sizes=[(96, 192), (96, 160), (80, 192), (96, 192)]),
             "target": _write_coco(str(tmp_path), "target", 4, rng, sizes=[(90, 180), (96, 192), (96, 140), (64, 192)])}
    out = str(tmp_path / "out")
    os.makedirs(out)
    root = os.path.dirname(HERE)
    cmd = [sys.executable, os.path.join(root, "tools", "train_net_da.py"), "--config-file",
           os.path.join(root, "configs/da_faster_rcnn/e2e_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml"),
           "--source", ",".join(specs["source"]), "--target", ",".join(specs["target"]),
           "SOLVER.MAX_ITER", "4", "SOLVER.CHECKPOINT_PERIOD", "2", "DATALOADER.NUM_WORKERS", "0",
           "INPUT.MIN_SIZE_TRAIN", "(96,)", "INPUT.MAX_SIZE_TRAIN", "192", "MODEL.OUTPUT_DIR", out, "MODEL.WEIGHT", ""]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "iter: 0" in res.stderr and "loss: " in res.stderr and "loss_da_consistency: " in res.stderr
    assert os.path.exists(os.path.join(out, "model_final.pth")) and os.path.exists(os.path.join(out, "model_0000002.pth"))
    ck = torch.load(os.path.join(out, "model_final.pth"), map_location="cpu", weights_only=False)
    assert set(ck) >= {"model", "optimizer", "scheduler", "iteration"}


@pytest.mark.gpu
def test_evaluation_entry_point_on_a_tiny_dataset(tmp_path):
    """tools/train_net_da.py -> checkpoint -> tools/test_net_da.py: json dataset -> host transforms -> eval forward ->
    COCO-style detection records (reference flow: tools/test_net.py + engine/inference.py:76-129)"""
    import subprocess
    import sys

    rng = np.random.default_rng(1)
    sizes = [(96, 192), (80, 160), (96, 128)]          # ragged test images: batches of 2 are zero-padded to a common size
    specs = {k: _write_coco(str(tmp_path), k, 3, rng, sizes=sizes) for k in ("source", "target")}
    out = str(tmp_path / "out")
    os.makedirs(out)
    root = os.path.dirname(HERE)
    yaml = os.path.join(root, "configs/da_faster_rcnn/e2e_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml")
    small = ["DATALOADER.NUM_WORKERS", "0", "INPUT.MIN_SIZE_TRAIN", "(96,)", "INPUT.MAX_SIZE_TRAIN", "192",
             "INPUT.MIN_SIZE_TEST", "96", "INPUT.MAX_SIZE_TEST", "192", "MODEL.WEIGHT", "",
             "MODEL.ROI_BOX_HEAD.NUM_CLASSES", "3"]          # background + the dataset's two categories
    train = [sys.executable, os.path.join(root, "tools", "train_net_da.py"), "--config-file", yaml,
             "--source", ",".join(specs["source"]), "--target", ",".join(specs["target"]),
             "SOLVER.MAX_ITER", "2", "SOLVER.CHECKPOINT_PERIOD", "0", "MODEL.OUTPUT_DIR", out] + small
    res = subprocess.run(train, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    ckpt = os.path.join(out, "model_final.pth")
    test = [sys.executable, os.path.join(root, "tools", "test_net_da.py"), "--config-file", yaml,
            "--dataset", ",".join(specs["target"]), "--ckpt", ckpt, "--output-dir", out,
            "TEST.IMS_PER_BATCH", "2", "MODEL.ROI_HEADS.SCORE_THRESH", "0.0"] + small
    res = subprocess.run(test, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    folder = os.path.join(out, "inference", "target")
    records = json.load(open(os.path.join(folder, "bbox.json")))
    assert os.path.exists(os.path.join(folder, "predictions.pth")) and len(records) > 0
    ids = {r["image_id"] for r in records}
    assert ids <= {100, 101, 102} and all(r["category_id"] in (24, 26) for r in records)
    assert all(len(r["bbox"]) == 4 and r["bbox"][2] >= 0 and r["bbox"][3] >= 0 and 0.0 <= r["score"] <= 1.0
               for r in records)
    for r in records:     # xywh in the ORIGINAL image's pixels; clipped in network-input pixels with the legacy
        H, W = sizes[r["image_id"] - 100]      # "+1" width convention, so x + w may exceed W by less than one pixel
        x, y, w, h = r["bbox"]
        assert x >= -1e-3 and y >= -1e-3 and x + w < W + 1 and y + h < H + 1, (r, H, W)


@pytest.mark.gpu
def test_triplet_training_entry_point_on_a_tiny_dataset(tmp_path):
    """the triplet recipe (source + foggy target + rainy auxiliary loaders, reference flow tools/train_net_triplet.py:
    118-197) through tools/train_net_da.py on generated datasets"""
    import subprocess
    import sys

    rng = np.random.default_rng(2)
    specs = {k: _write_coco(str(tmp_path), k, 4, rng, sizes=[(96, 192)] * 4) for k in ("source", "target", "auxiliary")}
    out = str(tmp_path / "out")
    os.makedirs(out)
    root = os.path.dirname(HERE)
    yaml = os.path.join(root, "configs/da_faster_rcnn/"
                              "e2e_triplet_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml")
    cmd = [sys.executable, os.path.join(root, "tools", "train_net_da.py"), "--config-file", yaml,
           "--source", ",".join(specs["source"]), "--target", ",".join(specs["target"]),
           "--auxiliary", ",".join(specs["auxiliary"]),
           "SOLVER.MAX_ITER", "3", "SOLVER.CHECKPOINT_PERIOD", "0", "DATALOADER.NUM_WORKERS", "0",
           "INPUT.MIN_SIZE_TRAIN", "(96,)", "INPUT.MAX_SIZE_TRAIN", "192", "MODEL.OUTPUT_DIR", out, "MODEL.WEIGHT", ""]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2500:]
    assert "iter: 0" in res.stderr and "triplet_loss_image: " in res.stderr
    assert os.path.exists(os.path.join(out, "model_final.pth"))


@pytest.mark.gpu
def test_aligned_triplet_recipe_uses_the_aligned_loader(tmp_path):
    """with TRIPLET_USE + ALIGNMENT the reference feeds ONE loader over index-aligned (source, target,
    auxiliary) samples (tools/train_net_triplet.py:123-134, data/build.py:23-63); independent loaders would pool
    unrelated images with the target's proposals.  Without --auxiliary the entry point refuses to start."""
    import subprocess
    import sys

    rng = np.random.default_rng(4)
    specs = {k: _write_coco(str(tmp_path), k, 4, rng, sizes=[(96, 192)] * 4) for k in ("source", "target", "auxiliary")}
    out = str(tmp_path / "out")
    root = os.path.dirname(HERE)
    yaml = os.path.join(root, "configs/da_faster_rcnn/"
                              "e2e_triplet_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml")
    base = [sys.executable, os.path.join(root, "tools", "train_net_da.py"), "--config-file", yaml,
            "--source", ",".join(specs["source"]), "--target", ",".join(specs["target"])]
    opts = ["SOLVER.MAX_ITER", "2", "SOLVER.CHECKPOINT_PERIOD", "0", "DATALOADER.NUM_WORKERS", "0",
            "INPUT.MIN_SIZE_TRAIN", "(96,)", "INPUT.MAX_SIZE_TRAIN", "192", "MODEL.OUTPUT_DIR", out, "MODEL.WEIGHT", "",
            "MODEL.DA_HEADS.ALIGNMENT", "True", "MODEL.DA_HEADS.DA_TRIPLET_INS_WEIGHT", "1.0"]
    res = subprocess.run(base + opts, capture_output=True, text=True, timeout=600)
    assert res.returncode != 0 and "--auxiliary" in res.stderr
    res = subprocess.run(base + ["--auxiliary", ",".join(specs["auxiliary"])] + opts, capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stderr[-2500:]
    assert "triplet_loss_instance: " in res.stderr and os.path.exists(os.path.join(out, "model_final.pth"))


@pytest.mark.gpu
def test_fine_tune_and_resume_from_a_checkpoint_given_as_weight(tmp_path):
    """MODEL.WEIGHT = a checkpoint.  Default (the reference fork: optimizer restore commented out, loop index forced to
    0 — checkpoint.py:62-70, trainer.py:177): weights only, the full schedule runs.  --resume: optimizer / scheduler
    state restored and the loop continues behind the stored iteration; every saved checkpoint carries the LIVE
    optimizer state."""
    import subprocess
    import sys

    rng = np.random.default_rng(3)
    specs = {k: _write_coco(str(tmp_path), k, 4, rng, sizes=[(96, 192)] * 4) for k in ("source", "target")}
    root = os.path.dirname(HERE)
    yaml = os.path.join(root, "configs/da_faster_rcnn/e2e_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml")

    def run(out, weight, max_iter, extra=()):
        cmd = [sys.executable, os.path.join(root, "tools", "train_net_da.py"), "--config-file", yaml,
               "--source", ",".join(specs["source"]), "--target", ",".join(specs["target"])] + list(extra) + [
               "SOLVER.MAX_ITER", str(max_iter), "SOLVER.CHECKPOINT_PERIOD", "2", "DATALOADER.NUM_WORKERS", "0",
               "INPUT.MIN_SIZE_TRAIN", "(96,)", "INPUT.MAX_SIZE_TRAIN", "192", "MODEL.OUTPUT_DIR", out,
               "MODEL.WEIGHT", weight, "SOLVER.WARMUP_ITERS", "2"]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2500:]
        return res.stderr

    first = str(tmp_path / "first" / "nested")           # the output directory is created by the entry point
    run(first, "", 3)
    ck = os.path.join(first, "model_0000002.pth")
    stored = torch.load(ck, map_location="cpu", weights_only=False)
    assert stored["iteration"] == 2
    # fine-tune: full schedule from iteration 0
    log = run(str(tmp_path / "second"), ck, 5)
    assert "Loading checkpoint from " + ck in log and "iter: 0" in log
    final = torch.load(os.path.join(str(tmp_path / "second"), "model_final.pth"), map_location="cpu", weights_only=False)
    assert final["iteration"] == 4
    # resume: iterations 3 and 4 only; momentum buffers came from the checkpoint
    log = run(str(tmp_path / "third"), ck, 5, extra=["--resume"])
    assert "Loading optimizer from " + ck in log and "iter: 0" not in log
    third = torch.load(os.path.join(str(tmp_path / "third"), "model_0000004.pth"), map_location="cpu", weights_only=False)
    assert third["iteration"] == 4
    lr_now = third["optimizer"]["param_groups"][0]["lr"]
    assert lr_now != stored["optimizer"]["param_groups"][0]["lr"], "the saved optimizer state must be the live one"
