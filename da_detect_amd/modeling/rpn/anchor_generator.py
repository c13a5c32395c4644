"""Anchor generation (reference: maskrcnn_benchmark/modeling/rpn/anchor_generator.py:34-125, 222-291)."""
import numpy as np
import torch
from torch import nn

from ...structures.bounding_box import BoxList


class BufferList(nn.Module):
    """nn.ParameterList analogue for buffers (anchor_generator.py:11-31)"""

    def __init__(self, buffers=None):
        super(BufferList, self).__init__()
        if buffers is not None:
            self.extend(buffers)

    def extend(self, buffers):
        offset = len(self)
        for i, b in enumerate(buffers):
            self.register_buffer(str(offset + i), b)
        return self

    def __len__(self):
        return len(self._buffers)

    def __iter__(self):
        return iter(self._buffers.values())


def _center_form(a):
    w, h = a[2] - a[0] + 1, a[3] - a[1] + 1
    return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)


def _corner_form(ws, hs, cx, cy):
    ws, hs = ws[:, None], hs[:, None]
    return np.hstack((cx - 0.5 * (ws - 1), cy - 0.5 * (hs - 1), cx + 0.5 * (ws - 1), cy + 0.5 * (hs - 1)))


def generate_anchors(stride=16, sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
    """cell anchors (x1,y1,x2,y2) centred on a stride x stride cell: for every aspect ratio (rounded widths /
    heights of equal area) every scale size/stride (anchor_generator.py:222-291, Detectron's generate_anchors)."""
    base = np.array([1, 1, stride, stride], dtype=np.float64) - 1
    scales = np.array(sizes, dtype=np.float64) / stride
    ratios = np.array(aspect_ratios, dtype=np.float64)
    w, h, cx, cy = _center_form(base)
    ws = np.round(np.sqrt(w * h / ratios))
    hs = np.round(ws * ratios)
    ratio_anchors = _corner_form(ws, hs, cx, cy)
    out = []
    for a in ratio_anchors:
        w, h, cx, cy = _center_form(a)
        out.append(_corner_form(w * scales, h * scales, cx, cy))
    return torch.from_numpy(np.vstack(out))


class AnchorGenerator(nn.Module):
    def __init__(self, sizes=(128, 256, 512), aspect_ratios=(0.5, 1.0, 2.0), anchor_strides=(8, 16, 32),
                 straddle_thresh=0):
        super(AnchorGenerator, self).__init__()
        if len(anchor_strides) == 1:
            cell_anchors = [generate_anchors(anchor_strides[0], sizes, aspect_ratios).float()]
        else:
            if len(anchor_strides) != len(sizes):
                raise RuntimeError("FPN should have #anchor_strides == #sizes")
            cell_anchors = [
                generate_anchors(s, size if isinstance(size, (tuple, list)) else (size,), aspect_ratios).float()
                for s, size in zip(anchor_strides, sizes)]
        self.strides = anchor_strides
        self.cell_anchors = BufferList(cell_anchors)
        self.straddle_thresh = straddle_thresh
        self._cache = {}
        self.last_call_was_cached = False

    def num_anchors_per_location(self):
        return [len(c) for c in self.cell_anchors]

    def grid_anchors(self, grid_sizes):
        """per level [H*W*A, 4], anchor index = (h*W + w)*A + a (anchor_generator.py:73-97)"""
        anchors = []
        for (gh, gw), stride, base in zip(grid_sizes, self.strides, self.cell_anchors):
            dev = base.device
            sx = torch.arange(0, gw * stride, step=stride, dtype=torch.float32, device=dev)
            sy = torch.arange(0, gh * stride, step=stride, dtype=torch.float32, device=dev)
            yy, xx = torch.meshgrid(sy, sx, indexing="ij")
            xx, yy = xx.reshape(-1), yy.reshape(-1)
            shifts = torch.stack((xx, yy, xx, yy), dim=1)
            anchors.append((shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4))
        return anchors

    def add_visibility_to(self, boxlist):
        w, h = boxlist.size
        a = boxlist.bbox
        if self.straddle_thresh >= 0:
            t = self.straddle_thresh
            inside = (a[..., 0] >= -t) & (a[..., 1] >= -t) & (a[..., 2] < w + t) & (a[..., 3] < h + t)
        else:
            inside = torch.ones(a.shape[0], dtype=torch.bool, device=a.device)
        boxlist.add_field("visibility", inside)

    def forward(self, image_list, feature_maps):
        """anchors depend on the image / feature-map sizes only: the tensors are built once per size signature and
        re-wrapped in fresh BoxLists afterwards (training batches of one crop size hit the cache every step)."""
        grid_sizes = [tuple(fm.shape[-2:]) for fm in feature_maps]
        key = (tuple(tuple(s) for s in image_list.image_sizes), tuple(grid_sizes), str(next(iter(self.cell_anchors)).device))
        self.last_call_was_cached = key in self._cache
        if not self.last_call_was_cached:
            per_level = self.grid_anchors(grid_sizes)
            entry = []
            for (ih, iw) in image_list.image_sizes:
                in_image = []
                for a in per_level:
                    bl = BoxList(a, (iw, ih), mode="xyxy")
                    self.add_visibility_to(bl)
                    in_image.append((a, (iw, ih), bl.get_field("visibility")))
                entry.append(in_image)
            if len(self._cache) >= 8:
                self._cache.clear()
            self._cache[key] = entry
        anchors = []
        for in_image in self._cache[key]:
            row = []
            for a, size, vis in in_image:
                bl = BoxList(a, size, mode="xyxy")
                bl.add_field("visibility", vis)
                row.append(bl)
            anchors.append(row)
        return anchors


def make_anchor_generator(config):
    sizes = config.MODEL.RPN.ANCHOR_SIZES
    stride = config.MODEL.RPN.ANCHOR_STRIDE
    if config.MODEL.RPN.USE_FPN:
        assert len(stride) == len(sizes), "FPN should have len(ANCHOR_STRIDE) == len(ANCHOR_SIZES)"
    else:
        assert len(stride) == 1, "Non-FPN should have a single ANCHOR_STRIDE"
    return AnchorGenerator(sizes, config.MODEL.RPN.ASPECT_RATIOS, stride, config.MODEL.RPN.STRADDLE_THRESH)
