"""which pieces of the device-side proposal selection can be captured into a HIP graph (one capture attempt per piece)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from da_detect_amd import _C

dev = torch.device("cuda:0")
torch.manual_seed(0)
scores = torch.rand(2, 49152, device=dev)
boxes = torch.rand(2000, 4, device=dev) * 500
boxes[:, 2:] += boxes[:, :2] + 4
boxes = boxes.contiguous()
deltas = torch.randn(49152, 4, device=dev) * 0.1
anch = torch.rand(49152, 4, device=dev) * 500
anch[:, 2:] += anch[:, :2] + 8
idx = torch.randint(0, 49152, (2000,), device=dev)
m = torch.rand(4000, device=dev) > 0.5
sc = torch.rand(4000, device=dev)


def attempt(name, fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = fn()
        g.replay()
        torch.cuda.synchronize()
        print("%-28s captured" % name, flush=True)
    except Exception as e:   # noqa: BLE001
        print("%-28s FAILED: %s" % (name, str(e).splitlines()[0][:110]), flush=True)
        torch.cuda.synchronize()


attempt("sigmoid", lambda: scores.sigmoid())
attempt("sort stable desc", lambda: torch.sort(scores, dim=1, descending=True, stable=True))
attempt("rpn_decode_clip", lambda: _C.rpn_decode_clip(deltas, anch, idx, (1.0, 1.0, 1.0, 1.0), 4.135, 2048, 1024))
attempt("nms_with_count", lambda: _C.nms_with_count(boxes, None, 0.7, max_keep=2000))
attempt("topk", lambda: torch.topk(sc, 2000, dim=0, sorted=True))
def mask_fn():
    mk = torch.zeros_like(sc, dtype=torch.bool)
    mk[idx.clamp(max=3999)] = True
    return mk & (sc >= 0)
attempt("mask index_put", mask_fn)
attempt("nonzero_static", lambda: torch.nonzero_static(m, size=4000, fill_value=0))
attempt("sum -> int32", lambda: m.sum().to(torch.int32).reshape(1))
def merge():
    keep, cnt = _C.nms_with_count(boxes, None, 0.7, max_keep=2000)
    return _C.fpn_merge_levels([[(boxes, sc[:2000].contiguous(), keep, cnt)]], 2000)
attempt("nms + fpn_merge_levels", merge)
def side():
    main = torch.cuda.current_stream(dev)
    s2 = torch.cuda.Stream(dev)
    s2.wait_stream(main)
    with torch.cuda.stream(s2):
        r = _C.nms_with_count(boxes, None, 0.7, max_keep=2000)
    main.wait_stream(s2)
    return r
attempt("nms on a forked stream", side)
