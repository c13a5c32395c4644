import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from da_detect_amd import _C
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
def boxes(n, W=2048, H=1024, side=300):
    xy = np.stack([rng.uniform(0, W - 2, n), rng.uniform(0, H - 2, n)], 1)
    wh = np.stack([rng.uniform(8, side, n), rng.uniform(8, side, n)], 1)
    return torch.from_numpy(np.concatenate([xy, np.minimum(xy + wh, [W - 1, H - 1])], 1).astype(np.float32)).to(dev)
for n, side in ((12000, 300), (12000, 80), (12000, 600)):
    b = boxes(n, side=side)
    for mk in (2000, -1):
        keep, cnt = _C.nms_with_count(b, None, 0.7, max_keep=mk)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            _C.nms_with_count(b, None, 0.7, max_keep=mk)
        e.record(); torch.cuda.synchronize()
        print("n %d side %d max_keep %d: kept %d, %.3f ms per NMS (mask + sweep + compaction)" % (n, side, mk, int(cnt), s.elapsed_time(e) / 20))
