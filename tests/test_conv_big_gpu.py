"""conv_big_kernel (csrc/conv_big.hip): the 256 x 256-tile forward / data-gradient kernel of contraction mode 4.

Checked against a float64 convolution and against the 128 x 128 kernel it replaces on the same inputs (same products,
another summation order), over the paths the kernel has: 1x1 and 3x3 taps with padding and stride, ragged M / Cout, every
epilogue, reductions cut into 1 .. 4 parts (the parked-part hand-over), several tiles per launch.
Reference of the operation: ATen conv2d + FrozenBatchNorm2d + relu_ behind mb/modeling/backbone/resnet.py:294-314 and
mb/modeling/rpn/rpn.py:39-46 (mb = maskrcnn_benchmark)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
CL = torch.channels_last


@pytest.fixture()
def big_mode():
    from da_detect_amd import _lib

    lib = _lib.load()
    prev = lib.dadet_get_big_gemm()
    yield lib
    lib.dadet_set_big_gemm(prev)


def _ref64(x, w, stride, pad, kw):
    y = torch.nn.functional.conv2d(x.double(), w.double(), stride=stride, padding=pad)
    if "scale" in kw:
        y = y * kw["scale"].double().view(1, -1, 1, 1) + kw["bias"].double().view(1, -1, 1, 1)
    if "addend" in kw:
        y = y + kw["addend"].double()
    if kw.get("relu_mode") == 1:
        y = y.clamp_min(0)
    if kw.get("relu_mode") == 2:
        y = y * (kw["mask_ref"] > 0)
    return y


# N, Cin, H, W, Cout, k, stride, pad, epilogue
CASES = [
    (1, 64, 40, 52, 256, 1, 1, 0, "none"),            # 9 tiles in M (ragged: 2080 rows), K = 64: 2 K-tiles, no split
    (2, 128, 24, 40, 512, 3, 1, 1, "affine_relu"),    # 3x3 with padding, 8 x 2 tiles, K = 1152
    (1, 256, 33, 47, 300, 3, 1, 1, "add_relu"),       # ragged M (1551) and Cout (300): 7 x 2 tiles
    (2, 512, 16, 24, 256, 1, 1, 0, "add_gate"),       # 3 tiles -> K = 512 cut into 2 parts (16 K-tiles)
    (1, 1024, 16, 16, 256, 1, 1, 0, "gate"),          # 1 tile  -> 4 parts of 8 K-tiles
    (1, 256, 20, 20, 512, 3, 2, 1, "affine"),         # stride 2, 3x3: 100 rows, 2 tiles, K = 2304 -> 4 parts
    (3, 96, 14, 14, 260, 3, 1, 1, "add_relu"),        # Cin = 96 (three K-tiles per tap), 588 rows, ragged Cout
    (1, 32, 64, 64, 512, 1, 1, 0, "none"),            # K = 32: one K-tile
]


@pytest.mark.parametrize("tile_n", [256, 128])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_c%d_%dx%d_o%d_k%d_s%d_p%d_%s" % c)
def test_big_tile_kernel_against_float64_and_the_128_tile_kernel(device, big_mode, case, tile_n, monkeypatch):
    from da_detect_amd import _C

    monkeypatch.setenv("DADET_BIG_TILE_N", str(tile_n))   # conv_big_kernel<256> / conv_big128_kernel

    N, Cin, H, W, Cout, k, stride, pad, epi = case
    g = torch.Generator().manual_seed(sum(case[:8]))
    x = torch.randn((N, Cin, H, W), generator=g).to(device).contiguous(memory_format=CL)
    w = (torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (Cin * k * k)) ** 0.5).to(device).contiguous(memory_format=CL)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    kw = {}
    if epi in ("affine", "affine_relu", "add_relu", "add_gate"):
        kw["scale"] = (torch.rand(Cout, generator=g) + 0.5).to(device)
        kw["bias"] = torch.randn(Cout, generator=g).to(device)
    if epi in ("add_relu", "add_gate"):
        kw["addend"] = torch.randn((N, Cout, Ho, Wo), generator=g).to(device).contiguous(memory_format=CL)
    if epi in ("affine_relu", "add_relu"):
        kw["relu_mode"] = 1
    if epi in ("add_gate", "gate"):
        kw["relu_mode"] = 2
        kw["mask_ref"] = torch.randn((N, Cout, Ho, Wo), generator=g).clamp_min(0).to(device).contiguous(memory_format=CL)
    out = {}
    for mode in (0, 2):
        big_mode.dadet_set_big_gemm(mode)
        d = _C._desc(N, H, W, Cin, Cout, k, k, stride, pad, Ho, Wo)
        variant = big_mode.dadet_conv_forward_variant(ctypes.byref(d))
        assert (variant == (4 if tile_n == 256 else 5)) == (mode == 2), "mode %d must%s take the large-tile kernel (variant %d)" % (
            mode, "" if mode == 2 else " not", variant)
        out[mode] = _C.conv_forward(x, w, stride=stride, pad=pad, **kw)
        again = _C.conv_forward(x, w, stride=stride, pad=pad, **kw)
        assert torch.equal(out[mode], again), "two runs of one launch must agree bit for bit (parts summed in part order)"
    ref = _ref64(x, w, stride, pad, kw)
    errs = {}
    for mode in (0, 2):
        e = (out[mode].double() - ref).abs()
        errs[mode] = (float(e.pow(2).mean().sqrt()), float(e.max()))
    top = max(1.0, float(ref.abs().max()))
    assert errs[2][1] <= 2e-5 * top, errs
    # RMS error: fp32 accumulation error grows with the length of ONE accumulator's reduction, and the two kernels cut K
    # differently (tools/probes/big_parts_error.py: K = 2304 in one part 7.2e-7, in four 3.8e-7, the 128 x 128 kernel's
    # split-K launch 2.3e-7) — same class, not the same number
    assert errs[2][0] <= 4.0 * errs[0][0] + 1e-9 * (float(ref.abs().mean()) + 1e-30), errs
    # ReLU at an exact tie may fire on one side only; everything else agrees to fp32 rounding
    torch.testing.assert_close(out[2], out[0], rtol=2e-5, atol=2e-5 * top)


@pytest.mark.parametrize("tile_n", [256, 128])
@pytest.mark.parametrize("splits", [1, 2, 3, 4])
def test_big_tile_kernel_part_counts_agree(device, big_mode, splits, tile_n, monkeypatch):
    """the same 3x3 layer with its reduction in 1 .. 4 parts: fp32 rounding apart, and run-to-run identical"""
    from da_detect_amd import _C

    g = torch.Generator().manual_seed(11)
    x = torch.randn((2, 128, 20, 28), generator=g).to(device).contiguous(memory_format=CL)
    w = (torch.randn((320, 128, 3, 3), generator=g) * 0.03).to(device).contiguous(memory_format=CL)
    big_mode.dadet_set_big_gemm(2)
    monkeypatch.setenv("DADET_BIG_TILE_N", str(tile_n))
    monkeypatch.setenv("DADET_BIG_SPLITS", "1")
    one = _C.conv_forward(x, w, pad=1)
    monkeypatch.setenv("DADET_BIG_SPLITS", str(splits))
    got = _C.conv_forward(x, w, pad=1)
    for _ in range(3):
        assert torch.equal(got, _C.conv_forward(x, w, pad=1))
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    assert float((got.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    torch.testing.assert_close(got, one, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))


@pytest.mark.parametrize("tile_n", [256, 128])
@pytest.mark.parametrize("epi", ["none", "add_gate"])
def test_two_parts_meet_symmetrically_and_give_the_bits_of_the_one_sided_hand_over(device, big_mode, tile_n, epi, monkeypatch):
    """round 6: two K parts of a tile exchange HALVES (each parks the row blocks it does not own, adds the partner's to its
    own, stores its own) instead of part 0 parking everything for part 1.  a + b is b + a: the output must be the round-5
    hand-over's (DADET_BIG_ASYM=1) bit for bit — on ragged tiles, with a fused epilogue, and launch after launch on one
    stream with CHANGING tile counts (the meeting's words are never reset: per launch and tile the ticket counter moves by
    three, `started` by two, the flags carry the launch's epoch).  A third launch shape in between uses the 3-part protocol, whose counters are reset."""
    from da_detect_amd import _C

    big_mode.dadet_set_big_gemm(2)
    monkeypatch.setenv("DADET_BIG_TILE_N", str(tile_n))
    g = torch.Generator().manual_seed(21 + tile_n)
    # the last shape is 300 (600) tiles in two parts on 256 CUs: the first round's parts find partners that have not been
    # DISPATCHED — they must not wait for them (they park their whole tile and leave; the late partner finishes the tile)
    shapes = [(2, 128, 33, 47, 320, 3, 1), (1, 256, 24, 40, 512, 1, 0), (3, 64, 20, 28, 260, 3, 1), (1, 64, 240, 320, 256, 1, 0)]
    data = []
    for N, Cin, H, W, Cout, k, pad in shapes:
        x = torch.randn((N, Cin, H, W), generator=g).to(device).contiguous(memory_format=CL)
        w = (torch.randn((Cout, Cin, k, k), generator=g) * 0.03).to(device).contiguous(memory_format=CL)
        kw = {}
        if epi == "add_gate":
            kw = dict(scale=(torch.rand(Cout, generator=g) + 0.5).to(device), bias=torch.randn(Cout, generator=g).to(device),
                      addend=torch.randn((N, Cout, H, W), generator=g).to(device).contiguous(memory_format=CL), relu_mode=2,
                      mask_ref=torch.randn((N, Cout, H, W), generator=g).clamp_min(0).to(device).contiguous(memory_format=CL))
        data.append((x, w, pad, kw))
    monkeypatch.setenv("DADET_BIG_SPLITS", "2")
    monkeypatch.setenv("DADET_BIG_ASYM", "1")
    want = [_C.conv_forward(x, w, pad=pad, **kw) for x, w, pad, kw in data]
    monkeypatch.setenv("DADET_BIG_ASYM", "0")
    for rnd in range(6):
        for i in ((0, 1, 2, 3), (3, 2, 0, 1), (1, 3, 1, 0))[rnd % 3]:
            x, w, pad, kw = data[i]
            got = _C.conv_forward(x, w, pad=pad, **kw)
            assert torch.equal(got, want[i]), "round %d, shape %d: symmetric meeting != one-sided hand-over" % (rnd, i)
        if rnd == 2:      # the reset-to-zero protocol of three parts shares the stream (its own counter words)
            monkeypatch.setenv("DADET_BIG_SPLITS", "3")
            x, w, pad, kw = data[0]
            three = _C.conv_forward(x, w, pad=pad, **kw)
            torch.testing.assert_close(three, want[0], rtol=1e-5, atol=1e-5 * float(want[0].abs().max()))
            monkeypatch.setenv("DADET_BIG_SPLITS", "2")
    ref = torch.nn.functional.conv2d(data[0][0].double(), data[0][1].double(), padding=1)
    if epi == "none":
        assert float((want[0].double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("shape", [(1, 128, 257, 256, 256, 1), (512, 128, 7, 7, 2048, 1), (2, 64, 200, 330, 256, 3)],
                         ids=["257_tiles", "784_tiles", "516_tiles_3x3"])
def test_only_the_last_partly_filled_round_of_tiles_is_cut(device, big_mode, shape, monkeypatch):
    """round 6 (big_tail_plan): a grid of a few tiles more than a multiple of 256 — 784 for the res5 head on 512 ROIs — runs
    its first 256 k tiles whole and cuts only the tiles of the last round in two (symmetric meeting), instead of spending a
    whole tile's time on a nearly empty round.  Same results as without the cut (DADET_BIG_TAIL=0) up to the summation
    order of the cut tiles, and against float64; launch after launch (the meeting's words are indexed by tile)."""
    from da_detect_amd import _C

    N, Cin, H, W, Cout, k = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn((N, Cin, H, W), generator=g).to(device).contiguous(memory_format=CL)
    w = (torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (Cin * k * k)) ** 0.5).to(device).contiguous(memory_format=CL)
    add = torch.randn((N, Cout, H, W), generator=g).to(device).contiguous(memory_format=CL)
    big_mode.dadet_set_big_gemm(2)
    monkeypatch.setenv("DADET_BIG_TILE_N", "256")
    tiles = -(-(N * H * W) // 256) * -(-Cout // 256)
    assert tiles > 256 and 0 < tiles % 256 <= 64
    monkeypatch.setenv("DADET_BIG_TAIL", "0")
    whole = _C.conv_forward(x, w, pad=k // 2, addend=add, relu_mode=1)
    monkeypatch.setenv("DADET_BIG_TAIL", "1")
    for _ in range(3):
        cut = _C.conv_forward(x, w, pad=k // 2, addend=add, relu_mode=1)
        assert k == 1 or not torch.equal(cut, whole), "the tail cut did not change a single sum: was it taken?"
        torch.testing.assert_close(cut, whole, rtol=1e-5, atol=2e-6 * float(whole.abs().max()))
    # the rows of the whole tiles are the same bits; the cut tiles are the LAST ones
    body_rows = (tiles - tiles % 256) // (-(-Cout // 256)) * 256
    flat_c, flat_w = cut.permute(0, 2, 3, 1).reshape(-1, Cout), whole.permute(0, 2, 3, 1).reshape(-1, Cout)
    assert torch.equal(flat_c[: body_rows - 256], flat_w[: body_rows - 256])
    ref = torch.nn.functional.conv2d(x[:1].double(), w.double(), padding=k // 2)
    got1 = cut[:1].double()
    want1 = (ref + add[:1].double()).clamp_min(0)
    assert float((got1 - want1).abs().max()) <= 2e-5 * float(want1.abs().max())
    _C.check_nonfinite()


def test_big_tile_kernel_leaves_the_output_maximum(device, big_mode):
    """mode 4's hand-over: the epilogue merges max|y| into the caller's slot (the next GEMM's scale)"""
    from da_detect_amd import _C, amax as _amax

    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, 256, 24, 24), generator=g).to(device).contiguous(memory_format=CL)
    w = (torch.randn((512, 256, 3, 3), generator=g) * 0.03).to(device).contiguous(memory_format=CL)
    big_mode.dadet_set_big_gemm(2)
    y = _C.conv_forward(x, w, pad=1, relu_mode=1)
    assert _amax.value(y) == float(y.abs().max())


# N, Cin, H, W, Cout, k, stride, pad
WGRAD_CASES = [
    (1, 64, 24, 40, 256, 1, 1, 0),       # one tile, M = 960 rows
    (2, 128, 20, 28, 300, 3, 1, 1),      # 3x3 with padding, ragged Cout, K = 1152: 2 x 5 tiles, a tap change inside a tile
    (1, 256, 33, 47, 256, 3, 1, 1),      # ragged M (1551 rows: the last K-tile is partly beyond M)
    (2, 96, 14, 14, 260, 3, 1, 1),       # Cin = 96: column tiles straddle filter taps, 7 x 7-like small maps (Wo < 32)
    (1, 512, 30, 30, 512, 1, 2, 0),      # stride 2, 1x1 (the projection shortcut's gradient)
    (4, 64, 7, 7, 256, 3, 1, 1),         # 7 x 7 maps: 32 rows span 4.6 map rows and an image boundary
    (1, 32, 64, 64, 512, 3, 2, 1),       # stride 2, 3x3
]


@pytest.mark.parametrize("splits", [0, 1, 3])
@pytest.mark.parametrize("case", WGRAD_CASES, ids=lambda c: "n%d_c%d_%dx%d_o%d_k%d_s%d_p%d" % c)
def test_big_tile_weight_gradient_against_float64(device, big_mode, case, splits, monkeypatch):
    """conv_wgrad_big_kernel against the float64 weight gradient and against the 128 x 128 kernel, with the reduction over
    the pixels in the planned number of parts (0), one part (the kernel writes dW itself, scale and accumulation
    included) and three parts (partial sums + the deterministic reduction pass)"""
    from da_detect_amd import _C

    N, Cin, H, W, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn((N, Cin, H, W), generator=g).to(device).contiguous(memory_format=CL)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    gy = torch.randn((N, Cout, Ho, Wo), generator=g).to(device).contiguous(memory_format=CL)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(device)
    prev = torch.randn((Cout, Cin, k, k), generator=g).to(device).contiguous(memory_format=CL)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, k, k), gy.double(), stride=stride, padding=pad)
    ref = ref * scale.double().view(-1, 1, 1, 1) + prev.double()
    if splits:
        monkeypatch.setenv("DADET_WGRAD_BIG_SPLITS", str(splits))
    out = {}
    for mode in (0, 2):
        big_mode.dadet_set_big_gemm(mode)
        dw = prev.clone()
        out[mode] = _C.conv_wgrad(x, gy, (Cout, Cin, k, k), stride=stride, pad=pad, out_scale=scale, dw=dw, accumulate=True)
        dw2 = prev.clone()
        again = _C.conv_wgrad(x, gy, (Cout, Cin, k, k), stride=stride, pad=pad, out_scale=scale, dw=dw2, accumulate=True)
        assert torch.equal(out[mode], again)
    top = float(ref.abs().max())
    errs = {m: float((out[m].double() - ref).abs().max()) for m in out}
    assert errs[2] <= 2e-5 * top, errs
    torch.testing.assert_close(out[2], out[0], rtol=2e-5, atol=2e-5 * top)


def test_big_tile_weight_gradient_through_the_batched_reduction(device, big_mode):
    """the deferred form the training step uses: partial sums now, one reduction launch for several layers later"""
    from da_detect_amd import _C

    g = torch.Generator().manual_seed(3)
    big_mode.dadet_set_big_gemm(2)
    batch = _C.WgradBatch()
    items = []
    for (Cin, Cout, k, pad) in [(64, 256, 3, 1), (256, 512, 1, 0)]:
        x = torch.randn((2, Cin, 24, 24), generator=g).to(device).contiguous(memory_format=CL)
        gy = torch.randn((2, Cout, 24, 24), generator=g).to(device).contiguous(memory_format=CL)
        dw = _C.conv_wgrad(x, gy, (Cout, Cin, k, k), pad=pad, pending=batch)
        items.append((x, gy, dw, (Cout, Cin, k, k), pad))
    _C.conv_wgrad_reduce_batch(batch)
    for x, gy, dw, shape, pad in items:
        ref = torch.nn.grad.conv2d_weight(x.double(), shape, gy.double(), padding=pad)
        assert float((dw.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


GROUPS = [
    # a bottleneck block's backward: conv3, the 3x3 conv2, conv1 and the stride-2 projection shortcut (ragged everything)
    [(2, 96, 20, 28, 300, 1, 1, 0), (2, 96, 20, 28, 96, 3, 1, 1), (2, 160, 40, 56, 96, 1, 2, 0), (2, 160, 40, 56, 300, 1, 2, 0)],
    # two problems with different row counts (parts per problem differ), one of them a single tile
    [(1, 64, 24, 40, 256, 1, 1, 0), (8, 64, 7, 7, 256, 3, 1, 1)],
    [(1, 256, 33, 47, 256, 3, 1, 1)],      # a group of one
]


@pytest.mark.parametrize("rows", [0, 128, 4096])
@pytest.mark.parametrize("group", GROUPS, ids=lambda g: "%dproblems_%d" % (len(g), g[0][1]))
def test_grouped_weight_gradients_against_float64(device, big_mode, group, rows, monkeypatch):
    """dadet_conv_wgrad_group: several weight gradients in ONE launch of the 256 x 256-tile kernel, each accumulated into
    its own buffer with its FrozenBN scale — against float64 and against the per-layer launches (same products, another
    cut of the reduction), with the planned rows per part, the smallest (128: many parts) and one part per problem (4096:
    the kernel writes dW itself)"""
    big_mode.dadet_set_big_gemm(2)
    _check_group(device, group, rows, monkeypatch, 256)


NARROW_GROUPS = [
    # a res3-like block: 1x1 up, 3x3, 1x1 down on 128 / 512 channels (and ragged ones), maps wider than 32 pixels
    [(2, 128, 24, 40, 512, 1, 1, 0), (2, 128, 24, 40, 128, 3, 1, 1), (2, 500, 24, 40, 128, 1, 1, 0)],
    # small maps (Wo < 32: the kernel's other row walk), a stride-2 member, four problems
    [(4, 64, 14, 14, 256, 1, 1, 0), (4, 64, 14, 14, 64, 3, 1, 1), (4, 96, 28, 28, 64, 1, 2, 0), (4, 96, 28, 28, 256, 1, 2, 0)],
]


@pytest.mark.parametrize("rows", [0, 128, 4096])
@pytest.mark.parametrize("group", NARROW_GROUPS, ids=lambda g: "%dproblems_%d" % (len(g), g[0][1]))
def test_grouped_weight_gradients_of_narrow_layers_against_float64(device, big_mode, group, rows, monkeypatch):
    """the same for layers of fewer than 256 channels (res2 / res3): the grouped form of the 128 x 128 kernel"""
    big_mode.dadet_set_big_gemm(1)
    _check_group(device, group, rows, monkeypatch, 128)


def _check_group(device, group, rows, monkeypatch, kind):
    from da_detect_amd import _C, _lib

    if rows:
        monkeypatch.setenv("DADET_WGRAD_GROUP_ROWS", str(rows))
    g = torch.Generator().manual_seed(11 + len(group))
    reqs, refs, singles, prevs = [], [], [], []
    for (N, Cin, H, W, Cout, k, stride, pad) in group:
        x = torch.randn((N, Cin, H, W), generator=g).to(device).contiguous(memory_format=CL)
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        gy = torch.randn((N, Cout, Ho, Wo), generator=g).to(device).contiguous(memory_format=CL)
        scale = (torch.rand(Cout, generator=g) + 0.5).to(device)
        prev = torch.randn((Cout, Cin, k, k), generator=g).to(device).contiguous(memory_format=CL)
        ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, k, k), gy.double(), stride=stride, padding=pad)
        refs.append(ref * scale.double().view(-1, 1, 1, 1) + prev.double())
        singles.append(_C.conv_wgrad(x, gy, (Cout, Cin, k, k), stride=stride, pad=pad, out_scale=scale, dw=prev.clone(),
                                     accumulate=True))
        prevs.append(prev)
        reqs.append(dict(x=x, gy=gy, weight_shape=(Cout, Cin, k, k), stride=stride, pad=pad, out_scale=scale,
                         dw=None, accumulate=True))
    results = []
    for rep in range(2):
        for r, prev in zip(reqs, prevs):
            r["dw"] = prev.clone()
        batch = _C.WgradBatch()
        assert _C.conv_wgrad_group(reqs, batch)
        if rep == 0:
            n = len(reqs)
            descs = (_lib.ConvDesc * n)(*[
                _C._desc(r["x"].shape[0], r["x"].shape[2], r["x"].shape[3], r["x"].shape[1], r["weight_shape"][0],
                         r["weight_shape"][2], r["weight_shape"][3], r["stride"], r["pad"], r["gy"].shape[2], r["gy"].shape[3])
                for r in reqs])
            assert _lib.load().dadet_conv_wgrad_group_plan(descs, n, (ctypes.c_int * n)(), (ctypes.c_size_t * n)()) == kind
        if rows == 4096:
            assert not batch          # one part per problem: nothing left to reduce
        _C.conv_wgrad_reduce_batch(batch)
        results.append([r["dw"] for r in reqs])
    for got, again, ref, one in zip(results[0], results[1], refs, singles):
        top = float(ref.abs().max())
        assert torch.equal(got, again)                  # deterministic: fixed cut, fixed order of the partial sums
        assert float((got.double() - ref).abs().max()) <= 2e-5 * top
        torch.testing.assert_close(got, one, rtol=2e-5, atol=2e-5 * top)


def test_grouped_weight_gradients_refuse_what_the_kernel_does_not_cover(device, big_mode):
    """a member with gy rows padded beyond Cout (the offset branch of a deformable block: 18 channels in rows of 20), more
    than four problems, or another contraction mode: conv_wgrad_group returns False and launches nothing"""
    from da_detect_amd import _C

    big_mode.dadet_set_big_gemm(1)
    x = torch.randn((1, 64, 16, 16), device=device).contiguous(memory_format=CL)
    gy20 = torch.randn((1, 20, 16, 16), device=device).contiguous(memory_format=CL)
    dw18 = torch.zeros((18, 64, 1, 1), device=device).contiguous(memory_format=CL)
    batch = _C.WgradBatch()
    assert not _C.conv_wgrad_group([dict(x=x, gy=gy20, weight_shape=(18, 64, 1, 1), dw=dw18, accumulate=True)], batch)
    assert not batch and float(dw18.abs().max()) == 0.0
    gy = torch.randn((1, 64, 16, 16), device=device).contiguous(memory_format=CL)
    dw = torch.zeros((64, 64, 1, 1), device=device).contiguous(memory_format=CL)
    req = dict(x=x, gy=gy, weight_shape=(64, 64, 1, 1), dw=dw, accumulate=True)
    assert not _C.conv_wgrad_group([dict(req, dw=dw.clone()) for _ in range(5)], batch)
    mode = _C.get_gemm_mode()
    try:
        _C.set_gemm_mode(3)
        assert not _C.conv_wgrad_group([req], batch)
    finally:
        _C.set_gemm_mode(mode)
    assert not batch and float(dw.abs().max()) == 0.0
    assert _C.conv_wgrad_group([req], batch)            # ... and as it is, it is a group of one on the 128 x 128 kernel
    _C.conv_wgrad_reduce_batch(batch)
    ref = torch.nn.grad.conv2d_weight(x.double(), (64, 64, 1, 1), gy.double())
    assert float((dw.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("big", [0, 2])
def test_a_low_operand_maximum_is_reported_by_name_not_as_a_nan_loss(device, big_mode, big):
    """mode 4's guard (dadet_nonfinite_poll): a slot that claims a maximum far below the data makes the scaled operand
    overflow fp16; the GEMM records its launch and `_C.check_nonfinite()` names it — for the 128 x 128 kernel, the large-tile
    kernels, and the weight gradient"""
    from da_detect_amd import _C, amax as _amax

    g = torch.Generator().manual_seed(9)
    x = (torch.randn((1, 256, 24, 24), generator=g) * 3).to(device).contiguous(memory_format=CL)
    w = (torch.randn((256, 256, 3, 3), generator=g) * 0.03).to(device).contiguous(memory_format=CL)
    big_mode.dadet_set_big_gemm(big)
    _C.check_nonfinite()                                   # clean so far (also clears earlier tests' state)
    y = _C.conv_forward(x, w, pad=1)
    _C.check_nonfinite()                                   # honest maxima: nothing to report
    assert bool(torch.isfinite(y).all())
    low = _amax.new_slot(x.device)
    # the slot says max|x| = 2^-12: the kernel scales x by 2^26 and the leading fp16 terms overflow
    slot_view = low[1].view(8, -1)
    slot_view[0, (low[0] - low[1].data_ptr()) // 4] = 2.0 ** -12
    _amax.attach(x, low)
    y = _C.conv_forward(x, w, pad=1)
    assert not bool(torch.isfinite(y).all())
    with pytest.raises(FloatingPointError) as err:
        _C.check_nonfinite()
    assert "conv_forward" in str(err.value) and "M=576" in str(err.value) and "K=2304" in str(err.value), str(err.value)
    _C.check_nonfinite()                                   # reported once, then clean again
    gy = torch.randn((1, 256, 24, 24), generator=g).to(device).contiguous(memory_format=CL)
    _C.conv_wgrad(x, gy, (256, 256, 3, 3), pad=1)          # the weight gradient reads the same lying slot
    with pytest.raises(FloatingPointError) as err:
        _C.check_nonfinite()
    assert "conv_wgrad" in str(err.value), str(err.value)


def test_the_trainer_leaves_in_order_when_a_gemm_overflows(device, big_mode, caplog):
    """engine.trainer._loss_is_nan: the guard's FloatingPointError is caught, logged with the launch's
    name, and turned into the same verdict as a NaN loss — the training loop then closes its tuner and returns instead of
    dying with a traceback; a finite loss and clean GEMMs say False"""
    import logging

    from da_detect_amd import _C, amax as _amax
    from da_detect_amd.engine.trainer import _loss_is_nan

    g = torch.Generator().manual_seed(10)
    x = (torch.randn((1, 256, 24, 24), generator=g) * 3).to(device).contiguous(memory_format=CL)
    w = (torch.randn((256, 256, 3, 3), generator=g) * 0.03).to(device).contiguous(memory_format=CL)
    net = torch.nn.Linear(3, 2).to(device)
    _C.check_nonfinite()
    side = torch.cuda.Stream()                             # a NON-blocking side stream: the poll must still see its launch
    _C.conv_forward(x, w, pad=1)
    assert _loss_is_nan(net, torch.ones((), device=device)) is False
    low = _amax.new_slot(x.device)
    low[1].view(8, -1)[0, (low[0] - low[1].data_ptr()) // 4] = 2.0 ** -12
    _amax.attach(x, low)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        _C.conv_forward(x, w, pad=1)
    with caplog.at_level(logging.CRITICAL, logger="maskrcnn_benchmark.trainer"):
        assert _loss_is_nan(net, torch.ones((), device=device)) is True
    assert any("conv_forward" in r.getMessage() for r in caplog.records), [r.getMessage() for r in caplog.records]
    torch.cuda.current_stream().wait_stream(side)
    assert _loss_is_nan(net, torch.ones((), device=device)) is False      # reported once
