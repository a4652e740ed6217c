"""Per-kernel parity on a real MI355X: every C-ABI entry point against the CPU oracle's arithmetic
(stock PyTorch CPU ops, which is what the reference executes) on seeded inputs, ragged shapes included.

fp32 mode must match to fp32 round-off (different summation order only); bf16 mode is compared against the
same computation on bf16-rounded operands with fp32 accumulation."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from mdeical_image_segmentation_amd import ops
    return ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def to_nhwc(x, dtype, ctot=None, c0=0):
    """CPU NCHW/NCDHW fp32 -> device channels-last tensor of `dtype` (optionally inside a wider buffer)."""
    nd = x.dim()
    perm = (0, 2, 3, 1) if nd == 4 else (0, 2, 3, 4, 1)
    xl = x.permute(*perm).contiguous()
    if ctot is None:
        return xl.to(dtype).to(DEV)
    buf = torch.zeros(*xl.shape[:-1], ctot, dtype=dtype, device=DEV)
    buf[..., c0:c0 + xl.shape[-1]] = xl.to(dtype).to(DEV)
    return buf


def from_nhwc(y):
    nd = y.dim()
    perm = (0, 3, 1, 2) if nd == 4 else (0, 4, 1, 2, 3)
    return y.float().cpu().permute(*perm).contiguous()


def q(x, dtype):
    """round to the storage dtype and back (what the kernel actually sees)"""
    return x.to(dtype).float()


def tol(dtype, k):
    # fp32: different summation order over K terms; bf16: output rounding dominates
    if dtype == torch.float32:
        return dict(rtol=2e-5, atol=2e-6 * max(1.0, k ** 0.5))
    return dict(rtol=1.6e-2, atol=2e-2)


def assert_close(a, b, what, **kw):
    a, b = a.float().cpu(), b.float().cpu()
    if not torch.allclose(a, b, **kw):
        d = (a - b).abs()
        idx = np.unravel_index(int(d.argmax()), d.shape)
        raise AssertionError(f"{what}: max|diff| {d.max().item():.4g} at {idx} (got {a[idx].item():.6g} want {b[idx].item():.6g}), "
                             f"mean|diff| {d.mean().item():.4g}, ref absmax {b.abs().max().item():.4g}, "
                             f"frac bad {(d > kw.get('atol', 0) + kw.get('rtol', 0) * b.abs()).float().mean().item():.4g}")


# ---------------------------------------------------------------------------------------------------------
def test_fragment_probes():
    ops = _ops()
    a = rnd(16, 32, seed=1).to(torch.bfloat16).float()
    b = rnd(32, 16, seed=2).to(torch.bfloat16).float()
    c = torch.zeros(16, 16, device=DEV)
    ops.probe_mfma(0, a.to(DEV), b.to(DEV), c)
    assert_close(c, a @ b, "bf16 16x16x32 lane map", rtol=1e-5, atol=1e-5)
    a = rnd(16, 16, seed=3)
    b = rnd(16, 16, seed=4)
    c = torch.zeros(16, 16, device=DEV)
    ops.probe_mfma(1, a.to(DEV), b.to(DEV), c)
    assert_close(c, a @ b, "f32 16x16x4 b128 step", rtol=1e-5, atol=1e-5)
    # LDS transpose read: lane (i = lane&15, g = lane>>4) must receive pixels 8g..8g+7 of channel i
    img = torch.arange(32 * 16, dtype=torch.float32).view(32, 16) % 251
    c = torch.zeros(64, 8, device=DEV)
    ops.probe_mfma(2, img.to(DEV), None, c)
    want = torch.zeros(64, 8)
    for lane in range(64):
        i, g = lane & 15, lane >> 4
        want[lane] = img[8 * g:8 * g + 8, i]
    assert_close(c, want, "ds_read_b64_tr_b16 map", rtol=0, atol=0)


CONV_CASES = [
    # N, H, W, Cin, Cout
    (2, 20, 36, 64, 64),      # ragged vs the 16x16 tile, BN=64 config
    (1, 16, 32, 128, 128),    # BN=128 config
    (2, 9, 17, 64, 256),      # ragged, two cout tiles
    (1, 32, 32, 256, 64),     # several K chunks
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3_fwd(case, dtype):
    ops = _ops()
    N, H, W, Cin, Cout = case
    x = rnd(N, Cin, H, W, seed=10)
    w = rnd(Cout, Cin, 3, 3, seed=11, scale=(9 * Cin) ** -0.5)
    b = rnd(Cout, seed=12)
    xd = to_nhwc(x, dtype)
    wf = torch.empty(9, Cout, Cin, dtype=dtype, device=DEV)
    ops.pack_conv_weight(w.to(DEV), wf, None)
    y = torch.full((N, H, W, Cout), float("nan"), dtype=dtype, device=DEV)
    ops.conv_igemm(xd, wf, y, ksize=3, Cin=Cin, Cout=Cout, bias=b.to(DEV), relu=True)
    want = F.relu(F.conv2d(q(x, dtype), q(w, dtype), b, padding=1))
    assert_close(from_nhwc(y), want, f"conv3x3 fwd {case} {dtype}", **tol(dtype, 9 * Cin))


@pytest.mark.parametrize("shape", [(2, 256, 256), (3, 250, 203), (1, 512, 272)])
def test_conv3x3_weight_stationary_64_to_64(shape, switches):
    """bf16 64 -> 64 layers of at least 128K pixels run the weight-stationary persistent kernel (conv64_ws_kernel) when the deep-prefetch
    column-segment kernel is switched off: ragged 32x16 tiles, bias + ReLU, ReLU mask (dgrad form) and an output that is a channel slice of a wider buffer."""
    ops = _ops()
    switches("MIS_CONV_NOPPD", 1)
    dtype = torch.bfloat16
    N, H, W = shape
    x = rnd(N, 64, H, W, seed=30)
    w = rnd(64, 64, 3, 3, seed=31, scale=(9 * 64) ** -0.5)
    b = rnd(64, seed=32)
    xd = to_nhwc(x, dtype)
    wf = torch.empty(9, 64, 64, dtype=dtype, device=DEV)
    ops.pack_conv_weight(w.to(DEV), wf, None)
    ybuf = torch.full((N, H, W, 128), float("nan"), dtype=dtype, device=DEV)
    ops.conv_igemm(xd, wf, ops.View(ybuf, 64, 64), ksize=3, Cin=64, Cout=64, bias=b.to(DEV), relu=True)
    want = F.relu(F.conv2d(q(x, dtype), q(w, dtype), b, padding=1))
    assert ops.conv_last_dispatch() == "k3.2d.ws64"
    assert_close(from_nhwc(ybuf[..., 64:].contiguous()), want, f"ws64 fwd {shape}", **tol(dtype, 9 * 64))
    assert torch.isnan(ybuf[..., :64].float()).all(), "ws64 wrote outside its channel slice"
    m = rnd(N, 64, H, W, seed=33)
    y2 = torch.full((N, H, W, 64), float("nan"), dtype=dtype, device=DEV)
    ops.conv_igemm(xd, wf, y2, ksize=3, Cin=64, Cout=64, mask=to_nhwc(m, dtype))
    want2 = F.conv2d(q(x, dtype), q(w, dtype), None, padding=1) * (q(m, dtype) > 0)
    assert_close(from_nhwc(y2), want2, f"ws64 mask {shape}", **tol(dtype, 9 * 64))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv3x3_dgrad_mask_and_split_outputs(dtype):
    """dgrad = conv with the mirrored pack; ReLU mask in the epilogue; and the dual-destination epilogue used by
    up_conv.first (first half pixel-unshuffled, second half plain)."""
    ops = _ops()
    N, H, W, Cin, Cout = 2, 12, 20, 128, 64          # forward conv: 128 -> 64
    x = F.relu(rnd(N, Cin, H, W, seed=20))
    w = rnd(Cout, Cin, 3, 3, seed=21, scale=0.05)
    dy = rnd(N, Cout, H, W, seed=22)
    wf = torch.empty(9, Cout, Cin, dtype=dtype, device=DEV)
    wd = torch.empty(9, Cin, Cout, dtype=dtype, device=DEV)
    ops.pack_conv_weight(w.to(DEV), wf, wd)
    xq = q(x, dtype).requires_grad_(True)
    F.conv2d(xq, q(w, dtype), None, padding=1).backward(q(dy, dtype))
    want = xq.grad
    dyd = to_nhwc(dy, dtype)
    # (a) masked single output
    dx = torch.full((N, H, W, Cin), float("nan"), dtype=dtype, device=DEV)
    ops.conv_igemm(dyd, wd, dx, ksize=3, Cin=Cout, Cout=Cin, mask=to_nhwc(x, dtype))
    assert_close(from_nhwc(dx), want * (q(x, dtype) > 0), f"dgrad+mask {dtype}", **tol(dtype, 9 * Cout))
    # (b) split: columns [0,64) unshuffled into (N, H/2, W/2, 256), columns [64,128) plain
    d0 = torch.full((N, H // 2, W // 2, 4 * 64), float("nan"), dtype=dtype, device=DEV)
    d1 = torch.full((N, H, W, 64), float("nan"), dtype=dtype, device=DEV)
    ops.conv_igemm(dyd, wd, d0, ksize=3, Cin=Cout, Cout=Cin, y0_mode=ops.OUT_UNSHUFFLE2, y1=d1, Cout0=64)
    assert_close(from_nhwc(d1), want[:, 64:], f"dgrad split plain half {dtype}", **tol(dtype, 9 * Cout))
    uns = want[:, :64].view(N, 64, H // 2, 2, W // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(N, 256, H // 2, W // 2)
    assert_close(from_nhwc(d0), uns, f"dgrad split unshuffled half {dtype}", **tol(dtype, 9 * Cout))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_convtranspose_k2s2_fwd_dgrad_wgrad(dtype):
    ops = _ops()
    N, H, W, Cin, Cq = 2, 6, 10, 128, 64
    x = F.relu(rnd(N, Cin, H, W, seed=30))
    w = rnd(Cin, Cq, 2, 2, seed=31, scale=0.05)
    b = rnd(Cq, seed=32)
    wf = torch.empty(4 * Cq, Cin, dtype=dtype, device=DEV)
    wd = torch.empty(Cin, 4 * Cq, dtype=dtype, device=DEV)
    ops.pack_convt_weight(w.to(DEV), wf, wd)
    xd = to_nhwc(x, dtype)
    cat = torch.zeros(N, 2 * H, 2 * W, 2 * Cq, dtype=dtype, device=DEV)
    ops.conv_igemm(xd, wf, ops.View(cat, 0, Cq), ksize=1, Cin=Cin, Cout=4 * Cq, bias=b.to(DEV), y0_mode=ops.OUT_SHUFFLE2)
    xq = q(x, dtype).requires_grad_(True)
    wq = q(w, dtype).requires_grad_(True)
    bq = b.clone().requires_grad_(True)
    y = F.conv_transpose2d(xq, wq, bq, stride=2)
    assert_close(from_nhwc(cat)[:, :Cq], y.detach(), f"convT fwd {dtype}", **tol(dtype, Cin))
    assert float(cat[..., Cq:].abs().max()) == 0.0, "convT fwd wrote outside its concat slice"
    dy = rnd(N, Cq, 2 * H, 2 * W, seed=33)
    y.backward(q(dy, dtype))
    # gradient arrives pixel-unshuffled: (N, H, W, 4*Cq) with column ab*Cq + c
    dys = q(dy, dtype).view(N, Cq, H, 2, W, 2).permute(0, 3, 5, 1, 2, 4).reshape(N, 4 * Cq, H, W)
    dysd = to_nhwc(dys, dtype)
    dx = torch.full((N, H, W, Cin), float("nan"), dtype=dtype, device=DEV)
    ops.conv_igemm(dysd, wd, dx, ksize=1, Cin=4 * Cq, Cout=Cin, mask=xd)
    assert_close(from_nhwc(dx), xq.grad * (q(x, dtype) > 0), f"convT dgrad {dtype}", **tol(dtype, 4 * Cq))
    dw = torch.full((Cin, Cq, 2, 2), float("nan"), device=DEV)
    dbf = torch.full((Cq,), float("nan"), device=DEV)
    ops.wgrad(xd, dysd, dw, ksize=1, Cin=Cin, Cout=4 * Cq, dw_layout=1, dbias=dbf)
    assert_close(dbf, bq.grad, f"convT bias grad fused {dtype}", rtol=1e-4, atol=1e-3)
    wt = dict(rtol=2e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=5e-2)
    assert_close(dw, wq.grad, f"convT wgrad {dtype}", **wt)
    db = torch.full((Cq,), float("nan"), device=DEV)
    ops.colsum(dysd, db, fold=4)
    assert_close(db, bq.grad, f"convT bias grad {dtype}", rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("no_tr", ["0", "1"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 20, 36, 64, 64), (1, 16, 16, 128, 64), (3, 9, 17, 64, 128)])
def test_conv3x3_wgrad(case, dtype, no_tr):
    if dtype == torch.float32 and no_tr == "1":
        pytest.skip("f32 has a single path")
    ops = _ops()
    N, H, W, Cin, Cout = case
    x = rnd(N, Cin, H, W, seed=40)
    dy = rnd(N, Cout, H, W, seed=41)
    wq = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    F.conv2d(q(x, dtype), wq, None, padding=1).backward(q(dy, dtype))
    dw = torch.full((Cout, Cin, 3, 3), float("nan"), device=DEV)
    with ops.dispatch_switches(MIS_WGRAD_NO_TR=int(no_tr)):
        dbf = torch.full((Cout,), float("nan"), device=DEV)
        ops.wgrad(to_nhwc(x, dtype), to_nhwc(dy, dtype), dw, ksize=3, Cin=Cin, Cout=Cout, dbias=dbf)
        torch.cuda.synchronize()
    k = N * H * W
    assert_close(dw, wq.grad, f"wgrad {case} {dtype} no_tr={no_tr}", rtol=1e-4, atol=1e-4 * k ** 0.5)
    db = torch.full((Cout,), float("nan"), device=DEV)
    ops.colsum(to_nhwc(dy, dtype), db)
    assert_close(db, q(dy, dtype).sum((0, 2, 3)), "bias grad", rtol=1e-4, atol=1e-3)
    assert_close(dbf, q(dy, dtype).sum((0, 2, 3)), "bias grad fused in wgrad", rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("case", [(2, 20, 36, 64, 64, "k3.2d.f32s"), (3, 9, 17, 32, 64, "k3.2d.f32s32"), (1, 33, 40, 96, 128, "k3.2d.f32s32"), (2, 64, 48, 128, 64, "k3.2d.f32s")])
def test_conv3x3_wgrad_f32_streaming_kernels_2d(case):
    """the 2-D instantiations of wgrad_f32_stream_kernel (no bias gradient: the 3-D engines' form of the call, here on images): 64-channel blocks and, round 6, the
    32-channel blocks (Cin = 32 mod 64) - against torch autograd on CPU, ragged strips, several split-K ranges"""
    ops = _ops()
    N, H, W, Cin, Cout, want = case
    x = rnd(N, Cin, H, W, seed=44)
    dy = rnd(N, Cout, H, W, seed=45)
    wq = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    F.conv2d(x, wq, None, padding=1).backward(dy)
    dw = torch.full((Cout, Cin, 3, 3), float("nan"), device=DEV)
    ops.wgrad(to_nhwc(x, torch.float32), to_nhwc(dy, torch.float32), dw, ksize=3, Cin=Cin, Cout=Cout)
    tag, nsplit = ops.wgrad_last_dispatch()
    assert tag == want, (tag, want)
    k = N * H * W
    assert_close(dw, wq.grad, f"2-D f32 streaming wgrad {case} nsplit={nsplit}", rtol=1e-4, atol=1e-4 * k ** 0.5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin", [1, 3])
def test_first_layer(cin, dtype):
    ops = _ops()
    N, H, W = 2, 20, 28
    x = rnd(N, cin, H, W, seed=50)
    w = rnd(64, cin, 3, 3, seed=51, scale=0.3).requires_grad_(True)
    b = rnd(64, seed=52).requires_grad_(True)
    y = torch.full((N, H, W, 64), float("nan"), dtype=dtype, device=DEV)
    ops.first_conv_fwd(x.to(DEV), w.detach().to(DEV), b.detach().to(DEV), y)
    ref = F.conv2d(x, w, b, padding=1)
    assert_close(from_nhwc(y), F.relu(ref).detach(), f"first conv fwd cin={cin} {dtype}",
                 **(dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)))
    if dtype == torch.bfloat16:          # the ReLU bits of the stored output, written from the same epilogue (both kernels; layout decoded by the dispatch-parity helper)
        from test_gpu_dispatch_parity import _decode_relu_bits
        for sw in ({}, {"MIS_FIRST2D_UNTILED": 1}):
            with ops.dispatch_switches(**sw):
                y2 = torch.full((N, H, W, 64), float("nan"), dtype=dtype, device=DEV)
                rb = torch.full((ops.relu_bits_bytes(N, H, W, 64),), 0x5A, dtype=torch.uint8, device=DEV)
                ops.first_conv_fwd(x.to(DEV), w.detach().to(DEV), b.detach().to(DEV), y2, relu_bits=rb)
            assert torch.equal(_decode_relu_bits(rb, N, H, W, 64), y2.float() > 0), f"first conv relu bits {sw}"
    dy = rnd(N, 64, H, W, seed=53)
    ref.backward(q(dy, dtype))
    dw = torch.full((64, cin, 3, 3), float("nan"), device=DEV)
    db = torch.full((64,), float("nan"), device=DEV)
    ops.first_conv_wgrad(x.to(DEV), to_nhwc(dy, dtype), dw, db)
    assert_close(dw, w.grad, f"first conv wgrad cin={cin}", rtol=1e-4, atol=2e-3)
    assert_close(db, b.grad, f"first conv bias grad cin={cin}", rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool_fwd_bwd_ties(dtype):
    """includes the planted ties of the golden DownSample vectors (first maximum wins)."""
    ops = _ops()
    g = load_golden("g1_blocks2d.npz")
    x4 = torch.from_numpy(g["ds_x"])                     # (2, 4, 10, 14)
    # widen to 8 channels (one 16-byte bf16 chunk) by appending a shifted copy
    x = torch.cat([x4, x4.roll(1, dims=3)], 1)
    gy = rnd(2, 8, 5, 7, seed=60)
    add = rnd(2, 8, 10, 14, seed=61)
    xq = q(x, dtype).requires_grad_(True)
    yq = F.max_pool2d(xq, 2)
    yq.backward(q(gy, dtype))
    xd = to_nhwc(x, dtype)
    y = torch.empty(2, 5, 7, 8, dtype=dtype, device=DEV)
    ops.maxpool2_fwd(xd, y)
    assert torch.equal(from_nhwc(y), yq.detach()), "maxpool fwd"
    if dtype == torch.float32:
        assert torch.equal(from_nhwc(y)[:, :4], torch.from_numpy(g["ds_y"])), "maxpool fwd vs golden"
    dx = to_nhwc(add, dtype)   # in-place accumulate target (aliases `add`)
    ops.maxpool2_bwd(xd, to_nhwc(gy, dtype), dx, add=dx, relu_mask=True)
    want = (xq.grad + q(add, dtype)) * (q(x, dtype) > 0)
    assert_close(from_nhwc(dx), q(want, dtype), f"maxpool bwd {dtype}", rtol=1e-6 if dtype == torch.float32 else 1e-2, atol=1e-6 if dtype == torch.float32 else 1e-2)
    if dtype == torch.float32:
        dx2 = torch.zeros(2, 10, 14, 8, device=DEV)
        ops.maxpool2_bwd(xd, to_nhwc(torch.cat([torch.from_numpy(g["ds_gy"])] * 2, 1), dtype), dx2, add=None, relu_mask=False)
        assert torch.equal(from_nhwc(dx2)[:, :4], torch.from_numpy(g["ds_gx"])), "maxpool bwd vs golden (ties)"
    # the "pool bits" pair (arg-max position + input sign per pooled element instead of re-reading x): bit for bit the results above, ties included
    yb = torch.full_like(y, float("nan"))
    pb = torch.full((2, 5, 7, 8), 0xFF, dtype=torch.uint8, device=DEV)
    ops.maxpool2_fwd(xd, yb, pbits=pb)
    assert torch.equal(yb, y), "maxpool fwd with pool bits"
    sel, pos = pb & 0xF, pb >> 4
    assert torch.equal((sel == 1).int() + (sel == 2).int() + (sel == 4).int() + (sel == 8).int(), torch.ones_like(sel, dtype=torch.int32)), "one arg-max per window"
    xw = to_nhwc(x, dtype).float().view(2, 5, 2, 7, 2, 8).permute(0, 1, 3, 5, 2, 4).reshape(2, 5, 7, 8, 4)       # window position k = kh*2 + kw last
    assert torch.equal(pos, ((xw > 0).to(torch.uint8) << torch.arange(4, device=DEV, dtype=torch.uint8)).sum(-1).to(torch.uint8)), "sign bits"
    for with_add in (True, False):
        dxa = to_nhwc(add, dtype)
        ops.maxpool2_bwd(xd, to_nhwc(gy, dtype), dxa, add=dxa if with_add else None, relu_mask=True)
        dxb = to_nhwc(add, dtype)
        ops.maxpool2_bwd(None, to_nhwc(gy, dtype), dxb, add=dxb if with_add else None, relu_mask=True, pbits=pb)
        assert torch.equal(dxa, dxb), f"maxpool bwd with pool bits (add={with_add})"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,loss", [(2, "ce"), (4, "ce"), (1, "bce"), (3, "bcedice")])
def test_head_loss(C, loss, dtype):
    ops = _ops()
    N, H, W = 2, 12, 20
    y = F.relu(rnd(N, 64, H, W, seed=70))
    w = rnd(C, 64, seed=71, scale=0.2)
    b = rnd(C, seed=72, scale=0.2)
    yq = q(y, dtype).requires_grad_(True)
    wq = w.clone().requires_grad_(True)
    bq = b.clone().requires_grad_(True)
    logits = F.conv2d(yq, wq.view(C, 64, 1, 1), bq)
    g = torch.Generator().manual_seed(73)
    if loss == "ce":
        labels = torch.randint(0, C, (N, H, W), generator=g)
        L = F.cross_entropy(logits, labels)
        kind = ops.LOSS_CE
    else:
        labels = (torch.rand(N, C, H, W, generator=g) > 0.5).float()
        L = F.binary_cross_entropy_with_logits(logits, labels)
        kind = ops.LOSS_BCE
        if loss == "bcedice":
            from oracle import unet3d_oracle as o3
            L = o3.bce_dice_loss(logits, labels)
            kind = ops.LOSS_BCEDICE
    L.backward()
    yd = to_nhwc(y, dtype)
    lg = torch.full((N, C, H, W), float("nan"), device=DEV)
    am = torch.full((N, H, W), 255, dtype=torch.uint8, device=DEV)
    lo = torch.zeros(16, device=DEV)
    dy = torch.full((N, H, W, 64), float("nan"), dtype=dtype, device=DEV)
    dw = torch.full((C, 64), float("nan"), device=DEV)
    db = torch.full((C,), float("nan"), device=DEV)
    ops.head_loss(yd, w.to(DEV), b.to(DEV), loss=kind, labels=labels.to(DEV), logits=lg, argmax=am, loss_out=lo, dy=dy, dw=dw, db=db)
    assert_close(lg, logits.detach(), f"head logits {loss} {dtype}", rtol=1e-5, atol=1e-5)
    assert abs(lo[0].item() - L.item()) < 1e-5 * max(1.0, abs(L.item())), (lo[0].item(), L.item())
    want_am = logits.detach().argmax(1) if C > 1 else (logits.detach()[:, 0] > 0).long()
    near = torch.zeros(N, H, W, dtype=torch.bool)
    if C > 1:
        top2 = logits.detach().topk(2, dim=1).values
        near = (top2[:, 0] - top2[:, 1]) < 1e-5
    else:
        near = logits.detach()[:, 0].abs() < 1e-5
    assert torch.equal(am.cpu().long()[~near], want_am[~near]), "argmax"
    # the argmax kernel itself is exact: bit-identical to argmax of ITS OWN logits
    own = lg.cpu().argmax(1) if C > 1 else (lg.cpu()[:, 0] > 0).long()
    assert torch.equal(am.cpu().long(), own), "argmax not bit-exact on its own logits"
    gt = dict(rtol=1e-4, atol=1e-7) if dtype == torch.float32 else dict(rtol=2e-2, atol=1e-5)
    assert_close(from_nhwc(dy), yq.grad * (q(y, dtype) > 0), f"head dy {loss} {dtype}", **gt)
    assert_close(dw, wq.grad, f"head dw {loss}", rtol=1e-4, atol=1e-6)
    assert_close(db, bq.grad, f"head db {loss}", rtol=1e-4, atol=1e-6)


def test_adamw_clip_matches_oracle():
    ops = _ops()
    from oracle import unet2d_oracle as o2
    n = 100_003
    p = {"a.weight": rnd(n, seed=80), "a.bias": rnd(1000, seed=81)}
    opt = o2.AdamW({k: v.clone() for k, v in p.items()})
    pd = {k: v.clone() for k, v in p.items()}
    P = torch.cat([p["a.weight"], torch.zeros(61), p["a.bias"]]).to(DEV)   # padded like FlatParams
    M, V = torch.zeros_like(P), torch.zeros_like(P)
    nd = n + 61
    partials = torch.zeros(ops.sumsq_npartials(P.numel()), device=DEV)
    gn = torch.zeros(1, device=DEV)
    for step in range(1, 4):
        grads = {"a.weight": rnd(n, seed=82 + step, scale=0.01 * step), "a.bias": rnd(1000, seed=90 + step, scale=0.01)}
        total, clipped = o2.clip_grad_norm(grads, 1.0)
        opt.step(pd, clipped)
        G = torch.cat([grads["a.weight"], torch.zeros(61), grads["a.bias"]]).to(DEV)
        ops.sumsq(G, partials)
        kw = dict(partials=partials, max_norm=1.0, lr=5e-3, beta1=0.9, beta2=0.999, eps=1e-8, step=step)
        ops.adamw_step(P[:nd], G[:nd], M[:nd], V[:nd], weight_decay=1e-3, gradnorm_out=gn, **kw)
        ops.adamw_step(P[nd:], G[nd:], M[nd:], V[nd:], weight_decay=0.0, **kw)
        assert abs(gn.item() - total.item()) < 1e-5 * total.item()
        assert_close(P[:n], pd["a.weight"], f"adamw weights step {step}", rtol=1e-5, atol=1e-6)
        assert_close(P[nd:], pd["a.bias"], f"adamw biases step {step}", rtol=1e-5, atol=1e-6)


def test_errors_are_loud():
    ops = _ops()
    from mdeical_image_segmentation_amd._lib import MisError
    x = torch.zeros(1, 8, 8, 48, device=DEV)          # Cin not a multiple of the K chunk
    w = torch.zeros(9, 64, 48, device=DEV)
    y = torch.zeros(1, 8, 8, 64, device=DEV)
    with pytest.raises(MisError):
        ops.conv_igemm(x, w, y, ksize=3, Cin=48, Cout=64)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_batched_weight_repack_equals_per_layer(dtype, monkeypatch):
    """mis_pack_batch (every layer's operand repack in one launch) against the per-layer mis_pack_conv_weight / mis_pack_convt_weight calls: bit-equal operands"""
    from mdeical_image_segmentation_amd.engine2d import UNet2DEngine
    eng = UNet2DEngine(3, 4, dtype=dtype, device=DEV, seed=1)
    monkeypatch.setenv("MISAMD_REPACK_PER_LAYER", "1")
    eng.repack()
    torch.cuda.synchronize()
    want = {k: (eng.wf[k].clone(), eng.wd_[k].clone()) for k in eng.wf}
    for k in eng.wf:
        eng.wf[k].zero_()
        eng.wd_[k].zero_()
    monkeypatch.delenv("MISAMD_REPACK_PER_LAYER")
    eng.repack()
    torch.cuda.synchronize()
    assert eng._pack_table.n == len(eng.wf) >= 20
    for k, (wf, wd) in want.items():
        assert torch.equal(eng.wf[k], wf) and torch.equal(eng.wd_[k], wd), k
