"""Per-DISPATCH-BRANCH parity of the two MFMA kernels that carry the train step: every instantiation that
`mis_conv_igemm` / `mis_wgrad` can select is driven by at least one case here, on ragged grids, with the epilogue / operand variants
the engines use (bias, ReLU, ReLU mask, two destinations + pixel-unshuffle, pixel-shuffle, output channel slice), and the test ASSERTS
which configuration ran (`mis_conv_last_dispatch`, `mis_wgrad_last_dispatch`) so a case cannot silently drift to another branch.

Reference arithmetic: reference model/unet2d/layers.py:122-126 (Conv2d k3 p1 + ReLU), :165 (ConvTranspose2d k2 s2) = ATen conv on the
bf16-rounded (or fp32) operands with fp32 accumulation; tolerances = test_gpu_kernels.tol()."""
import pytest
import torch
import torch.nn.functional as F

from test_gpu_kernels import DEV, _ops, assert_close, from_nhwc, q, rnd, to_nhwc, tol

pytestmark = pytest.mark.gpu

BF, F32 = torch.bfloat16, torch.float32

# (dtype, N, H, W, Cin, Cout, expected configuration)
K3_CASES = [
    # ping-pong kernel conv_pp_kernel<8 / 4> (16 x 16-pixel tiles): bf16, Cout % 128 == 0, grids the 32-row tiles of conv_ppc_kernel do not fit (or MIS_CONV_NOPPC=1)
    (BF, 2, 20, 36, 256, 256, "k3.2d.pp256", ""),        # ragged tiles, 4 K chunks
    (BF, 1, 9, 17, 512, 256, "k3.2d.pp256", ""),
    (BF, 1, 9, 17, 1024, 1024, "k3.2d.pp256", ""),       # middle_conv.second: 16 K chunks x 4 column tiles
    (BF, 2, 100, 150, 128, 256, "k3.2d.pp256", ""),      # 140 spatial tiles
    (BF, 3, 150, 170, 64, 256, "k3.2d.pp256", "MIS_CONV_NOPPC"),       # 330 tiles > 256 persistent blocks: several tiles per block, one K chunk per tile
    (BF, 1, 20, 36, 64, 128, "k3.2d.pp128", ""),
    (BF, 2, 20, 36, 256, 128, "k3.2d.pp128", ""),
    (BF, 3, 150, 170, 128, 128, "k3.2d.pp128", "MIS_CONV_NOPPC"),      # 330 tiles > 256 blocks, 2 K chunks
    (BF, 1, 16, 16, 64, 128, "k3.2d.pp128", ""),         # a single tile
    # column-segment ping-pong kernel (conv_ppc_kernel<8>: 32 x 16-pixel tiles, 32-channel K chunks, a filter column = 96 MFMAs per segment): the default for
    # Cout % 128 == 0 when its 32-row tiles fit the grid (MIS_CONV_PPC=1 takes it regardless)
    (BF, 3, 150, 170, 64, 256, "k3.2d.ppc8", ""),                    # default dispatch: 3 x 5 x 11 tiles x 2 column tiles = 330 > 256 persistent blocks
    (BF, 5, 150, 170, 128, 128, "k3.2d.ppc8", ""),                   # 275 tiles, 4 K chunks
    (BF, 2, 64, 64, 128, 128, "k3.2d.ppc8", ""),                     # interior tiles
    (BF, 1, 20, 36, 64, 128, "k3.2d.ppc8", "MIS_CONV_PPC"),          # ragged in both directions
    (BF, 2, 20, 36, 256, 256, "k3.2d.ppc8", "MIS_CONV_PPC"),         # two column tiles, 8 K chunks
    (BF, 1, 9, 17, 1024, 1024, "k3.2d.ppc8", "MIS_CONV_PPC"),        # 32 K chunks x 8 column tiles
    (BF, 1, 32, 16, 64, 128, "k3.2d.ppc8", "MIS_CONV_PPC"),          # a single tile, exactly: the whole run is prologue + six segments
    # the configurations behind it (MIS_CONV_NOPP=1): still reachable, still tested
    (BF, 2, 20, 36, 256, 256, "k3.2d.bn256", "MIS_CONV_NOPP"),
    (BF, 1, 9, 17, 1024, 1024, "k3.2d.bn256", "MIS_CONV_NOPP"),
    (BF, 2, 100, 150, 128, 256, "k3.2d.bn128.persist", "MIS_CONV_NOPP"),   # persistent tiles, 280 tiles > 256 blocks
    (BF, 1, 20, 36, 64, 128, "k3.2d.bn128.persist", "MIS_CONV_NOPP"),
    (BF, 2, 20, 36, 256, 128, "k3.2d.bn128.dma", "MIS_CONV_NOPP"),
    # 64-column layers: the weight-stationary / bn64 configurations; the ping-pong kernel with 64-column blocks is selectable (MIS_CONV_PP64=1; slower)
    (BF, 1, 70, 90, 128, 64, "k3.2d.bn64.persist", ""),  # 32x16-pixel persistent tiles
    (BF, 2, 20, 36, 128, 64, "k3.2d.bn64.v1", ""),       # small grid: 64-column 4-wave config
    (BF, 2, 256, 256, 64, 64, "k3.2d.ws64", ""),         # weight-stationary (LDS) kernel
    (BF, 2, 256, 256, 64, 64, "k3.2d.rs64", "MIS_CONV_RS64"),          # 64 -> 64: register-stationary ping-pong kernel (filter in VGPRs; opt-in, slower)
    (BF, 3, 150, 170, 64, 64, "k3.2d.rs64", "MIS_CONV_RS64"),          # ragged, 330 tiles > 256 blocks
    (BF, 1, 20, 36, 64, 64, "k3.2d.rs64", "MIS_CONV_RS64"),
    (BF, 1, 70, 90, 128, 64, "k3.2d.pp64", "MIS_CONV_PP64"),
    (BF, 2, 20, 36, 64, 64, "k3.2d.pp64", "MIS_CONV_PP64"),
    (BF, 3, 150, 170, 64, 64, "k3.2d.pp64", "MIS_CONV_PP64"),          # 330 tiles > 256 blocks: one K chunk per tile, tile loop taken
    (BF, 1, 70, 90, 128, 64, "k3.2d.ppc8n2", "MIS_CONV_PPC64"),        # 64-column blocks of the column-segment kernel (opt-in)
    (BF, 2, 20, 36, 64, 64, "k3.2d.ppc8n2", "MIS_CONV_PPC64"),
    (BF, 3, 150, 170, 64, 64, "k3.2d.ppc8n2", "MIS_CONV_PPC64"),       # 165 tiles
    (BF, 9, 150, 170, 64, 64, "k3.2d.ppc8n2", "MIS_CONV_PPC64"),       # 495 tiles > 256 persistent blocks
    (BF, 1, 40, 40, 64, 192, "k3.2d.ppc8n2", "MIS_CONV_PPC64"),        # three column tiles
    (F32, 2, 20, 36, 64, 128, "k3.2d.bn128.persist", ""),
    (F32, 1, 9, 17, 128, 256, "k3.2d.bn128.dma", ""),
    (F32, 1, 70, 90, 64, 64, "k3.2d.bn64.persist", ""),
    (F32, 2, 9, 17, 32, 64, "k3.2d.bn64.v1", ""),
]


def _conv_ref(x, w, b, dtype):
    return F.conv2d(q(x, dtype), q(w, dtype), b, padding=w.shape[-1] // 2)


@pytest.mark.parametrize("case", K3_CASES, ids=lambda c: f"{'bf16' if c[0] == BF else 'f32'}-{c[1]}x{c[2]}x{c[3]}-{c[4]}to{c[5]}-{c[6]}")
def test_conv3x3_every_branch(case, monkeypatch):
    ops = _ops()
    dtype, N, H, W, Cin, Cout, want_cfg, env = case
    if env:
        monkeypatch.setenv(*(env.split("=") if "=" in env else (env, "1")))
    x = rnd(N, Cin, H, W, seed=110)
    w = rnd(Cout, Cin, 3, 3, seed=111, scale=(9 * Cin) ** -0.5)
    b = rnd(Cout, seed=112)
    m = rnd(N, Cout, H, W, seed=113)
    xd = to_nhwc(x, dtype)
    wf = torch.empty(9, Cout, Cin, dtype=dtype, device=DEV)
    ops.pack_conv_weight(w.to(DEV), wf, None)
    ref = _conv_ref(x, w, None, dtype)
    t = tol(dtype, 9 * Cin)
    # (a) forward form: bias + ReLU, written into a channel slice of a wider buffer
    ybuf = torch.full((N, H, W, Cout + 64), float("nan"), dtype=dtype, device=DEV)
    ops.conv_igemm(xd, wf, ops.View(ybuf, 64, Cout), ksize=3, Cin=Cin, Cout=Cout, bias=b.to(DEV), relu=True)
    cfg = ops.conv_last_dispatch()
    assert cfg.startswith(want_cfg), f"case meant for {want_cfg} ran {cfg}"
    assert_close(from_nhwc(ybuf[..., 64:].contiguous()), F.relu(ref + b.view(1, -1, 1, 1)), f"fwd {cfg}", **t)
    assert torch.isnan(ybuf[..., :64].float()).all(), "wrote outside the channel slice"
    # (b) dgrad form: no bias, ReLU mask of the layer input in the epilogue
    y2 = torch.full((N, H, W, Cout), float("nan"), dtype=dtype, device=DEV)
    ops.conv_igemm(xd, wf, y2, ksize=3, Cin=Cin, Cout=Cout, mask=to_nhwc(m, dtype))
    assert ops.conv_last_dispatch() == cfg
    assert_close(from_nhwc(y2), ref * (q(m, dtype) > 0), f"mask {cfg}", **t)
    # (c) two destinations (dgrad of up_conv.*.first): first half pixel-unshuffled, second half plain
    if Cout % 128 == 0 and (Cout // 2) % (128 if ("bn256" in cfg or "pp256" in cfg) else 64) == 0 and H % 2 == 0 and W % 2 == 0:
        h = Cout // 2
        d0 = torch.full((N, H // 2, W // 2, 4 * h), float("nan"), dtype=dtype, device=DEV)
        d1 = torch.full((N, H, W, h), float("nan"), dtype=dtype, device=DEV)
        ops.conv_igemm(xd, wf, d0, ksize=3, Cin=Cin, Cout=Cout, y0_mode=ops.OUT_UNSHUFFLE2, y1=d1, Cout0=h)
        assert ops.conv_last_dispatch() == cfg
        assert_close(from_nhwc(d1), ref[:, h:], f"split plain half {cfg}", **t)
        uns = ref[:, :h].reshape(N, h, H // 2, 2, W // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(N, 4 * h, H // 2, W // 2)
        assert_close(from_nhwc(d0), uns, f"split unshuffled half {cfg}", **t)


K1_CASES = [
    # (dtype, N, H, W, Cin, Cout, expected) - Cout = 4*Cq for the transposed-conv forward, = Cin_of_layer for its dgrad
    (BF, 2, 10, 18, 256, 512, "k1.2d.bn256"),
    (BF, 1, 9, 17, 2048, 1024, "k1.2d.bn256"),            # up_sample.0 dgrad: 32 K chunks
    (BF, 2, 100, 150, 128, 256, "k1.2d.bn128.persist"),   # 280 tiles > 256 blocks
    (BF, 2, 10, 18, 512, 128, "k1.2d.bn128.dma"),
    (BF, 2, 10, 18, 128, 64, "k1.2d.bn64"),
    (F32, 2, 10, 18, 64, 256, "k1.2d.bn128.persist"),
    (F32, 2, 10, 18, 256, 128, "k1.2d.bn128.dma"),
    (F32, 2, 10, 18, 64, 64, "k1.2d.bn64"),
]


@pytest.mark.parametrize("case", K1_CASES, ids=lambda c: f"{'bf16' if c[0] == BF else 'f32'}-{c[1]}x{c[2]}x{c[3]}-{c[4]}to{c[5]}")
def test_gemm1x1_every_branch(case):
    """the two 1x1 GEMMs of ConvTranspose2d(k2, s2): forward (bias + pixel-shuffle store into a concat slice) and dgrad (ReLU mask)"""
    ops = _ops()
    dtype, N, H, W, Cin, Cout, want_cfg = case
    x = rnd(N, Cin, H, W, seed=120)
    w = rnd(Cout, Cin, 1, 1, seed=121, scale=Cin ** -0.5)
    xd = to_nhwc(x, dtype)
    wf = q(w.view(Cout, Cin), dtype).to(dtype).to(DEV).contiguous()       # [column][Cin] = the packed forward operand of a 1x1 GEMM
    ref = _conv_ref(x, w, None, dtype)
    t = tol(dtype, Cin)
    m = rnd(N, Cout, H, W, seed=123)
    y = torch.full((N, H, W, Cout), float("nan"), dtype=dtype, device=DEV)
    ops.conv_igemm(xd, wf, y, ksize=1, Cin=Cin, Cout=Cout, mask=to_nhwc(m, dtype))
    cfg = ops.conv_last_dispatch()
    assert cfg.startswith(want_cfg), f"case meant for {want_cfg} ran {cfg}"
    assert_close(from_nhwc(y), ref * (q(m, dtype) > 0), f"1x1 mask {cfg}", **t)
    if Cout % 256 == 0:
        cq = Cout // 4
        b = rnd(cq, seed=122)
        cat = torch.full((N, 2 * H, 2 * W, 2 * cq), float("nan"), dtype=dtype, device=DEV)
        ops.conv_igemm(xd, wf, ops.View(cat, 0, cq), ksize=1, Cin=Cin, Cout=Cout, bias=b.to(DEV), y0_mode=ops.OUT_SHUFFLE2)
        assert ops.conv_last_dispatch() == cfg
        # column ab*Cq + c of pixel (h, w) -> pixel (2h + a, 2w + b'), channel c
        sh = (ref.view(N, 2, 2, cq, H, W) + b.view(1, 1, 1, cq, 1, 1)).permute(0, 3, 4, 1, 5, 2).reshape(N, cq, 2 * H, 2 * W)
        assert_close(from_nhwc(cat[..., :cq].contiguous()), sh, f"1x1 shuffle {cfg}", **t)
        assert torch.isnan(cat[..., cq:].float()).all(), "wrote outside the concat slice"


WG_CASES = [
    # (dtype, ksize, N, H, W, Cin, Cout, expected configuration, environment switch)
    (BF, 3, 2, 20, 36, 64, 64, "k3.2d.pps", ""),        # 64-column tiles: pixel-split wide wave tiles, ragged tiles, two slabs per block
    (BF, 3, 2, 20, 36, 64, 64, "k3.2d.pp", "MIS_WGRAD_PP_NOWIDE"),
    (BF, 3, 1, 9, 17, 256, 256, "k3.2d.ppw", ""),        # a single (ragged) pixel tile per block
    (BF, 3, 1, 9, 17, 1024, 1024, "k3.2d.ppw", ""),      # 256 channel-tile pairs
    (BF, 3, 2, 40, 40, 512, 256, "k3.2d.ppw", ""),
    (BF, 3, 2, 40, 40, 512, 256, "k3.2d.pp", "MIS_WGRAD_PP_NOWIDE"),   # the 64 x 64 ping-pong kernel on a multi-pair shape
    (BF, 3, 3, 50, 70, 64, 128, "k3.2d.ppw", ""),        # 60 pixel tiles over 128 blocks... several tiles per block for 2 pairs
    (BF, 3, 2, 150, 170, 64, 64, "k3.2d.pps", ""),       # 220 tiles, 1 pair: persistent blocks with the tile loop taken
    (BF, 3, 1, 40, 40, 128, 64, "k3.2d.pps", ""),        # two input-channel tiles: only ci tile 0 writes the bias column sums
    (BF, 3, 2, 150, 170, 64, 128, "k3.2d.ppw", ""),      # wide kernel, persistent blocks with the tile loop taken, ragged 8-row tiles
    (BF, 3, 2, 20, 36, 64, 64, "k3.2d.tr", "MIS_WGRAD_NOPP"),     # the kernel behind it
    (BF, 3, 1, 9, 17, 1024, 1024, "k3.2d.tr", "MIS_WGRAD_NOPP"),
    (BF, 3, 2, 40, 40, 512, 256, "k3.2d.tr", "MIS_WGRAD_NOPP"),
    (BF, 1, 2, 10, 18, 256, 512, "k1.2d.wide", ""),   # 128 x 128 channel tiles
    (BF, 1, 1, 9, 17, 2048, 1024, "k1.2d.wide", ""),
    (BF, 1, 2, 10, 18, 64, 256, "k1.2d.tr", ""),      # Cin not a multiple of 128: 64 x 64 tiles
    (F32, 3, 2, 20, 36, 32, 64, "k3.2d", ""),
    (F32, 3, 1, 9, 17, 256, 128, "k3.2d", ""),
    (F32, 1, 2, 10, 18, 128, 256, "k1.2d", ""),
]


@pytest.mark.parametrize("case", WG_CASES, ids=lambda c: f"{'bf16' if c[0] == BF else 'f32'}-k{c[1]}-{c[2]}x{c[3]}x{c[4]}-{c[5]}to{c[6]}-{c[8]}")
def test_wgrad_every_branch(case, monkeypatch):
    ops = _ops()
    dtype, ks, N, H, W, Cin, Cout, want_cfg, env = case
    if env:
        monkeypatch.setenv(env, "1")
    x = rnd(N, Cin, H, W, seed=130)
    dy = rnd(N, Cout, H, W, seed=131)
    wq = torch.zeros(Cout, Cin, ks, ks, requires_grad=True)
    F.conv2d(q(x, dtype), wq, None, padding=ks // 2).backward(q(dy, dtype))
    k = N * H * W
    wt = dict(rtol=1e-4, atol=1e-4 * k ** 0.5)
    dbref = q(dy, dtype).sum((0, 2, 3))
    if ks == 3:
        dw = torch.full((Cout, Cin, 3, 3), float("nan"), device=DEV)
        dbf = torch.full((Cout,), float("nan"), device=DEV)
        ops.wgrad(to_nhwc(x, dtype), to_nhwc(dy, dtype), dw, ksize=3, Cin=Cin, Cout=Cout, dbias=dbf)
        cfg, nsplit = ops.wgrad_last_dispatch()
        assert cfg == want_cfg or (cfg.startswith(want_cfg) and not want_cfg.endswith(".pp")), f"case meant for {want_cfg} ran {cfg}"
        assert_close(dw, wq.grad, f"wgrad {cfg} nsplit={nsplit}", **wt)
        assert_close(dbf, dbref, f"fused bias grad {cfg}", rtol=1e-4, atol=1e-3)
    else:
        # transposed-conv layout: GEMM column ab*Cq + c -> dw[ci][c][a][b], bias folded over the 4 (a, b) groups
        cq = Cout // 4
        dw = torch.full((Cin, cq, 2, 2), float("nan"), device=DEV)
        dbf = torch.full((cq,), float("nan"), device=DEV)
        ops.wgrad(to_nhwc(x, dtype), to_nhwc(dy, dtype), dw, ksize=1, Cin=Cin, Cout=Cout, dw_layout=1, dbias=dbf)
        cfg, nsplit = ops.wgrad_last_dispatch()
        assert cfg.startswith(want_cfg), f"case meant for {want_cfg} ran {cfg}"
        want = wq.grad.view(2, 2, cq, Cin).permute(3, 2, 0, 1)
        assert_close(dw, want, f"wgrad {cfg} nsplit={nsplit}", **wt)
        assert_close(dbf, dbref.view(4, cq).sum(0), f"fused bias grad {cfg}", rtol=1e-4, atol=2e-3)
    # plain (non-accumulating) semantics: a second call overwrites, alpha scales
    ops.wgrad(to_nhwc(x, dtype), to_nhwc(dy, dtype), dw, ksize=ks, Cin=Cin, Cout=Cout, dw_layout=0 if ks == 3 else 1, alpha=0.5)
    want2 = wq.grad if ks == 3 else wq.grad.view(2, 2, Cout // 4, Cin).permute(3, 2, 0, 1)
    assert_close(dw, 0.5 * want2, "alpha", rtol=1e-4, atol=1e-4 * k ** 0.5)
