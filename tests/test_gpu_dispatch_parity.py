"""Per-DISPATCH-BRANCH parity of the two MFMA kernels that carry the train step: every instantiation that
`mis_conv_igemm` / `mis_wgrad` can select is driven by at least one case here, on ragged grids, with the epilogue / operand variants
the engines use (bias, ReLU, ReLU mask, two destinations + pixel-unshuffle, pixel-shuffle, output channel slice), and the test ASSERTS
which configuration ran (`mis_conv_last_dispatch`, `mis_wgrad_last_dispatch`) so a case cannot silently drift to another branch.  Non-default configurations are reached through the library's override entry point
(`mis_dispatch_override`, the `switches` fixture), not through the environment.

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
    (BF, 2, 256, 256, 64, 64, "k3.2d.ws64", "MIS_CONV_NOPPD"),         # weight-stationary (LDS) kernel (the default before the deep-prefetch kernel)
    (BF, 2, 256, 256, 64, 64, "k3.2d.rs64", "MIS_CONV_RS64"),          # 64 -> 64: register-stationary ping-pong kernel (filter in VGPRs; opt-in, slower)
    (BF, 3, 150, 170, 64, 64, "k3.2d.rs64", "MIS_CONV_RS64"),          # ragged, 330 tiles > 256 blocks
    (BF, 1, 20, 36, 64, 64, "k3.2d.rs64", "MIS_CONV_RS64"),
    (BF, 1, 70, 90, 128, 64, "k3.2d.pp64", "MIS_CONV_PP64"),
    (BF, 2, 20, 36, 64, 64, "k3.2d.pp64", "MIS_CONV_PP64"),
    (BF, 3, 150, 170, 64, 64, "k3.2d.pp64", "MIS_CONV_PP64"),          # 330 tiles > 256 blocks: one K chunk per tile, tile loop taken
    # 64-column blocks, deep-prefetch form (conv_ppd.hip): the default for Cout % 128 != 0 on grids that fill 32-row tiles
    (BF, 2, 64, 50, 128, 64, "k3.2d.ppd8", ""),                        # four K chunks, ragged in W
    (BF, 2, 256, 256, 64, 64, "k3.2d.ppd8", ""),                       # two K chunks: every halo prefetch crosses a tile boundary
    (BF, 3, 150, 170, 64, 64, "k3.2d.ppd8", ""),                       # ragged both ways, 165 tiles
    (BF, 9, 150, 170, 64, 64, "k3.2d.ppd8", ""),                       # 495 tiles > 256 persistent blocks
    (BF, 1, 70, 90, 128, 64, "k3.2d.ppd8", "MIS_CONV_PPC64"),          # ... and wherever eligible with the switch
    (BF, 2, 20, 36, 64, 64, "k3.2d.ppd8", "MIS_CONV_PPC64"),           # a single ragged tile per block
    (BF, 1, 40, 40, 64, 192, "k3.2d.ppd8", "MIS_CONV_PPC64"),          # three column tiles
    (BF, 1, 33, 17, 256, 64, "k3.2d.ppd8", "MIS_CONV_PPC64"),          # eight K chunks, 2 x 2 ragged tiles
    # ... and the kernel it replaces (same tiles, one prefetch stream per wave: conv_ppc_kernel<8, 2>)
    (BF, 2, 64, 50, 128, 64, "k3.2d.ppc8n2", "MIS_CONV_NOPPD"),
    (BF, 1, 70, 90, 128, 64, "k3.2d.ppc8n2", "MIS_CONV_PPC64,MIS_CONV_NOPPD"),
    (BF, 2, 20, 36, 64, 64, "k3.2d.ppc8n2", "MIS_CONV_PPC64,MIS_CONV_NOPPD"),
    (BF, 3, 150, 170, 64, 64, "k3.2d.ppc8n2", "MIS_CONV_PPC64,MIS_CONV_NOPPD"),       # 165 tiles
    (BF, 9, 150, 170, 64, 64, "k3.2d.ppc8n2", "MIS_CONV_PPC64,MIS_CONV_NOPPD"),       # 495 tiles > 256 persistent blocks
    (BF, 1, 40, 40, 64, 192, "k3.2d.ppc8n2", "MIS_CONV_PPC64,MIS_CONV_NOPPD"),        # three column tiles
    (F32, 2, 20, 36, 64, 128, "k3.2d.bn128.persist", ""),
    (F32, 1, 9, 17, 128, 256, "k3.2d.bn128.dma", ""),
    (F32, 1, 70, 90, 64, 64, "k3.2d.bn64.persist", ""),
    (F32, 2, 9, 17, 32, 64, "k3.2d.bn64.v1", ""),
]


def _decode_relu_bits(bits, N, H, W, C):
    """the ReLU-bits layout of csrc/relu_bits.hpp back to a bool (N, H, W, C) tensor: records [4][2][8 rows] bytes per (n, block of 8 rows, x, block of 64 channels);
    channel group g = (c >> 3) & 7 sits at [g & 3][g >> 2], bit c & 7"""
    H8 = (H + 7) // 8
    r = bits.view(N, H8, W, C // 64, 4, 2, 8)                                  # [u16idx][byteidx][row]
    r = r.permute(0, 1, 6, 2, 3, 5, 4).reshape(N, H8 * 8, W, C // 64, 8)         # (n, y, x, cb, g = byteidx*4 + u16idx)
    b = (r.unsqueeze(-1) >> torch.arange(8, device=bits.device, dtype=torch.uint8)) & 1
    return b.reshape(N, H8 * 8, W, C).bool()[:, :H]


def _bits_of(ops, t_nhwc):
    N, H, W, C = t_nhwc.shape
    bits = torch.zeros(ops.relu_bits_bytes(N, H, W, C), dtype=torch.uint8, device=DEV)
    ops.relu_bits(t_nhwc, bits)
    return bits


@pytest.mark.parametrize("shape", [(2, 8, 16, 64), (1, 13, 7, 128), (3, 33, 20, 192)])
def test_relu_bits_layout(shape):
    """mis_relu_bits against the layout's definition, signed zeros and ragged row blocks included"""
    ops = _ops()
    N, H, W, C = shape
    y = torch.randn(N, H, W, C, generator=torch.Generator().manual_seed(7)).clamp_(min=0)
    y[0, 0, 0, :8] = torch.tensor([0.0, -0.0, 1e-30, -1.0, 2.0, 0.0, 3.0, -0.0])
    yd = y.to(torch.bfloat16).to(DEV)
    assert ops.relu_bits_bytes(N, H, W, C) == N * ((H + 7) // 8) * W * C
    got = _decode_relu_bits(_bits_of(ops, yd), N, H, W, C)
    assert torch.equal(got, yd.float() > 0)


def _same_kernel(a, b):
    """dispatch tags name the instantiation; the column-segment kernels have one per epilogue mask path (suffix .mask / .bits) - the same kernel otherwise"""
    strip = lambda t: t.replace(".mask", "").replace(".bits", "")
    return strip(a) == strip(b)


def _conv_ref(x, w, b, dtype):
    return F.conv2d(q(x, dtype), q(w, dtype), b, padding=w.shape[-1] // 2)


@pytest.mark.parametrize("case", K3_CASES, ids=lambda c: f"{'bf16' if c[0] == BF else 'f32'}-{c[1]}x{c[2]}x{c[3]}-{c[4]}to{c[5]}-{c[6]}")
def test_conv3x3_every_branch(case, switches):
    ops = _ops()
    dtype, N, H, W, Cin, Cout, want_cfg, env = case
    for e in filter(None, env.split(",")):
        switches(*((e.split("=")[0], int(e.split("=")[1])) if "=" in e else (e, 1)))
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
    assert _same_kernel(ops.conv_last_dispatch(), cfg)
    assert_close(from_nhwc(y2), ref * (q(m, dtype) > 0), f"mask {cfg}", **t)
    if dtype == BF:
        # (b') the same mask as ReLU BITS: bit-for-bit the masked result (the kernels that cannot read bits hand the call to one that can)
        y3 = torch.full((N, H, W, Cout), float("nan"), dtype=dtype, device=DEV)
        ops.conv_igemm(xd, wf, y3, ksize=3, Cin=Cin, Cout=Cout, mask_bits=_bits_of(ops, to_nhwc(m, dtype)))
        cfg_bits = ops.conv_last_dispatch()
        assert_close(from_nhwc(y3), ref * (q(m, dtype) > 0), f"mask_bits {cfg_bits}", **t)
        if _same_kernel(cfg_bits, cfg):          # same kernel, same sums
            assert torch.equal(y3.view(torch.int16), y2.view(torch.int16)), f"mask_bits differs from the bf16 mask ({cfg})"
        else:                        # the pre-column-segment ping-pong kernels do not read bits: the dispatcher must have chosen another kernel, not dropped the mask
            assert "pp" in cfg or "rs64" in cfg, f"{cfg} handed a mask_bits call to {cfg_bits}"
        # (a') the forward form also writes the ReLU bits of its output
        y4 = torch.full((N, H, W, Cout), float("nan"), dtype=dtype, device=DEV)
        rb = torch.full((ops.relu_bits_bytes(N, H, W, Cout),), 0xA5, dtype=torch.uint8, device=DEV)
        ops.conv_igemm(xd, wf, y4, ksize=3, Cin=Cin, Cout=Cout, bias=b.to(DEV), relu=True, relu_bits=rb)
        assert _same_kernel(ops.conv_last_dispatch(), cfg)
        assert torch.equal(y4.view(torch.int16), ybuf[..., 64:].contiguous().view(torch.int16))
        assert torch.equal(_decode_relu_bits(rb, N, H, W, Cout), y4.float() > 0), f"relu_bits {cfg}"
    # (c) two destinations (dgrad of up_conv.*.first): first half pixel-unshuffled, second half plain
    if Cout % 128 == 0 and (Cout // 2) % (128 if ("bn256" in cfg or "pp256" in cfg) else 64) == 0 and H % 2 == 0 and W % 2 == 0:
        h = Cout // 2
        d0 = torch.full((N, H // 2, W // 2, 4 * h), float("nan"), dtype=dtype, device=DEV)
        d1 = torch.full((N, H, W, h), float("nan"), dtype=dtype, device=DEV)
        ops.conv_igemm(xd, wf, d0, ksize=3, Cin=Cin, Cout=Cout, y0_mode=ops.OUT_UNSHUFFLE2, y1=d1, Cout0=h)
        assert _same_kernel(ops.conv_last_dispatch(), cfg)
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
    # the ping-pong 1x1 kernel (32-row tiles must fit the grid): ragged in W / in both, two K steps, 64 K steps, more tiles than blocks
    (BF, 2, 64, 50, 128, 256, "k1.2d.pp"),
    (BF, 2, 60, 70, 256, 512, "k1.2d.pp"),
    (BF, 3, 96, 40, 64, 128, "k1.2d.pp"),
    (BF, 1, 32, 16, 2048, 1024, "k1.2d.pp"),
    (BF, 4, 128, 176, 64, 256, "k1.2d.pp"),
    (BF, 2, 64, 50, 128, 256, "k1.2d.bn128.persist", "MIS_GEMM1_NOPP"),     # ... and the kernel behind it
    (F32, 2, 10, 18, 64, 256, "k1.2d.bn128.persist"),
    (F32, 2, 10, 18, 256, 128, "k1.2d.bn128.dma"),
    (F32, 2, 10, 18, 64, 64, "k1.2d.bn64"),
]


@pytest.mark.parametrize("case", K1_CASES, ids=lambda c: f"{'bf16' if c[0] == BF else 'f32'}-{c[1]}x{c[2]}x{c[3]}-{c[4]}to{c[5]}{'-' + c[7] if len(c) > 7 else ''}")
def test_gemm1x1_every_branch(case, switches):
    """the two 1x1 GEMMs of ConvTranspose2d(k2, s2): forward (bias + pixel-shuffle store into a concat slice) and dgrad (ReLU mask)"""
    ops = _ops()
    dtype, N, H, W, Cin, Cout, want_cfg = case[:7]
    if len(case) > 7:
        switches(case[7], 1)
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
    if dtype == BF:
        yb = torch.full((N, H, W, Cout), float("nan"), dtype=dtype, device=DEV)
        ops.conv_igemm(xd, wf, yb, ksize=1, Cin=Cin, Cout=Cout, mask_bits=_bits_of(ops, to_nhwc(m, dtype)))
        assert _same_kernel(ops.conv_last_dispatch(), cfg)
        assert torch.equal(yb.view(torch.int16), y.view(torch.int16)), f"1x1 mask_bits differs from the bf16 mask ({cfg})"
    if Cout % 256 == 0:
        cq = Cout // 4
        b = rnd(cq, seed=122)
        cat = torch.full((N, 2 * H, 2 * W, 2 * cq), float("nan"), dtype=dtype, device=DEV)
        ops.conv_igemm(xd, wf, ops.View(cat, 0, cq), ksize=1, Cin=Cin, Cout=Cout, bias=b.to(DEV), y0_mode=ops.OUT_SHUFFLE2)
        assert _same_kernel(ops.conv_last_dispatch(), cfg)
        # column ab*Cq + c of pixel (h, w) -> pixel (2h + a, 2w + b'), channel c
        sh = (ref.view(N, 2, 2, cq, H, W) + b.view(1, 1, 1, cq, 1, 1)).permute(0, 3, 4, 1, 5, 2).reshape(N, cq, 2 * H, 2 * W)
        assert_close(from_nhwc(cat[..., :cq].contiguous()), sh, f"1x1 shuffle {cfg}", **t)
        assert torch.isnan(cat[..., cq:].float()).all(), "wrote outside the concat slice"


WG_CASES = [
    # (dtype, ksize, N, H, W, Cin, Cout, expected configuration, dispatch switch)
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
    # row variants (32-pixel-wide tiles, x fragments reused across the k-steps of a tile): the default wherever 32-wide tiles fit the grid
    (BF, 3, 2, 64, 64, 64, 128, "k3.2d.ppwr", "MIS_WGRAD_PP_NOSTREAM"),
    (BF, 3, 1, 9, 32, 256, 256, "k3.2d.ppwr", ""),       # ragged 4-row tiles, 8 channel-tile pairs
    (BF, 3, 1, 32, 32, 1024, 1024, "k3.2d.ppwr", "MIS_WGRAD_PP_NOSTREAM"),    # 128 pairs x 2 splits
    # streaming form of the row kernel (H % 4 == 0, W % 32 == 0): blocks walk down 32-pixel strips through a 12-slot row ring
    (BF, 3, 2, 64, 64, 64, 128, "k3.2d.ppst", ""),       # 64 tiles over 64 blocks: one 4-row segment each (prefetch deeper than the stream)
    (BF, 3, 1, 32, 32, 1024, 1024, "k3.2d.ppst", ""),    # 128 pairs x 2 splits: half a strip each
    (BF, 3, 2, 152, 160, 64, 128, "k3.2d.ppst", ""),     # 380 tiles over 190 blocks: 8-row segments, some across a strip boundary
    (BF, 3, 3, 64, 96, 128, 256, "k3.2d.ppst", ""),      # 4 pairs x 48 splits of 3 tiles
    (BF, 3, 2, 128, 64, 256, 128, "k3.2d.ppst", ""),     # 4 pairs x 64 splits: whole 128-row strips (the ring wraps ten times)
    (BF, 3, 2, 20, 36, 64, 128, "k3.2d.ppwr", "MIS_WGRAD_PP_ROW"),     # ragged in W as well (forced)
    (BF, 3, 2, 150, 160, 64, 128, "k3.2d.ppwr", ""),     # 380 tiles over 128 blocks... the tile loop is taken
    (BF, 3, 2, 64, 96, 64, 64, "k3.2d.ppsr", "MIS_WGRAD_PP_NOSTREAM"),        # 64-column tiles: 32 x 8-pixel tiles, four rows per wave group, two slabs per block
    # ... and their streaming form (H % 8 == 0, W % 32 == 0): two row streams per block, one per wave group
    (BF, 3, 2, 64, 96, 64, 64, "k3.2d.ppss", ""),        # 48 tiles over 48 blocks: one 4-row segment per group
    (BF, 3, 2, 152, 160, 128, 64, "k3.2d.ppss", ""),     # two input-channel tiles; 190 tiles over 128 blocks: the groups' streams differ in length at strip boundaries
    (BF, 3, 4, 128, 64, 64, 64, "k3.2d.ppss", ""),       # 128 tiles, one pair: whole and half strips, the ring wraps
    (BF, 3, 2, 20, 36, 64, 64, "k3.2d.ppsr", "MIS_WGRAD_PP_ROW"),
    (BF, 3, 2, 150, 160, 128, 64, "k3.2d.ppsr", ""),
    (BF, 3, 2, 64, 64, 64, 128, "k3.2d.ppw", "MIS_WGRAD_PP_NOROW"),    # the 16-wide tiles stay reachable
    (BF, 3, 2, 20, 36, 64, 64, "k3.2d.tr", "MIS_WGRAD_NOPP"),     # the kernel behind it
    (BF, 3, 1, 9, 17, 1024, 1024, "k3.2d.tr", "MIS_WGRAD_NOPP"),
    (BF, 3, 2, 40, 40, 512, 256, "k3.2d.tr", "MIS_WGRAD_NOPP"),
    (BF, 1, 2, 10, 18, 256, 512, "k1.2d.wide", ""),   # 128 x 128 channel tiles
    (BF, 1, 1, 9, 17, 2048, 1024, "k1.2d.wide", ""),
    (BF, 1, 2, 10, 18, 64, 256, "k1.2d.tr", ""),      # Cin not a multiple of 128: 64 x 64 tiles
    # ping-pong 1x1 weight gradient (Cin % 128 == 0, Cout % 256 == 0, N*H*W % 64 == 0): 128 x 256 channel tiles, 64-pixel stream elements
    (BF, 1, 2, 16, 32, 128, 256, "k1.2d.ppg", ""),    # 16 elements over 16 blocks: one step each
    (BF, 1, 2, 64, 64, 256, 512, "k1.2d.ppg", ""),    # 4 channel-tile pairs x 64 splits of two elements
    (BF, 1, 4, 64, 96, 128, 256, "k1.2d.ppg", ""),    # the whole dW in one tile; 384 elements over 192 blocks
    (BF, 1, 1, 32, 32, 1024, 2048, "k1.2d.ppg", ""),  # 64 pairs x 4 splits of four elements: the ring wraps
    (BF, 1, 3, 64, 64, 128, 256, "k1.2d.ppg", ""),    # 192 elements over 192 blocks... and (below) the kernel it replaces
    (BF, 1, 2, 64, 64, 256, 512, "k1.2d.wide", "MIS_WGRAD_K1_NOPP"),
    (F32, 3, 2, 20, 36, 32, 64, "k3.2d", ""),
    (F32, 3, 1, 9, 17, 256, 128, "k3.2d", ""),
    (F32, 1, 2, 10, 18, 128, 256, "k1.2d", ""),
]


@pytest.mark.parametrize("case", WG_CASES, ids=lambda c: f"{'bf16' if c[0] == BF else 'f32'}-k{c[1]}-{c[2]}x{c[3]}x{c[4]}-{c[5]}to{c[6]}-{c[8]}")
def test_wgrad_every_branch(case, switches):
    ops = _ops()
    dtype, ks, N, H, W, Cin, Cout, want_cfg, env = case
    if env:
        switches(env, 1)
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


# ---------------------------------------------------------------------------------------------------------
# 3-D: nn.Conv3d(k3, p1, bias=False) behind GroupNorm (reference model/unet3d/buildingblocks.py:64-66,87-92), the decoder's concat + nearest upsample
# (:546-548, :671-673) as two-source addressing.  Two families:
#   * the ping-pong kernels (conv3d_pp.hip `k3.3d.ppc*`, wgrad_pp.hip `k3.3d.ppw / pps`): bf16, plain single-source operand (the engines hand over the normalised
#     tensor written by mis_gn_apply) - every instantiation (20- / 32-row tiles x 128- / 64-column blocks) on the channel plans of UNet3D's layers, ragged grids,
#     several tiles per persistent block, ragged depth groups;
#   * the lock-step kernels (`k3.3d.bn*`, `k3.3d.tr`, `k3.3d`): fp32 always, bf16 when the operand carries the GroupNorm fold / a second source.
# (dtype, (N, D, H, W), Cin or (C0, C1) for [encoder | nearest-upsampled] sources + GroupNorm fold, Cout, conv configuration, wgrad configuration, switches)
K3D_CASES = [
    (BF, (1, 5, 20, 20), 64, 64, "k3.3d.ppc5n2", "k3.3d.pps", {}),                       # 20-row tiles fit; W ragged (16 + 4); depth groups 4 + 1
    (BF, (2, 3, 22, 36), 192, 64, "k3.3d.ppc8n2", "k3.3d.pps", {}),                      # dec2.conv1's plan: 6 K chunks x 3 slices; 32-row tiles, ragged both ways
    (BF, (1, 4, 22, 36), 192, 64, "k3.3d.ppc5n2", "k3.3d.pps", {"MIS_CONV3D_PF": 5}),    # the same plan on ragged 20-row tiles
    (BF, (1, 6, 20, 24), 384, 128, "k3.3d.ppc5", "k3.3d.ppw", {}),                       # dec1.conv1
    (BF, (1, 3, 10, 18), 768, 256, "k3.3d.ppc5", "k3.3d.ppw", {}),                       # dec0.conv1: 24 K chunks, two column tiles (spatial-major order)
    (BF, (1, 3, 10, 18), 768, 256, "k3.3d.ppc8", "k3.3d.ppw", {"MIS_CONV3D_PF": 8}),
    (BF, (1, 2, 9, 17), 256, 512, "k3.3d.ppc8", "k3.3d.ppw", {"MIS_CONV3D_PF": 8, "MIS_CONV3D_ZG": 1}),      # enc3.conv2: four column tiles, no depth grouping
    (BF, (1, 3, 12, 20), 128, 384, "k3.3d.ppc5", "k3.3d.ppw", {}),                       # dgrad of dec1.conv1: three column tiles -> column-tile-major order
    (BF, (1, 3, 12, 20), 64, 192, "k3.3d.ppc5n6", "k3.3d.pps", {}),                      # dgrad of dec2.conv1: ONE 192-column tile (wave tile 80 px x 96 ch)
    (BF, (2, 5, 40, 36), 128, 192, "k3.3d.ppc5n6", "k3.3d.pps", {}),                     # 192-column blocks, 4 K chunks, 60 tiles, ragged width
    (BF, (2, 24, 40, 48), 64, 64, "k3.3d.ppc10n2", "k3.3d.pps", {}),                     # 40-row tiles (64-column blocks only); planes cross the sample boundary
    (BF, (1, 3, 22, 36), 64, 64, "k3.3d.ppc10n2", "k3.3d.pps", {"MIS_CONV3D_PF": 10}),   # ... on a ragged grid
    (BF, (3, 30, 40, 48), 64, 64, "k3.3d.ppc5n2", "k3.3d.pps", {"MIS_CONV3D_PF": 5}),    # 540 tiles > 256 persistent blocks: the tile loop is taken
    (BF, (1, 40, 64, 48), 64, 128, "k3.3d.ppc8", "k3.3d.ppw", {}),                       # 240 tiles... 32-row tiles exact; interior tiles; depth groups of 4
    (BF, (1, 1, 20, 16), 64, 64, "k3.3d.ppc5n2", "k3.3d.pps", {}),                       # exactly one tile, one plane: both neighbour planes are padding
    (BF, (1, 6, 20, 32), 64, 64, "k3.3d.ppc5n2", "k3.3d.ppsr", {}),                      # weight gradient on the row variants (32-wide tiles fit)
    (BF, (2, 5, 12, 64), 128, 128, "k3.3d.ppc5", "k3.3d.ppwr", {"MIS_WGRAD_PP_NOSTREAM": 1}),
    (BF, (2, 5, 12, 64), 128, 128, "k3.3d.ppc5", "k3.3d.ppst", {}),                       # ... and their streaming forms: planes above / below the volume read as zero
    (BF, (1, 6, 24, 32), 64, 64, "k3.3d.ppc8n2", "k3.3d.ppss", {}),
    (BF, (2, 3, 16, 64), 192, 64, "k3.3d.ppc5n2", "k3.3d.ppss", {}),                      # three input-channel tiles x three depth slices
    (BF, (1, 3, 10, 18), 192, 64, "k3.3d.ppc5n2", "k3.3d.ppsr", {"MIS_WGRAD_PP_ROW": 1}),        # ... ragged both ways, three input-channel tiles x three depth slices
    (BF, (1, 7, 20, 16), 64, 128, "k3.3d.ppc5", "k3.3d.ppw", {"MIS_CONV3D_COLMAJOR": 1, "MIS_CONV3D_ZG": 3}),
    # the kernels behind them
    (BF, (1, 5, 20, 20), 64, 64, "k3.3d.bn64", "k3.3d.tr", {"MIS_CONV3D_NOPP": 1, "MIS_WGRAD3D_NOPP": 1}),
    (BF, (1, 6, 20, 24), 384, 128, "k3.3d.bn128", "k3.3d.tr", {"MIS_CONV3D_NOPP": 1, "MIS_WGRAD3D_NOPP": 1}),
    # operand with the GroupNorm fold and two sources: the lock-step kernels in both precisions
    (BF, (2, 4, 12, 20), (64, 128), 64, "k3.3d.bn64", "k3.3d.tr", {}),
    (BF, (1, 4, 12, 20), (128, 256), 128, "k3.3d.bn128", "k3.3d.tr", {}),
    (BF, (1, 2, 6, 10), (256, 512), 256, "k3.3d.bn128", "k3.3d.tr", {}),
    (F32, (2, 4, 12, 20), (64, 128), 64, "k3.3d.bn64", "k3.3d", {}),
    (F32, (1, 4, 12, 20), (128, 256), 128, "k3.3d.bn128", "k3.3d", {}),
    (F32, (1, 2, 6, 10), (256, 512), 256, "k3.3d.bn128", "k3.3d", {}),
    (F32, (1, 2, 9, 17), 256, 512, "k3.3d.bn128", "k3.3d", {"MIS_CONV3D_F32_NOPP": 1, "MIS_WGRAD_F32_NOPP": 1}),
    (F32, (1, 5, 20, 20), 64, 64, "k3.3d.bn64", "k3.3d", {"MIS_CONV3D_F32_NOPP": 1, "MIS_WGRAD_F32_NOPP": 1}),
    # round 5: the fp32 all-DMA kernels (conv3d_f32.hip: 8 x 16-pixel plane tiles x 64 | 128 columns; wgrad_f32.hip: 16-pixel strips streamed row by row) - the channel
    # plans of UNet3D's layers on ragged grids, single planes (both neighbour planes padding), planes of two samples, several split-K ranges per (ci, co, kd) tile
    (F32, (1, 2, 9, 17), 256, 512, "k3.3d.f32pp128", "k3.3d.f32s", {"MIS_CONV3D_F32_WIDE": 1}),                    # enc3.conv2: four column tiles, eight K chunks, ragged both ways
    (F32, (1, 5, 20, 20), 64, 64, "k3.3d.f32pp64", "k3.3d.f32s", {}),
    (F32, (2, 4, 16, 32), 32, 64, "k3.3d.f32pp64", "k3.3d.f32s32", {}),                    # enc0.conv2's forward over its 32 real channels: ONE K chunk per depth slice; round 6: its weight gradient on 32-channel blocks
    (F32, (2, 5, 24, 40), 64, 32, "k3.3d.f32pp32", None, {}),                              # round 6: enc0.conv2's dgrad on 32-column tiles (planes of two samples, ragged width)
    (F32, (1, 3, 9, 17), 64, 96, "k3.3d.f32pp32", None, {}),                               # three 32-column tiles, ragged both ways
    (F32, (1, 3, 9, 17), 96, 64, "k3.3d.f32pp64", "k3.3d.f32s32", {}),                     # three 32-channel ci tiles x three depth slices, ragged strips
    (F32, (2, 8, 40, 48), 32, 128, "k3.3d.f32pp128", "k3.3d.f32s32", {"MIS_CONV3D_F32_WIDE": 1}),          # two co tiles; rows split over many blocks (segments across strip / plane boundaries)
    (F32, (1, 1, 20, 16), 32, 64, "k3.3d.f32pp64", "k3.3d.f32s32", {}),                    # one plane: kd = 0 / 2 blocks see only padding
    (F32, (2, 6, 24, 40), 192, 64, "k3.3d.f32pp64", "k3.3d.f32s", {}),                     # dec2.conv1: six chunks; weight gradient: three ci tiles x three depth slices
    (F32, (1, 3, 12, 20), 64, 192, "k3.3d.f32pp64", "k3.3d.f32s", {}),                     # its dgrad: three 64-column tiles
    (F32, (2, 3, 8, 16), 768, 256, "k3.3d.f32pp128", "k3.3d.f32s", {"MIS_CONV3D_F32_WIDE": 1}),                    # dec0.conv1: 24 chunks, one tile per plane
    (F32, (1, 1, 20, 16), 64, 64, "k3.3d.f32pp64", "k3.3d.f32s", {}),                      # one plane: only the centre depth slice contributes
    (F32, (2, 8, 40, 48), 64, 128, "k3.3d.f32pp128", "k3.3d.f32s", {"MIS_CONV3D_F32_WIDE": 1}),                    # 240 tiles; strips of three 16-pixel columns, rows split over many blocks
    (F32, (3, 2, 7, 5), 128, 128, "k3.3d.f32pp128", "k3.3d.f32s", {"MIS_CONV3D_F32_WIDE": 1}),
    (F32, (2, 16, 16, 16), 256, 256, "k3.3d.f32pp64", "k3.3d.f32s", {}),                   # cfg4's 16^3 level: 64 spatial tiles - 64-column tiles are chosen to fill the chip
]


def _id3(c):
    cin = f"{c[2][0]}+{c[2][1]}" if isinstance(c[2], tuple) else str(c[2])
    return f"{'bf16' if c[0] == BF else 'f32'}-{'x'.join(map(str, c[1]))}-{cin}to{c[3]}-{c[4]}" + "".join(f"-{k[4:]}{v}" for k, v in c[6].items())


@pytest.mark.parametrize("case", K3D_CASES, ids=_id3)
def test_conv3d_every_branch(case, switches):
    ops = _ops()
    dtype, (N, D, H, W), cin, Cout, want_conv, want_wg, sw = case
    for k, v in sw.items():
        switches(k, v)
    two = isinstance(cin, tuple)
    Cin = sum(cin) if two else cin
    w = rnd(Cout, Cin, 3, 3, 3, seed=141, scale=(27 * Cin) ** -0.5)
    wf = torch.empty(27, Cout, Cin, dtype=dtype, device=DEV)
    ops.pack_conv_weight(w.to(DEV), wf, None)
    m = rnd(N, Cout, D, H, W, seed=143)
    dy = rnd(N, Cout, D, H, W, seed=144)
    if two:
        C0, C1 = cin
        x0 = rnd(N, C0, D, H, W, seed=140)
        x1 = rnd(N, C1, D // 2, H // 2, W // 2, seed=145)
        scale = 1 + 0.3 * rnd(N, Cin, seed=146)
        shift = 0.3 * rnd(N, Cin, seed=147)
        xcat = torch.cat((q(x0, dtype), F.interpolate(q(x1, dtype), size=(D, H, W), mode="nearest")), 1)
        xop = q(xcat * scale.view(N, Cin, 1, 1, 1) + shift.view(N, Cin, 1, 1, 1), dtype)       # the fold is rounded to the storage type before the MFMA
        kw = dict(x1=to_nhwc(x1, dtype), in_scale=scale.to(DEV), in_shift=shift.to(DEV))
        xd = to_nhwc(x0, dtype)
    else:
        x = rnd(N, Cin, D, H, W, seed=140)
        xop = q(x, dtype)
        kw = {}
        xd = to_nhwc(x, dtype)
    wq = q(w, dtype).requires_grad_(True)
    ref = F.conv3d(xop, wq, None, padding=1)
    ref.backward(q(dy, dtype))
    ref = ref.detach()
    t = tol(dtype, 27 * Cin)
    grid = (N, D, H, W)
    # (a) forward form: ReLU, written into a channel slice of a wider buffer
    ybuf = torch.full((N, D, H, W, Cout + 64), float("nan"), dtype=dtype, device=DEV)
    ops.conv_igemm(xd, wf, ops.View(ybuf, 64, Cout), ksize=3, Cin=Cin, Cout=Cout, grid=grid, relu=True, **kw)
    cfg = ops.conv_last_dispatch()
    assert cfg == want_conv or (cfg.startswith(want_conv) and want_conv.startswith("k3.3d.bn")), f"case meant for {want_conv} ran {cfg}"
    assert_close(from_nhwc(ybuf[..., 64:].contiguous()), F.relu(ref), f"3-D fwd {cfg}", **t)
    assert torch.isnan(ybuf[..., :64].float()).all(), "wrote outside the channel slice"
    # (b) masked form (ReLU backward in the epilogue), plain destination
    y2 = torch.full((N, D, H, W, Cout), float("nan"), dtype=dtype, device=DEV)
    ops.conv_igemm(xd, wf, y2, ksize=3, Cin=Cin, Cout=Cout, grid=grid, mask=to_nhwc(m, dtype), **kw)
    assert _same_kernel(ops.conv_last_dispatch(), cfg)
    assert_close(from_nhwc(y2), ref * (q(m, dtype) > 0), f"3-D mask {cfg}", **t)
    # (c) weight gradient of the same operand
    if want_wg is None:         # (a 32-column case: the weight gradients need Cout % 64 == 0)
        return
    dw = torch.full((Cout, Cin, 3, 3, 3), float("nan"), device=DEV)
    ops.wgrad(xd, to_nhwc(dy, dtype), dw, ksize=3, Cin=Cin, Cout=Cout, grid=grid, **kw)
    wcfg, nsplit = ops.wgrad_last_dispatch()
    assert wcfg == want_wg, f"case meant for {want_wg} ran {wcfg}"
    k = N * D * H * W
    wt = dict(rtol=1e-4, atol=1e-4 * k ** 0.5) if not (two and dtype == BF) else dict(rtol=1e-4, atol=2e-4 * k ** 0.5)
    assert_close(dw, wq.grad, f"3-D wgrad {wcfg} nsplit={nsplit}", **wt)


@pytest.mark.parametrize("switch", ["MIS_CONV_PPS", "MIS_CONV_PPC2"])
@pytest.mark.parametrize("shape", [(2, 64, 64, 128, 128), (3, 150, 170, 64, 256), (1, 32, 16, 64, 128), (5, 150, 170, 128, 128)], ids=lambda s: "x".join(map(str, s)))
def test_conv_pps_experiment_is_bit_identical_to_conv_ppc(shape, switch):
    """the two round-5 experiments on the dominant 2-D kernel - conv_pps_kernel (MIS_CONV_PPS=1: one wave per SIMD at 512 registers, both fragment sets resident; 12 % slower) and
    conv_ppc2_kernel (MIS_CONV_PPC2=1: halo DMA offsets from an LDS table, weight fragments rolling through the M segment; 0-7 % slower) - are off by default (EXPERIMENTS.md) and
    sum in conv_ppc_kernel<8, 4>'s order: bit-identical outputs in the forward, bf16-mask and ReLU-bits forms, ragged grids included"""
    ops = _ops()
    if not ops.build_has_experiments():
        pytest.skip("the shipped library does not contain csrc/experiments (round 6: `make -C mdeical_image_segmentation_amd/csrc EXPERIMENTS=1` builds them in)")
    tagp = "k3.2d.pps" if switch == "MIS_CONV_PPS" else "k3.2d.ppc2"
    N, H, W, Cin, Cout = shape
    x = to_nhwc(rnd(N, Cin, H, W, seed=190), BF)
    w = rnd(Cout, Cin, 3, 3, seed=191, scale=(9 * Cin) ** -0.5)
    wf = torch.empty(9, Cout, Cin, dtype=BF, device=DEV)
    ops.pack_conv_weight(w.to(DEV), wf, None)
    b = rnd(Cout, seed=192).to(DEV)
    m = to_nhwc(rnd(N, Cout, H, W, seed=193), BF)
    bits = torch.empty(ops.relu_bits_bytes(N, H, W, Cout), dtype=torch.uint8, device=DEV)
    ops.relu_bits(m, bits)
    for form, kw in (("fwd", dict(bias=b, relu=True)), ("mask", dict(mask=m)), ("bits", dict(mask_bits=bits))):
        outs = []
        for pps in (0, 1):
            y = torch.full((N, H, W, Cout), float("nan"), dtype=BF, device=DEV)
            with ops.dispatch_switches(**{switch: pps}):
                ops.conv_igemm(x, wf, y, ksize=3, Cin=Cin, Cout=Cout, **kw)
                assert ops.conv_last_dispatch().startswith(tagp if pps else "k3.2d.ppc8"), ops.conv_last_dispatch()
            outs.append(y)
        assert torch.equal(outs[0], outs[1]), (form, (outs[0].float() - outs[1].float()).abs().max().item())


@pytest.mark.parametrize("shape", [((2, 5, 20, 24), 64, 64), ((1, 3, 9, 17), 192, 64), ((1, 4, 16, 16), 128, 256), ((2, 1, 8, 40), 32, 64)], ids=lambda s: f"{'x'.join(map(str, s[0]))}-{s[1]}to{s[2]}")
def test_conv3d_f32_kernel_is_bit_identical_to_the_lockstep_kernel(shape):
    """round 5: conv3d_f32_kernel sums the products of an output element in the order of the lock-step kernel it replaces (chunk, kd, kh, kw, channel; planes outside the
    volume skipped instead of added as zeros) - the fp32 parity mode's forward bits did not move when the kernel changed"""
    ops = _ops()
    (N, D, H, W), Cin, Cout = shape
    x = to_nhwc(rnd(N, Cin, D, H, W, seed=180), F32)
    w = rnd(Cout, Cin, 3, 3, 3, seed=181, scale=(27 * Cin) ** -0.5)
    wf = torch.empty(27, Cout, Cin, dtype=F32, device=DEV)
    ops.pack_conv_weight(w.to(DEV), wf, None)
    ya = torch.empty(N, D, H, W, Cout, dtype=F32, device=DEV)
    yb = torch.empty_like(ya)
    ops.conv_igemm(x, wf, ya, ksize=3, Cin=Cin, Cout=Cout, grid=(N, D, H, W), relu=True)
    assert ops.conv_last_dispatch().startswith("k3.3d.f32pp")
    with ops.dispatch_switches(MIS_CONV3D_F32_NOPP=1):
        ops.conv_igemm(x, wf, yb, ksize=3, Cin=Cin, Cout=Cout, grid=(N, D, H, W), relu=True)
        assert ops.conv_last_dispatch().startswith("k3.3d.bn")
    assert torch.equal(ya, yb), (ya - yb).abs().max().item()


@pytest.mark.parametrize("shape", [((2, 5, 24, 40), 64, 32), ((1, 3, 9, 17), 128, 96)], ids=lambda s: f"{'x'.join(map(str, s[0]))}-{s[1]}to{s[2]}")
def test_conv3d_f32_32_column_tiles_are_bit_identical_to_the_padded_64_column_launch(shape):
    """round 6: conv3d_f32_kernel<1> (32-column tiles: the dgrad of encoders.0 SingleConv2, reference buildingblocks.py:202-211) against what the engine ran until round 5 -
    the same weights zero-padded to the next multiple of 64 columns on conv3d_f32_kernel<2>, and against the lock-step kernel on the padded weights: the real columns are
    bit-identical (an output element's products are summed in the same order whatever the column tile)"""
    ops = _ops()
    (N, D, H, W), Cin, Cout = shape
    Cp = (Cout + 63) // 64 * 64
    x = to_nhwc(rnd(N, Cin, D, H, W, seed=182), F32)
    w = rnd(Cout, Cin, 3, 3, 3, seed=183, scale=(27 * Cin) ** -0.5)
    wp = torch.zeros(Cp, Cin, 3, 3, 3)
    wp[:Cout] = w
    wf = torch.empty(27, Cout, Cin, dtype=F32, device=DEV)
    wfp = torch.empty(27, Cp, Cin, dtype=F32, device=DEV)
    ops.pack_conv_weight(w.to(DEV), wf, None)
    ops.pack_conv_weight(wp.to(DEV), wfp, None)
    m = to_nhwc(rnd(N, Cout, D, H, W, seed=184), F32)
    ya = torch.full((N, D, H, W, Cout), float("nan"), dtype=F32, device=DEV)
    ops.conv_igemm(x, wf, ya, ksize=3, Cin=Cin, Cout=Cout, grid=(N, D, H, W), mask=m)
    assert ops.conv_last_dispatch() == "k3.3d.f32pp32"
    mp = torch.ones(N, D, H, W, Cp, dtype=F32, device=DEV)
    mp[..., :Cout] = m
    for sw, tag in ((dict(), "k3.3d.f32pp64"), (dict(MIS_CONV3D_F32_NOPP=1), "k3.3d.bn128" if Cp % 128 == 0 else "k3.3d.bn64")):
        yb = torch.full((N, D, H, W, Cp), float("nan"), dtype=F32, device=DEV)
        with ops.dispatch_switches(**sw):
            ops.conv_igemm(x, wfp, yb, ksize=3, Cin=Cin, Cout=Cp, grid=(N, D, H, W), mask=mp)
            assert ops.conv_last_dispatch().startswith(tag), ops.conv_last_dispatch()
        assert torch.equal(ya, yb[..., :Cout]), (tag, (ya - yb[..., :Cout]).abs().max().item())
        assert (yb[..., Cout:] == 0).all()


@pytest.mark.parametrize("case", [((2, 5, 24, 40), 64, 32, None), ((1, 3, 9, 17), 128, 64, None), ((2, 4, 12, 20), 64, (64, 128), True), ((1, 4, 10, 18), 128, (128, 256), False),
                                  ((2, 16, 16, 16), 256, 256, None)],
                         ids=lambda c: f"{'x'.join(map(str, c[0]))}-{c[1]}to{c[2] if not isinstance(c[2], tuple) else '+'.join(map(str, c[2]))}{'' if c[3] is None else ('-up' if c[3] else '-full')}")
def test_conv3d_f32_statistics_epilogue(case):
    """round 6 (MisConvDesc.st_mode): the fp32 3x3x3 kernels leave per-channel sums of their output as per-tile partial rows - mode 1: S1 = sum out, S2 = sum out * x, the two
    reductions of the GroupNorm backward (reference buildingblocks.py:87-92), x from one tensor or from two (the second on the half grid: the decoder's nearest-upsampled
    source); mode 2: sum out / sum out^2 of the stored post-ReLU output (the next GroupNorm's statistics).  Against fp64 sums of the kernel's OWN output; ragged tiles,
    planes of two samples, 32- / 64- / 128-column launches; the output itself must not move (bit-identical to a launch without the epilogue)."""
    ops = _ops()
    (N, D, H, W), Cin, cout, up = case
    two = isinstance(cout, tuple)
    Cout = sum(cout) if two else cout
    x = to_nhwc(rnd(N, Cin, D, H, W, seed=185), F32)
    w = rnd(Cout, Cin, 3, 3, 3, seed=186, scale=(27 * Cin) ** -0.5)
    wf = torch.empty(27, Cout, Cin, dtype=F32, device=DEV)
    ops.pack_conv_weight(w.to(DEV), wf, None)
    grid = (N, D, H, W)
    y_plain = torch.empty(N, D, H, W, Cout, dtype=F32, device=DEV)
    ops.conv_igemm(x, wf, y_plain, ksize=3, Cin=Cin, Cout=Cout, grid=grid)
    assert ops.conv_stats_supported(x, y_plain, Cin=Cin, Cout=Cout, grid=grid)
    S1 = torch.full((N, Cout), float("nan"), device=DEV)
    S2 = torch.full((N, Cout), float("nan"), device=DEV)
    # mode 1
    if two:
        C0, C1 = cout
        g0 = to_nhwc(rnd(N, C0, D, H, W, seed=187), F32)
        g1 = to_nhwc(rnd(N, C1, D // 2, H // 2, W // 2, seed=188) if up else rnd(N, C1, D, H, W, seed=188), F32)
        xs = torch.cat((g0, g1.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3) if up else g1), -1)
        st = dict(mode=1, x0=g0, x1=g1, up=up, S1=S1, S2=S2)
    else:
        xs = to_nhwc(rnd(N, Cout, D, H, W, seed=187), F32)
        st = dict(mode=1, x0=xs, S1=S1, S2=S2)
    y = torch.full_like(y_plain, float("nan"))
    ops.conv_igemm(x, wf, y, ksize=3, Cin=Cin, Cout=Cout, grid=grid, stats=st)
    assert ops.conv_last_dispatch().startswith("k3.3d.f32pp")
    assert torch.equal(y, y_plain)
    yd, xd = y.double().view(N, -1, Cout), xs.double().view(N, -1, Cout)
    k = D * H * W
    def tight(got, terms, what):          # |error| <= 1e-6 x the sum of the terms' magnitudes (fp32 partials of 64 elements, the rest in double): cancellation is not excused
        want, mag = terms.sum(1), terms.abs().sum(1)
        err = (got.double().cpu() - want.cpu()).abs()
        assert bool((err <= 1e-6 * mag.cpu() + 1e-30).all()), (what, (err / (mag.cpu() + 1e-30)).max().item())

    tight(S1, yd, "S1")
    tight(S2, yd * xd, "S2")
    # mode 2 (ReLU output)
    S1.fill_(float("nan")); S2.fill_(float("nan"))
    ops.conv_igemm(x, wf, y, ksize=3, Cin=Cin, Cout=Cout, grid=grid, relu=True, stats=dict(mode=2, S1=S1, S2=S2))
    assert torch.equal(y, torch.relu(y_plain))
    yd = y.double().view(N, -1, Cout)
    tight(S1, yd, "sum")
    tight(S2, yd * yd, "sumsq")
    if Cout % 128 == 0:         # the 128-column instantiation (mode 1 again)
        S1.fill_(float("nan")); S2.fill_(float("nan"))
        st.update(S1=S1, S2=S2)
        with ops.dispatch_switches(MIS_CONV3D_F32_WIDE=1):
            ops.conv_igemm(x, wf, y, ksize=3, Cin=Cin, Cout=Cout, grid=grid, stats=st)
            assert ops.conv_last_dispatch() == "k3.3d.f32pp128"
        assert torch.equal(y, y_plain)
        yd = y.double().view(N, -1, Cout)
        tight(S1, yd, "S1 wide")
        tight(S2, yd * xd, "S2 wide")


@pytest.mark.parametrize("dtype", [BF, F32])
def test_gn_apply_matches_the_operand_fold(dtype):
    """mis_gn_apply (the normalised tensor the bf16 engines write once per SingleConv) == what mis_conv_igemm computes when it folds the affine into staging:
    the two routes through a SingleConv must give BIT-identical outputs"""
    ops = _ops()
    N, D, H, W, C0, C1, Cout = 2, 4, 12, 20, 64, 128, 64
    Cin = C0 + C1
    x0, x1 = rnd(N, C0, D, H, W, seed=150), rnd(N, C1, D // 2, H // 2, W // 2, seed=151)
    scale, shift = (1 + 0.3 * rnd(N, Cin, seed=152)).to(DEV), (0.3 * rnd(N, Cin, seed=153)).to(DEV)
    w = rnd(Cout, Cin, 3, 3, 3, seed=154, scale=(27 * Cin) ** -0.5)
    wf = torch.empty(27, Cout, Cin, dtype=dtype, device=DEV)
    ops.pack_conv_weight(w.to(DEV), wf, None)
    x0d, x1d = to_nhwc(x0, dtype), to_nhwc(x1, dtype)
    grid = (N, D, H, W)
    xn = torch.full((N, D, H, W, Cin), float("nan"), dtype=dtype, device=DEV)
    ops.gn_apply(x0d, C0, False, grid, scale, shift, Cin, 0, xn)
    ops.gn_apply(x1d, C1, True, grid, scale, shift, Cin, C0, xn)
    xcat = torch.cat((q(x0, dtype), F.interpolate(q(x1, dtype), size=(D, H, W), mode="nearest")), 1)
    want = torch.addcmul(shift.cpu().view(N, Cin, 1, 1, 1), xcat, scale.cpu().view(N, Cin, 1, 1, 1))
    assert_close(from_nhwc(xn), q(want, dtype), "gn_apply", rtol=1e-2 if dtype == BF else 1e-6, atol=1e-2 if dtype == BF else 1e-6)
    ya = torch.empty(N, D, H, W, Cout, dtype=dtype, device=DEV)
    yb = torch.empty_like(ya)
    with ops.dispatch_switches(MIS_CONV3D_NOPP=1, MIS_CONV3D_F32_NOPP=1):          # the same (lock-step) kernel for both routes: only the operand path differs
        ops.conv_igemm(x0d, wf, ya, ksize=3, Cin=Cin, Cout=Cout, grid=grid, x1=x1d, in_scale=scale, in_shift=shift, relu=True)
        ops.conv_igemm(xn, wf, yb, ksize=3, Cin=Cin, Cout=Cout, grid=grid, relu=True)
    assert torch.equal(ya, yb), (ya.float() - yb.float()).abs().max().item()


# GroupNorm backward continued in the dgrad epilogue (MisConvDesc.gn_p, round 4): (grid, Cin = dy channels, Cout = padded operand channels, real channels of x / dx, configuration, switches)
GN_EPI_CASES = [
    ((2, 3, 40, 24), 64, 64, 64, "k3.3d.ppc10n2.gn", {}),                      # decoders.2 SingleConv2: 40-row tiles, ragged width, planes of two samples (two p / q / r rows)
    ((2, 5, 40, 36), 64, 64, 32, "k3.3d.ppc10n2.gn", {}),                      # encoders.0 SingleConv2: 32 real channels, the upper wave slice is dropped (NaN canary)
    ((1, 4, 22, 36), 128, 64, 64, "k3.3d.ppc5n2.gn", {"MIS_CONV3D_PF": 5}),    # encoders.1 SingleConv2's dgrad (128 -> 64) on ragged 20-row tiles
    ((3, 3, 24, 20), 64, 64, 64, "k3.3d.ppc8n2.gn", {"MIS_CONV3D_PF": 8}),
    ((2, 6, 20, 24), 128, 128, 128, "k3.3d.ppc5.gn", {}),                      # decoders.1 SingleConv2
    ((1, 3, 10, 18), 512, 256, 256, "k3.3d.ppc5.gn", {}),                      # encoders.3 SingleConv2's dgrad: two column tiles, 16 K chunks
    ((2, 2, 34, 20), 256, 128, 128, "k3.3d.ppc8.gn", {"MIS_CONV3D_PF": 8}),    # the 256-VGPR instantiation (vector bias add, accumulators re-armed by the epilogue)
    ((2, 30, 40, 48), 64, 64, 64, "k3.3d.ppc5n2.gn", {"MIS_CONV3D_PF": 5}),    # 360 tiles > 256 persistent blocks: the next tile's p / q / r are staged under the K loop
]


@pytest.mark.parametrize("relu_mask", [True, False])
@pytest.mark.parametrize("case", GN_EPI_CASES, ids=lambda c: f"{'x'.join(map(str, c[0]))}-{c[1]}to{c[2]}r{c[3]}-{c[4]}")
def test_conv3d_gn_backward_epilogue(case, relu_mask, switches):
    """out = [x > 0] * (p * dgrad + q * x + r) from the fp32 accumulator of the 3x3x3 dgrad (the 'gcr' SingleConv's backward through GroupNorm + ReLU, reference
    model/unet3d/buildingblocks.py:87-92) against the same expression in torch on the bf16-rounded operands."""
    ops = _ops()
    (N, D, H, W), Cin, Cout, Cr, want, sw = case
    for k, v in sw.items():
        switches(k, v)
    dy = rnd(N, Cin, D, H, W, seed=170)
    w = rnd(Cin, Cout, 3, 3, 3, seed=171, scale=(27 * Cin) ** -0.5)          # forward weight [co = Cin here][ci = Cout here]: the dgrad contracts over co
    wd = torch.empty(27, Cout, Cin, dtype=BF, device=DEV)
    wt = w.flip(2, 3, 4).transpose(0, 1).contiguous()                       # dgrad as a forward convolution with the mirrored, transposed filter
    ops.pack_conv_weight(wt.to(DEV), wd, None)
    x = F.relu(rnd(N, Cr, D, H, W, seed=172)) if relu_mask else rnd(N, Cr, D, H, W, seed=172)
    p_, q_, r_ = 1 + 0.3 * rnd(N, Cr, seed=173), 0.2 * rnd(N, Cr, seed=174), 0.1 * rnd(N, Cr, seed=175)
    dgrad = F.conv3d(q(dy, BF), q(wt, BF), None, padding=1)[:, :Cr]
    bc = lambda t_: t_.view(N, Cr, 1, 1, 1)
    ref = bc(p_) * dgrad + (bc(q_) * q(x, BF) + bc(r_))
    if relu_mask:
        ref = ref * (q(x, BF) > 0)
    out = torch.full((N, D, H, W, Cr + 64), float("nan"), dtype=BF, device=DEV)          # a channel slice of a wider buffer: nothing else may be written
    xd = to_nhwc(x, BF)
    ops.conv_igemm(to_nhwc(dy, BF), wd, ops.View(out, 0, Cr), ksize=3, Cin=Cin, Cout=Cout, Cout0=Cr, grid=(N, D, H, W), mask=xd,
                   gn_bwd=(p_.to(DEV).contiguous(), q_.to(DEV).contiguous(), r_.to(DEV).contiguous(), relu_mask))
    cfg = ops.conv_last_dispatch()
    assert cfg == want, f"case meant for {want} ran {cfg}"
    assert_close(from_nhwc(out[..., :Cr].contiguous()), ref, f"3-D GN-backward epilogue {cfg}", **tol(BF, 27 * Cin))
    assert torch.isnan(out[..., Cr:].float()).all(), "wrote outside the real channels"


@pytest.mark.parametrize("grid", [(1, 5, 20, 20), (2, 3, 40, 24)])
def test_conv3d_pp_32_channel_slice(grid):
    """encoders.0 SingleConv2 of the bf16 engine: 32 REAL input channels read out of a 64-channel buffer (one 32-channel K chunk per depth slice) - the padding
    channels carry NaN here: the kernel must not touch them"""
    ops = _ops()
    N, D, H, W = grid
    Cin, Cout = 32, 64
    x = rnd(N, Cin, D, H, W, seed=160)
    w = rnd(Cout, Cin, 3, 3, 3, seed=161, scale=(27 * Cin) ** -0.5)
    wf = torch.empty(27, Cout, Cin, dtype=BF, device=DEV)
    ops.pack_conv_weight(w.to(DEV), wf, None)
    buf = torch.full((N, D, H, W, 64), float("nan"), dtype=BF, device=DEV)
    buf[..., :Cin] = to_nhwc(x, BF)
    y = torch.full((N, D, H, W, Cout), float("nan"), dtype=BF, device=DEV)
    ops.conv_igemm(ops.View(buf, 0, Cin), wf, y, ksize=3, Cin=Cin, Cout=Cout, grid=grid, relu=True)
    assert ops.conv_last_dispatch().startswith("k3.3d.ppc")
    assert_close(from_nhwc(y), F.relu(F.conv3d(q(x, BF), q(w, BF), None, padding=1)), "3-D conv, 32-channel slice", **tol(BF, 27 * Cin))


@pytest.mark.parametrize("grid,C0,C1,Cout", [((2, 6, 20, 36), 64, 128, 64), ((2, 4, 12, 32), 64, 0, 128), ((3, 5, 10, 18), 32, 0, 64)])
def test_gn_bwd_stats_from_per_sample_dw(grid, C0, C1, Cout):
    """mis_gn_bwd_stats_from_dw (round 3): S1 = sum_v dyn and S2 = sum_v dyn * x per (sample, channel) from the per-sample weight gradients of the ping-pong wgrad
    kernel and the border sums of g_y - against float64 sums over the exact (unrounded) dyn = conv_transpose(g_y, W), and against the statistics kernel that reads the
    bf16-stored dyn.  Two-source operand (encoder | nearest-upsampled), a single source, and the padded 32-of-64-channel case of encoders.0."""
    ops = _ops()
    N, D, H, W = grid
    Cin = C0 + C1
    Cp = (Cin + 63) // 64 * 64
    x0 = rnd(N, C0, D, H, W, seed=170)
    scale = 1 + 0.3 * rnd(N, Cp, seed=171)
    shift = 0.3 * rnd(N, Cp, seed=172)
    scale[:, Cin:] = 0
    shift[:, Cin:] = 0
    srcs = [q(x0, BF)]
    if C1:
        x1 = rnd(N, C1, D // 2, H // 2, W // 2, seed=173)
        srcs.append(F.interpolate(q(x1, BF), size=(D, H, W), mode="nearest"))
    xc = torch.cat(srcs, 1)
    w = rnd(Cout, Cin, 3, 3, 3, seed=174, scale=(27 * Cin) ** -0.5)
    wpad = torch.zeros(Cout, Cp, 3, 3, 3)
    wpad[:, :Cin] = w
    gy = rnd(N, Cout, D, H, W, seed=175)
    # device: xn by mis_gn_apply, per-sample wgrad, stats from dW
    xn = torch.zeros(N, D, H, W, Cp, dtype=BF, device=DEV)
    sc, sh = scale.to(DEV), shift.to(DEV)
    ops.gn_apply(to_nhwc(x0, BF), C0, False, grid, sc, sh, Cp, 0, xn)
    if C1:
        ops.gn_apply(to_nhwc(x1, BF), C1, True, grid, sc, sh, Cp, C0, xn)
    gyd = to_nhwc(gy, BF)
    dw = torch.empty(Cout, Cp, 3, 3, 3, device=DEV)
    dwn = torch.empty(N, Cout, Cp, 3, 3, 3, device=DEV)
    gys = torch.empty(N, Cout, device=DEV)
    ops.wgrad(xn, gyd, dw, ksize=3, Cin=Cp, Cout=Cout, grid=grid, dw_per_sample=dwn, dbias_per_sample=gys)
    assert ops.wgrad_last_dispatch()[0].startswith("k3.3d.pp")
    assert_close(dwn.sum(0), dw, "sum of the per-sample gradients", rtol=1e-5, atol=1e-5)
    assert_close(gys, q(gy, BF).sum((2, 3, 4)), "per-sample column sums of gy", rtol=1e-4, atol=2e-3)
    xnq = from_nhwc(xn)[:, :Cin]
    for n in range(N):          # each sample's gradient = the wgrad of that sample alone
        wz = torch.zeros(Cout, Cin, 3, 3, 3, requires_grad=True)
        F.conv3d(xnq[n:n + 1], wz, None, padding=1).backward(q(gy, BF)[n:n + 1])
        assert_close(dwn[n, :, :Cin], wz.grad, f"dW of sample {n}", rtol=1e-4, atol=1e-4 * (D * H * W) ** 0.5)
    G = 8
    mean = torch.zeros(N, G, device=DEV)                     # (only used for channels whose scale is 0)
    S1, S2 = torch.full((N, Cin), float("nan"), device=DEV), torch.full((N, Cin), float("nan"), device=DEV)
    ops.gn_bwd_stats_from_dw(gyd, wpad.to(DEV), dwn, gys, sc, sh, mean, G, Cin, S1, S2)
    # float64 reference on the exact dyn
    dyn = F.conv_transpose3d(q(gy, BF).double(), q(w, BF).double(), padding=1)
    r1, r2 = dyn.sum((2, 3, 4)), (dyn * xc.double()).sum((2, 3, 4))
    den = dyn.abs().sum((2, 3, 4)).clamp_min(1e-30)          # sums of signed terms: errors are judged against the sum of magnitudes
    e1 = ((S1.cpu().double() - r1).abs() / den).max().item()
    e2 = ((S2.cpu().double() - r2).abs() / (dyn.abs() * xc.double().abs()).sum((2, 3, 4)).clamp_min(1e-30)).max().item()
    assert e1 < 2e-4 and e2 < 2e-3, (e1, e2)                 # (S2 goes through xn, which is rounded to bf16: 2^-9 per element, averaged over the volume)
    # the statistics kernel on the bf16-stored dyn agrees to its own rounding
    wd = torch.empty(27, Cp, Cout, dtype=BF, device=DEV)
    wf = torch.empty(27, Cout, Cp, dtype=BF, device=DEV)
    ops.pack_conv_weight(wpad.to(DEV), wf, wd)
    dynd = torch.empty(N, D, H, W, Cp, dtype=BF, device=DEV)
    ops.conv_igemm(gyd, wd, dynd, ksize=3, Cin=Cout, Cout=Cp, grid=grid)
    K1, K2 = torch.zeros(N, Cin, device=DEV), torch.zeros(N, Cin, device=DEV)
    ops.gn_bwd_stats(dynd, to_nhwc(x0, BF), C0, False, grid, K1, K2, Cin, 0)
    if C1:
        ops.gn_bwd_stats(dynd, to_nhwc(x1, BF), C1, True, grid, K1, K2, Cin, C0)
    k1 = ((K1.cpu().double() - r1).abs() / den).max().item()
    print(f"S1 / S2 vs float64: from dW {e1:.2e} / {e2:.2e}; statistics kernel S1 {k1:.2e}")
    assert k1 < 2e-3


TQ_CASES = [("2d", (5, 150, 170), 128, 128, "k3.2d.ppc8"), ("2d", (32, 64, 64), 512, 512, "k3.2d.ppc8"), ("2d", (3, 512, 512), 64, 64, "k3.2d.ppd8"), ("2d", (2, 512, 512), 128, 64, "k3.2d.ppd8"),
            ("3d", (2, 16, 60, 80), 128, 128, "k3.3d.ppc5"), ("3d", (1, 24, 160, 160), 192, 64, "k3.3d.ppc10n2"), ("3d", (1, 7, 33, 21), 64, 192, "k3.3d.ppc5n6")]


@pytest.mark.parametrize("case", TQ_CASES, ids=lambda c: f"{c[0]}-{'x'.join(map(str, c[1]))}-{c[2]}to{c[3]}")
def test_tile_queue_is_bit_identical_to_the_static_stride_and_leaves_its_counters_at_zero(case):
    """round 5: the persistent conv kernels draw their tiles (after the first) from per-XCD ticket counters (csrc/conv_pp_common.hpp TileQ; MIS_TILEQ_OFF=1 = the static
    stride).  Same tiles, same arithmetic per tile: the outputs are bit-identical, with and without the epilogue mask, over several launches in a row - which also shows
    that the kernel that draws a counter's last ticket has put it back to zero (a dirty counter would skip tiles: NaNs from the fill would remain)"""
    import ctypes
    ops = _ops()
    kind, grid, Cin, Cout, tag = case
    taps = 9 if kind == "2d" else 27
    gen = torch.Generator(device=DEV).manual_seed(77)
    x = torch.randn(*grid, Cin, device=DEV, generator=gen).to(BF)
    w = (torch.randn(taps, Cout, Cin, device=DEV, generator=gen) * (taps * Cin) ** -0.5).to(BF)
    m = torch.randn(*grid, Cout, device=DEV, generator=gen).to(BF)
    lib = ops.load()
    out8 = (ctypes.c_uint * 8)()
    for kw in (dict(relu=True), dict(mask=m)):
        if kind == "3d":
            kw = dict(kw, grid=grid)
        ref = torch.full((*grid, Cout), float("nan"), dtype=BF, device=DEV)
        with ops.dispatch_switches(MIS_TILEQ_OFF=1):
            ops.conv_igemm(x, w, ref, ksize=3, Cin=Cin, Cout=Cout, **kw)
            assert ops.conv_last_dispatch().startswith(tag), ops.conv_last_dispatch()
        assert not torch.isnan(ref.float()).any()
        for launch in range(3):
            y = torch.full((*grid, Cout), float("nan"), dtype=BF, device=DEV)
            ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, **kw)
            assert ops.conv_last_dispatch().startswith(tag)
            assert torch.equal(y, ref), (launch, int(torch.isnan(y.float()).sum()))
            assert lib.mis_debug_tile_queue(ops.stream_ptr(), out8) == 0 and list(out8) == [0] * 8, (launch, list(out8))
    ops.tile_queue_check()          # no launch drew a ticket past its last one


def test_tile_queue_reports_a_launch_that_started_on_dirty_counters():
    """a ticket past the launch's last one used to end the block silently - output tiles unwritten, nothing said.  A counter that does not start at zero
    (planted here with mis_debug_tile_queue_poke: what an unfinished launch leaves) now makes the launch LOUD: the kernel records the ticket, ops.tile_queue_check() raises;
    ops.tile_queue_reset() (the engines' first call of every step) restores the block and the next launch is bit-identical to the reference again."""
    from mdeical_image_segmentation_amd._lib import MisError
    ops = _ops()
    lib = ops.load()
    N, H, W, Cin, Cout = 5, 150, 170, 128, 128          # 275 tiles > 256 persistent blocks: tiles are drawn from the queue
    x = to_nhwc(rnd(N, Cin, H, W, seed=170), BF)
    w = rnd(Cout, Cin, 3, 3, seed=171, scale=(9 * Cin) ** -0.5)
    wf = torch.empty(9, Cout, Cin, dtype=BF, device=DEV)
    ops.pack_conv_weight(w.to(DEV), wf, None)
    ref = torch.full((N, H, W, Cout), float("nan"), dtype=BF, device=DEV)
    ops.tile_queue_reset()
    ops.conv_igemm(x, wf, ref, ksize=3, Cin=Cin, Cout=Cout, relu=True)
    assert ops.conv_last_dispatch().startswith("k3.2d.ppc8")
    ops.tile_queue_check()
    for xcd in range(8):
        assert lib.mis_debug_tile_queue_poke(ops.stream_ptr(), xcd, 1000) == 0
    y = torch.full_like(ref, float("nan"))
    ops.conv_igemm(x, wf, y, ksize=3, Cin=Cin, Cout=Cout, relu=True)
    with pytest.raises(MisError, match="ticket past"):
        ops.tile_queue_check()
    assert torch.isnan(y.float()).any(), "the planted counters should have cost this launch tiles"
    ops.tile_queue_check()          # the report also cleared the error words
    ops.tile_queue_reset()
    y.fill_(float("nan"))
    ops.conv_igemm(x, wf, y, ksize=3, Cin=Cin, Cout=Cout, relu=True)
    assert torch.equal(y, ref)
    ops.tile_queue_check()


def test_tile_queue_gives_each_captured_graph_its_own_counter_block():
    """a captured launch bakes in a counter block; all graphs used to share the block of torch's capture stream.  Now a launch under capture takes the block of its
    CAPTURE (hipStreamGetCaptureInfo id): two graphs hold different blocks, replays stay bit-identical and leave every counter at zero."""
    ops = _ops()
    N, H, W, Cin, Cout = 5, 150, 170, 128, 128
    x = to_nhwc(rnd(N, Cin, H, W, seed=172), BF)
    w = rnd(Cout, Cin, 3, 3, seed=173, scale=(9 * Cin) ** -0.5)
    wf = torch.empty(9, Cout, Cin, dtype=BF, device=DEV)
    ops.pack_conv_weight(w.to(DEV), wf, None)
    ref = torch.empty(N, H, W, Cout, dtype=BF, device=DEV)
    ops.conv_igemm(x, wf, ref, ksize=3, Cin=Cin, Cout=Cout, relu=True)
    torch.cuda.synchronize()
    ys = [torch.full_like(ref, float("nan")) for _ in range(2)]
    graphs = []
    for y in ys:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ops.tile_queue_reset()
            ops.conv_igemm(x, wf, y, ksize=3, Cin=Cin, Cout=Cout, relu=True)
        graphs.append(g)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        with torch.cuda.stream(s1):
            graphs[0].replay()
        with torch.cuda.stream(s2):
            graphs[1].replay()
    torch.cuda.synchronize()
    assert torch.equal(ys[0], ref) and torch.equal(ys[1], ref)
    ops.tile_queue_check()


def test_tile_queue_keeps_a_conv_launch_from_doubling_beside_a_kernel_that_holds_cus():
    """what the queue is for: an RCCL kernel on the side stream takes CUs and keeps them (nothing of ours fits beside another workgroup).  With the static stride the 8 blocks
    that find no CU start when the first ones END (+40-63 % measured, scripts/hog_probe.sh); with the queue the running blocks share the tiles (+0-17 %).  The measured
    ratios are printed (and warned about beyond 1.35 x), not asserted; the CU holder is scripts/cu_hog.hip, built here into a scratch library - it is a
    diagnostic, not part of libmisamd.so (round 6)."""
    import ctypes
    import os
    import shutil
    import subprocess
    import tempfile
    ops = _ops()
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box: the CU-holder diagnostic kernel cannot be built")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(tempfile.mkdtemp(prefix="cuhog"), "libcuhog.so")
    subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, os.path.join(root, "scripts", "cu_hog.hip")], check=True, capture_output=True)
    hog = ctypes.CDLL(so)
    hog.cu_hog.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
    sink = torch.zeros(4, dtype=torch.int32, device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(5)
    N, H, W, Cin, Cout = 32, 64, 64, 512, 512
    x = torch.randn(N, H, W, Cin, device=DEV, generator=gen).to(BF)
    w = (torch.randn(9, Cout, Cin, device=DEV, generator=gen) * (9 * Cin) ** -0.5).to(BF)
    y = torch.empty(N, H, W, Cout, dtype=BF, device=DEV)
    side = torch.cuda.Stream()

    def timed(hold):
        for _ in range(2):
            ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, relu=True)
        torch.cuda.synchronize()
        if hold:
            assert hog.cu_hog(hold, 30_000_000, sink.data_ptr(), side.cuda_stream) == 0          # ~15 ms at 2 GHz: longer than the six launches below
            torch.cuda._sleep(2_000_000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, relu=True)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 6

    alone = min(timed(0) for _ in range(2))
    beside = min(timed(8) for _ in range(2))
    assert ops.conv_last_dispatch().startswith("k3.2d.ppc8")
    with ops.dispatch_switches(MIS_TILEQ_OFF=1):
        static_beside = timed(8)
    print(f"conv 512->512 at 64^2: alone {alone:.3f} ms, beside a holder of 8 CUs {beside:.3f} ms (static stride: {static_beside:.3f} ms)")
    # a wall-clock ratio is not a gate (a shared or throttled box fails it without a code defect) - the measured ratios are printed (profiles/r0N_hog_probe.txt
    # keeps them per round); what is asserted is what the test can know: the launches beside the holder ran the queue's kernel and left clean counters
    if beside >= 1.35 * alone:
        import warnings
        warnings.warn(f"tile queue beside a holder of 8 CUs: {beside / alone:.2f} x its time alone (expected < 1.35 x; static stride {static_beside / alone:.2f} x)")
    ops.tile_queue_check()


@pytest.mark.parametrize("case", [((2, 6, 40, 48), 128, 128), ((1, 5, 80, 36), 384, 128), ((3, 3, 33, 21), 64, 256), ((1, 2, 40, 16), 256, 512)],
                         ids=lambda c: f"{'x'.join(map(str, c[0]))}-{c[1]}to{c[2]}")
def test_conv3d_streamed_40_row_tile_is_bit_identical_to_the_20_row_tile(case):
    """round 5: conv3d_ppc_kernel<10, 4> (40 x 16-pixel tiles x 128 columns, the pixel-row fragments streamed through a three-slot rotation inside the M segment; plain
    epilogue only) adds the taps of every accumulator in the order of the 20-row tile: bit-identical, ragged grids and several column tiles included; a masked launch of the
    same layer falls back to the 20- / 32-row tiles"""
    ops = _ops()
    grid, Cin, Cout = case
    gen = torch.Generator(device=DEV).manual_seed(91)
    x = torch.randn(*grid, Cin, device=DEV, generator=gen).to(BF)
    w = (torch.randn(27, Cout, Cin, device=DEV, generator=gen) * (27 * Cin) ** -0.5).to(BF)
    outs = []
    for pf, tag in ((5, "k3.3d.ppc5"), (10, "k3.3d.ppc10")):
        y = torch.full((*grid, Cout), float("nan"), dtype=BF, device=DEV)
        with ops.dispatch_switches(MIS_CONV3D_PF=pf):
            ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, grid=grid, relu=True)
            assert ops.conv_last_dispatch() == tag, ops.conv_last_dispatch()
        outs.append(y)
    assert not torch.isnan(outs[0].float()).any() and torch.equal(outs[0], outs[1])
    m = torch.randn(*grid, Cout, device=DEV, generator=gen).to(BF)
    with ops.dispatch_switches(MIS_CONV3D_PF=10):
        ops.conv_igemm(x, w, outs[1], ksize=3, Cin=Cin, Cout=Cout, grid=grid, mask=m)
        assert ops.conv_last_dispatch() in ("k3.3d.ppc5.mask", "k3.3d.ppc8.mask"), ops.conv_last_dispatch()
