"""End-to-end parity of the fused 2-D engine (HIP, through the C ABI) against the golden vectors captured from the
reference (tests/golden/g2_unet_*.npz) and against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): fp32 mode - argmax masks bit-exact, loss within 1e-4; bf16 mode - reported, looser."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.asarray(a))


def stat(t):
    t = t.detach().double().cpu().flatten()
    idx = torch.linspace(0, t.numel() - 1, steps=64).long()
    return np.concatenate([[t.sum().item(), t.abs().sum().item(), (t * t).sum().item()], t[idx].numpy()])


def _engine(cin, cout, dtype):
    from mdeical_image_segmentation_amd.engine2d import UNet2DEngine
    return UNet2DEngine(cin, cout, dtype=dtype, device=DEV, seed=0)


def near_tie_mask(logits, eps=1e-4):
    if logits.shape[1] == 1:
        return logits[:, 0].abs() < eps
    top2 = logits.topk(2, dim=1).values
    return (top2[:, 0] - top2[:, 1]) < eps


@pytest.mark.parametrize("tag,cin,cout", [("1_2", 1, 2), ("3_4", 3, 4), ("1_1", 1, 1)])
def test_fp32_engine_matches_reference_goldens(tag, cin, cout):
    g = load_golden(f"g2_unet_{tag}.npz")
    eng = _engine(cin, cout, torch.float32)
    names = [str(n) for n in g["names"]]
    # same seed => same parameters as the reference's `torch.manual_seed(0); UNet(cin, cout)`
    ps = np.stack([stat(eng.P[n]) for n in names])
    # sampled values bit-identical; the double sums only up to summation order (thread count differs per box)
    assert np.array_equal(ps[:, 3:], g["param_stats"][:, 3:]), "seeded init differs from the reference"
    assert np.allclose(ps[:, :3], g["param_stats"][:, :3], rtol=1e-10, atol=1e-10), "seeded init differs from the reference"
    images = T(g["images"]).to(DEV)
    labels = T(g["labels"]).to(DEV)
    loss, logits, am = eng.forward(images, labels, train=True)
    ref_logits = T(g["logits"])
    d = (logits.cpu() - ref_logits).abs().max().item()
    assert d < 1e-4, f"logits max|diff| {d}"
    assert abs(loss.item() - float(g["loss"])) < 1e-4, (loss.item(), float(g["loss"]))
    ref_am = T(g["argmax"])
    ref_am = ref_am if ref_am.dim() == 3 else ref_am[:, 0]
    nt = near_tie_mask(ref_logits)
    # the bar is BIT-EXACT argmax on the goldens: no pixel may be excused as a near-tie here (the mask only serves the non-golden shapes below)
    assert int(nt.sum()) == 0, f"{int(nt.sum())} golden pixels lie within 1e-4 of a tie: regenerate the golden on structured inputs"
    flips = int((am.cpu().long() != ref_am).sum())
    assert flips == 0, f"{flips} argmax flips against the reference"
    assert torch.equal(am.cpu().long(), (logits.cpu().argmax(1) if cout > 1 else (logits.cpu()[:, 0] > 0).long()))
    print(f"[{tag}] logits max|diff| {d:.3g}; near-tie pixels excluded: {int(nt.sum())} of {nt.numel()}; "
          f"flips incl. near-ties: {int((am.cpu().long() != ref_am).sum())}")
    eng.backward()
    torch.cuda.synchronize()
    gs = np.stack([stat(eng.G[n]) for n in names])
    ref = g["grad_stats"]
    for i, n in enumerate(names):
        scale = max(1e-6, abs(ref[i, 1]) / max(1, eng.G[n].numel()))       # mean |g|
        assert abs(gs[i, 0] - ref[i, 0]) <= 2e-3 * abs(ref[i, 1]) + 1e-6, (n, "sum", gs[i, 0], ref[i, 0])
        assert abs(gs[i, 1] - ref[i, 1]) <= 2e-3 * abs(ref[i, 1]) + 1e-6, (n, "abssum", gs[i, 1], ref[i, 1])
        assert np.allclose(gs[i, 3:], ref[i, 3:], rtol=5e-3, atol=20 * scale * 1e-2), (n, "samples")
    assert torch.allclose(eng.G["final_conv.weight"].cpu(), T(g["g_final_w"]), rtol=1e-3, atol=1e-6)
    assert torch.allclose(eng.G["down_conv.0.first.weight"].cpu(), T(g["g_down0_first_w"]), rtol=2e-3, atol=1e-5)
    assert torch.allclose(eng.G["up_sample.0.up.bias"].cpu(), T(g["g_up0_b"]), rtol=2e-3, atol=1e-6)
    # three optimizer steps like the reference's Trainer inner loop (clip 1.0, AdamW, decay split)
    eng.optimizer_step()
    assert abs(eng.gradnorm.item() - g["step_gradnorms"][0]) < 2e-3 * g["step_gradnorms"][0]
    for step in (1, 2):
        loss = eng.train_step(images, labels)
        assert abs(loss.item() - g["step_losses"][step]) < 1e-4 * max(1.0, g["step_losses"][step]), (step, loss.item())
        assert abs(eng.gradnorm.item() - g["step_gradnorms"][step]) < 5e-3 * g["step_gradnorms"][step]
    st = np.stack([stat(eng.P[n]) for n in names])
    ref = g["param_stats_step3"]
    assert np.allclose(st[:, :3], ref[:, :3], rtol=5e-3, atol=1e-4), "parameters after 3 steps"


# (net, fixture, input, FIXED bar against the bf16-storage oracle).  Round 6: no data-derived bars.  The 3 -> 4 net runs on the reference golden's OWN
# input again (one 32 x 48 image: 24 pixels at the deepest level, where one ReLU flip moves down_conv.3.first.weight / middle_conv.first.weight by several percent - measured
# worst 0.071 / 0.062 on two builds, hence 0.10) AND on a batch of the 1 -> 2 fixture's shape from the pinned oracle (0.08); the tight statement for the four-class fused
# head is test_fused_head_kernel_against_fp64_on_the_same_bf16_operands below.
BF16_E2E_CASES = [(1, 2, "g2_unet_1_2.npz", "golden", 6e-2), (3, 4, "g2_unet_3_4.npz", "golden", 1e-1), (3, 4, "g2_unet_3_4.npz", "oracle-2x64x64", 8e-2)]


@pytest.mark.parametrize("net", BF16_E2E_CASES, ids=["1x2", "3x4-golden-input", "3x4-oracle-batch"])
def test_bf16_engine_close_to_oracle(net):
    """bf16 storage / fp32 accumulate.  (round 5: also UNet(3, 4), cfg3's net - its four-class head runs conv_ppd_head_kernel<4>.)  Two bars: (a) per gradient tensor,
    against the oracle with bf16 storage emulated at the same tensor boundaries (oracle.unet2d_oracle.loss_and_grads_bf16_storage) - the parity statement for the bf16
    kernels end to end, a FIXED number per case; (b) against the fp32 reference, where bf16 storage itself costs 3-12 % relative L2 on this net (the emulation shows it
    without any device code)."""
    from oracle import unet2d_oracle as o2
    cin, cout, fixture, which, bar_emu = net
    g = load_golden(fixture)
    eng = _engine(cin, cout, torch.bfloat16)
    p = o2.init_params(cin, cout, seed=0)
    if which == "golden":
        im_c, lb_c = T(g["images"]), T(g["labels"])
        ref_logits, ref_loss = T(g["logits"]), float(g["loss"])
    else:
        gen = torch.Generator().manual_seed(34)
        im_c, lb_c = torch.randn(2, 3, 64, 64, generator=gen), torch.randint(0, 4, (2, 64, 64), generator=gen)
        rl_, ref_logits, _ = o2.loss_and_grads(p, im_c, lb_c)
        ref_loss = rl_.item()
    images, labels = im_c.to(DEV), lb_c.to(DEV)
    loss, logits, am = eng.forward(images, labels, train=True)
    assert eng.features_valid is False, "the fused head (conv_ppd_head_kernel) did not run: this test is its end-to-end case against the oracle"
    d = (logits.cpu() - ref_logits).abs().max().item()
    rel = d / ref_logits.abs().max().item()
    print(f"bf16: logits max|diff| {d:.3g} (rel {rel:.3g}), loss {loss.item():.5f} vs {ref_loss:.5f}")
    assert rel < 0.01
    assert abs(loss.item() - ref_loss) < 1e-3
    eng.backward()
    _, _, g32 = o2.loss_and_grads(p, im_c, lb_c)
    el, elogits, g16 = o2.loss_and_grads_bf16_storage(p, im_c, lb_c)
    assert (logits.cpu() - elogits).abs().max().item() < 2e-5 + 2e-3 * elogits.abs().max().item()
    assert abs(loss.item() - el.item()) < 1e-4
    worst_emu, worst_f32, storage = ("", 0.0), ("", 0.0), 0.0
    for n in g32:          # all 46 tensors
        a, b, c = eng.G[n].cpu().flatten().double(), g16[n].flatten().double(), g32[n].flatten().double()
        r_emu = ((a - b).norm() / (b.norm() + 1e-30)).item()
        r_f32 = ((a - c).norm() / (c.norm() + 1e-30)).item()
        storage = max(storage, ((b - c).norm() / (c.norm() + 1e-30)).item())
        worst_emu = max(worst_emu, (n, r_emu), key=lambda t: t[1])
        worst_f32 = max(worst_f32, (n, r_f32), key=lambda t: t[1])
        # measured worst 2.6e-2 ... 4.3e-2 depending on the build's summation order: every differently rounded activation perturbs the ReLU masks downstream,
        # so two correct bf16-storage pipelines agree to a few percent here while bf16 storage itself costs up to 12 %; a wrong tap / slice would be O(1)
    # (the worst tensors are printed before anything is asserted: a failure names its tensor AND shows the rest)
    print(f"bf16 {cin}->{cout} ({which}): worst gradient rel-L2 vs the bf16-storage oracle {worst_emu[1]:.3g} ({worst_emu[0]}; fixed bar {bar_emu}), vs the fp32 oracle "
          f"{worst_f32[1]:.3g} ({worst_f32[0]}); bf16-storage oracle vs fp32 oracle (no device code) up to {storage:.3g}")
    assert worst_emu[1] <= bar_emu, (worst_emu, "vs bf16-storage oracle", bar_emu)
    assert worst_f32[1] <= 0.16, (worst_f32, "vs fp32 oracle")


def test_larger_batch_vs_oracle_fp32():
    """A second, non-golden shape (3 x 64 x 48, ragged against the 8x16 / 16x16 tiles) against the CPU oracle."""
    from oracle import unet2d_oracle as o2
    eng = _engine(1, 2, torch.float32)
    gen = torch.Generator().manual_seed(5)
    images = torch.randn(3, 1, 64, 48, generator=gen)
    labels = torch.randint(0, 2, (3, 64, 48), generator=gen)
    p = o2.init_params(1, 2, seed=0)
    rl, rlogits, rgrads = o2.loss_and_grads(p, images, labels)
    loss, logits, am = eng.forward(images.to(DEV), labels.to(DEV), train=True)
    eng.backward()
    assert (logits.cpu() - rlogits).abs().max().item() < 1e-4
    assert abs(loss.item() - rl.item()) < 1e-4
    nt = near_tie_mask(rlogits)
    assert int((am.cpu().long() != rlogits.argmax(1))[~nt].sum()) == 0
    for n, gref in rgrads.items():
        a = eng.G[n].cpu()
        err = (a - gref).abs().max().item()
        assert err <= 2e-3 * gref.abs().max().item() + 1e-7, (n, err, gref.abs().max().item())


def test_persistent_grids_leave_cus_free(switches):
    """MIS_PERSIST_CUS (csrc/dispatch_cfg.hpp): the persistent kernels launch at most that many blocks, so that a concurrent RCCL kernel finds free CUs (the A/B switch of
    the first multi-GPU run).  A bf16 step at 8 x 128 x 128 (up to 1024 tiles per launch: every persistent grid clips) with 200 instead of 256 blocks:
    the convolutions walk the same tiles in another order - logits and loss bit-identical; the weight gradients split K differently - equal to fp32 summation order."""
    gen = torch.Generator().manual_seed(11)
    images = torch.randn(8, 1, 128, 128, generator=gen).to(DEV)
    labels = torch.randint(0, 2, (8, 128, 128), generator=gen).to(DEV)

    def run():
        eng = _engine(1, 2, torch.bfloat16)
        loss, logits, _ = eng.forward(images, labels, train=True)
        eng.backward()
        torch.cuda.synchronize()
        return loss.clone(), logits.clone(), eng.flat.g.clone()

    l0, lg0, g0 = run()
    switches("MIS_PERSIST_CUS", 200)
    l1, lg1, g1 = run()
    assert torch.equal(lg0, lg1) and torch.equal(l0, l1)
    rel = ((g0.double() - g1.double()).norm() / g0.double().norm()).item()
    assert rel < 1e-5, rel


@pytest.mark.parametrize("dtype,shape", [(torch.bfloat16, (8, 128, 128)), (torch.float32, (2, 48, 64))])
def test_batched_slab_reduction_is_bit_identical(dtype, shape):
    """mis_wgrad_reduce_batch (MisWgradDesc.defer: the slab reductions of a stage's layers as two launches) against the three per-layer kernels: the same arithmetic in the
    same order - every gradient bit-identical (3x3 layers with 2 ... 256 slabs, the transposed convolutions' layout-1 gradients with folded bias sums, both precisions)."""
    N, H, W = shape
    gen = torch.Generator().manual_seed(12)
    images = torch.randn(N, 1, H, W, generator=gen).to(DEV)
    labels = torch.randint(0, 2, (N, H, W), generator=gen).to(DEV)
    out = []
    for batch in (True, False):
        eng = _engine(1, 2, dtype)
        assert eng.batch_reduce
        eng.batch_reduce = batch
        eng.forward(images, labels, train=True)
        eng.backward()
        torch.cuda.synchronize()
        assert not eng._red
        out.append(eng.flat.g.clone())
    assert torch.equal(out[0], out[1])
    assert float(out[0].abs().sum()) > 0


@pytest.mark.parametrize("cout,shape", [(2, (3, 64, 48)), (2, (2, 96, 80)), (1, (2, 64, 64)), (2, (9, 160, 176)), (4, (2, 64, 80)), (3, (2, 96, 48)), (4, (9, 160, 176))])
def test_fused_head_matches_the_separate_kernels(cout, shape, monkeypatch):
    """csrc/conv_ppd_head.hip (up_conv.3.second + final_conv + CE / BCE + their backward in ONE kernel; the last feature map is never written) against the separate
    convolution + mis_head_loss: same loss, logits, arg-max and gradients up to fp32 summation order (the logits come off the matrix pipe with Wh split into bf16 hi + lo
    parts) and the bf16 rounding flips that follow from it.  Ragged tiles (48 / 80 / 176 columns), 1 .. 4 classes, several tiles per persistent block (9 x 5 x 11 = 495)."""
    from mdeical_image_segmentation_amd import ops
    N, H, W = shape
    gen = torch.Generator().manual_seed(31)
    images = torch.randn(N, 1, H, W, generator=gen).to(DEV)
    labels = (torch.randint(0, cout, (N, H, W), generator=gen) if cout >= 2 else torch.randint(0, 2, (N, 1, H, W), generator=gen).float()).to(DEV)
    res = []
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("MISAMD_HEAD_UNFUSED", "1")
        eng = _engine(1, cout, torch.bfloat16)
        loss, logits, am = eng.forward(images, labels, train=True, grad_scale=0.5)
        assert eng.features_valid == (not fused), "the case did not take the path it is meant for"
        eng.backward()
        torch.cuda.synchronize()
        res.append((loss.clone(), logits.clone(), am.clone(), eng.flat.g.clone(), {k: eng.G[k].clone() for k in ("final_conv.weight", "final_conv.bias", "up_conv.3.second.weight", "down_conv.0.first.weight")}))
    (l0, lg0, am0, g0, d0), (l1, lg1, am1, g1, d1) = res
    assert abs(l0.item() - l1.item()) < 2e-6, (l0.item(), l1.item())
    assert (lg0 - lg1).abs().max().item() <= 2e-5 * lg1.abs().max().item() + 1e-6
    top = lg1 if cout == 1 else lg1.topk(2, dim=1).values
    near = (lg1.abs()[:, 0] < 1e-4) if cout == 1 else ((top[:, 0] - top[:, 1]) < 1e-4)
    assert torch.equal(am0[~near], am1[~near])
    for k in d0:
        r = ((d0[k].double() - d1[k].double()).norm() / (d1[k].double().norm() + 1e-30)).item()
        assert r < (2e-5 if k.startswith("final_conv") else 3e-3), (k, r)
    r = ((g0.double() - g1.double()).norm() / g1.double().norm()).item()
    assert r < 3e-3, r


@pytest.mark.parametrize("C", [1, 2, 3, 4])
@pytest.mark.parametrize("shape", [(2, 64, 80, 64), (3, 96, 48, 128)], ids=lambda s: "x".join(map(str, s)))
def test_fused_head_kernel_against_fp64_on_the_same_bf16_operands(C, shape):
    """conv_ppd_head_kernel<C> on its own: bf16 input features / bf16-rounded weights / labels in, through mis_conv3x3_head_fused, against an fp64 evaluation
    of THE SAME operands - Conv2d(Cin, 64, 3, p1) + bias + ReLU (reference layers.py:122-126), the features rounded to bf16 as the unfused pipeline stores them, final_conv
    (unet.py:89), cross entropy / BCE-with-logits (unet.py:1184-1188), arg-max, dL/dfeatures, head dW / db.  Fixed numeric bars: logits / loss / dW / db <= 2e-3 relative L2
    (measured ~1e-5: Wh travels through the matrix pipe as bf16 hi + lo parts), dL/dfeatures <= 2e-3 (stored in bf16: its rounding is ~1e-3), arg-max exact away from
    1e-4 ties."""
    import torch.nn.functional as F
    from mdeical_image_segmentation_amd import ops
    N, H, W, Cin = shape
    gen = torch.Generator().manual_seed(40 + C)
    bf = torch.bfloat16
    x = (torch.randn(N, Cin, H, W, generator=gen).clamp_min(0)).to(bf)                     # a ReLU output, as up_conv.3.first leaves it
    w = (torch.randn(64, Cin, 3, 3, generator=gen) * (9 * Cin) ** -0.5)
    b = torch.randn(64, generator=gen) * 0.1
    wh = torch.randn(C, 64, generator=gen) * 0.2
    bh = torch.randn(C, generator=gen) * 0.1
    labels = torch.randint(0, C, (N, H, W), generator=gen) if C > 1 else torch.randint(0, 2, (N, 1, H, W), generator=gen).float()
    gs = 0.37
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wf = torch.empty(9, 64, Cin, dtype=bf, device=DEV)
    ops.pack_conv_weight(w.to(DEV), wf, None)
    dy = torch.full((N, H, W, 64), float("nan"), dtype=bf, device=DEV)
    logits = torch.full((N, C, H, W), float("nan"), device=DEV)
    am = torch.full((N, H, W), 255, dtype=torch.uint8, device=DEV)
    loss_out = torch.zeros(16, device=DEV)
    dw = torch.full((C, 64), float("nan"), device=DEV)
    db = torch.full((C,), float("nan"), device=DEV)
    ok = ops.conv3x3_head_fused(xd, wf, b.to(DEV), dy, wh.to(DEV), bh.to(DEV), Cin=Cin, loss=ops.LOSS_CE if C > 1 else ops.LOSS_BCE, labels=labels.to(DEV), logits=logits,
                                argmax=am, loss_out=loss_out, dw=dw, db=db, grad_scale=gs)
    assert ok, "the fused head refused an eligible configuration"
    torch.cuda.synchronize()
    # ---- fp64 on the same operands ----
    pre = F.conv2d(x.double(), w.to(bf).double(), b.double(), padding=1)
    f = pre.clamp_min(0).float().to(bf).double().requires_grad_(True)                      # the feature map as bf16 storage holds it
    whd, bhd = wh.double().requires_grad_(True), bh.double().requires_grad_(True)
    lg = torch.einsum("nkhw,ck->nchw", f, whd) + bhd.view(1, C, 1, 1)
    ref_loss = F.cross_entropy(lg, labels) if C > 1 else F.binary_cross_entropy_with_logits(lg, labels.double())
    (ref_loss * gs).backward()
    rel = lambda a, r: ((a.double().cpu() - r).norm() / (r.norm() + 1e-300)).item()
    r_lg, r_dw, r_db = rel(logits, lg.detach()), rel(dw, whd.grad), rel(db, bhd.grad)
    r_df = rel(dy.permute(0, 3, 1, 2), f.grad * (f.detach() > 0))
    print(f"fused head C={C}: logits {r_lg:.2e}, loss {abs(loss_out[0].item() - ref_loss.item()):.2e}, dfeatures {r_df:.2e}, dW {r_dw:.2e}, db {r_db:.2e}")
    assert r_lg <= 2e-3 and r_dw <= 2e-3 and r_db <= 2e-3 and r_df <= 2e-3, (r_lg, r_df, r_dw, r_db)
    assert abs(loss_out[0].item() - ref_loss.item()) <= 1e-4 * max(1.0, abs(ref_loss.item()))
    if C > 1:
        top = lg.detach().topk(2, dim=1).values
        near = (top[:, 0] - top[:, 1]) < 1e-4
        want = lg.detach().argmax(1)
    else:
        near = lg.detach()[:, 0].abs() < 1e-4
        want = (lg.detach()[:, 0] > 0).long()
    assert torch.equal(am.cpu().long()[~near], want[~near])
    assert int((~near).sum()) > 0.99 * near.numel()


def test_fused_head_then_external_gradient_through_logits():
    """the rare autograd path after a FUSED forward: a gradient arriving through `logits` needs the last feature map, which the fused kernel never wrote - head_backward
    regenerates it (one convolution) before the external-gradient head pass"""
    gen = torch.Generator().manual_seed(32)
    images = torch.randn(2, 1, 64, 64, generator=gen).to(DEV)
    labels = torch.randint(0, 2, (2, 64, 64), generator=gen).to(DEV)
    dlog = torch.randn(2, 2, 64, 64, generator=gen).to(DEV) * 1e-3
    out = []
    for fused in (True, False):
        eng = _engine(1, 2, torch.bfloat16)
        if not fused:
            eng._fused_head = lambda *a, **k: False
        eng.forward(images, labels, train=True)
        assert eng.features_valid == (not fused)
        eng.head_backward(dlog)
        eng.backward()
        torch.cuda.synchronize()
        out.append(eng.flat.g.clone())
    assert torch.equal(out[0], out[1])
