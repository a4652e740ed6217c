"""Size-independent properties of the hot path at BASELINE.json's full sizes (configs[1]: bs 32, 512 x 512, bf16; cfg4: 2 x 128^3 fp32), where the
CPU oracle would take minutes:
  * run-to-run determinism (fixed-order split-K reductions, side-stream reductions included): loss, logits and every gradient bit-identical;
  * batch independence: the U-Net has no cross-sample coupling, so logits / arg-max of a sample do not depend on what else is in the batch, the batch
    loss is the mean of the chunk losses and the batch gradient the mean of the chunk gradients;
  * 3-D (GroupNorm, per-sample statistics): sample 0 of a batch of two equals the same volume alone."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_2d_full_size_determinism_and_batch_additivity():
    from mdeical_image_segmentation_amd.engine2d import UNet2DEngine
    B, S = 32, 512
    eng = UNet2DEngine(1, 2, dtype=torch.bfloat16, device=DEV, seed=0)
    gen = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn(B, 1, S, S, device=DEV, generator=gen)
    y = torch.randint(0, 2, (B, S, S), device=DEV, generator=gen)

    def run(xb, yb):
        loss, logits, am = eng.forward(xb, yb, train=True)
        eng.backward()
        torch.cuda.synchronize()
        return loss.clone(), logits.clone(), am.clone(), eng.flat.g.clone()

    l1, lg1, am1, g1 = run(x, y)
    l2, lg2, am2, g2 = run(x, y)
    assert torch.equal(l1, l2) and torch.equal(lg1, lg2) and torch.equal(am1, am2) and torch.equal(g1, g2), "a step is not bit-reproducible"
    assert torch.isfinite(g1).all() and g1.abs().sum().item() > 0
    # arg-max is the arg-max of the returned logits (first maximum)
    assert torch.equal(am1.long(), lg1.argmax(1))
    # chunks of 8 samples
    losses, gsum = [], torch.zeros_like(g1)
    for c in range(4):
        sl = slice(8 * c, 8 * c + 8)
        lc, lgc, amc, gc = run(x[sl].contiguous(), y[sl].contiguous())
        assert torch.equal(lgc, lg1[sl]) and torch.equal(amc, am1[sl]), "a sample's logits depend on the batch around it"
        losses.append(lc.item())
        gsum += gc
    assert abs(sum(losses) / 4 - l1.item()) < 1e-5
    rel = ((gsum / 4 - g1).norm() / g1.norm()).item()
    assert rel < 2e-3, rel          # bf16 operands, fp32 accumulation in a batch-size dependent split-K order


def test_3d_full_size_determinism_and_sample_independence():
    from mdeical_image_segmentation_amd.engine3d import UNet3DEngine
    eng = UNet3DEngine(1, 3, dtype=torch.float32, device=DEV, seed=0)
    gen = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(2, 1, 128, 128, 128, device=DEV, generator=gen)
    t = (torch.rand(2, 3, 128, 128, 128, device=DEV, generator=gen) > 0.5).float()

    def run(xb, tb):
        loss, logits, am = eng.forward(xb, tb, train=True)
        eng.backward()
        torch.cuda.synchronize()
        return loss.clone(), logits.clone(), am.clone(), eng.flat.g.clone()

    l1, lg1, am1, g1 = run(x, t)
    l2, lg2, am2, g2 = run(x, t)
    assert torch.equal(l1, l2) and torch.equal(lg1, lg2) and torch.equal(am1, am2) and torch.equal(g1, g2), "a step is not bit-reproducible"
    assert torch.isfinite(g1).all()
    _, lg0, am0, _ = run(x[:1].contiguous(), t[:1].contiguous())
    assert torch.equal(lg0, lg1[:1]) and torch.equal(am0, am1[:1]), "GroupNorm statistics leak across samples"


def test_3d_bf16_cfg5_shape_determinism_and_sample_independence():
    """cfg5's per-GPU shape (2 x 160^3, bf16): the 3-D ping-pong convolutions and the streaming / tile-staged weight gradients (level 0: W = 160 = 5 strips of 32; the
    deeper levels' W is not a multiple of 32) - a step is bit-reproducible, and sample 0's logits do not depend on sample 1."""
    from mdeical_image_segmentation_amd import ops
    from mdeical_image_segmentation_amd.engine3d import UNet3DEngine
    eng = UNet3DEngine(1, 3, dtype=torch.bfloat16, device=DEV, seed=0)
    gen = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(2, 1, 160, 160, 160, device=DEV, generator=gen)
    t = (torch.rand(2, 3, 160, 160, 160, device=DEV, generator=gen) > 0.5).float()

    def run(xb, tb):
        loss, logits, am = eng.forward(xb, tb, train=True)
        eng.backward()
        torch.cuda.synchronize()
        return loss.clone(), logits.clone(), am.clone(), eng.flat.g.clone()

    l1, lg1, am1, g1 = run(x, t)
    l2, lg2, am2, g2 = run(x, t)
    assert torch.equal(l1, l2) and torch.equal(lg1, lg2) and torch.equal(am1, am2) and torch.equal(g1, g2), "a step is not bit-reproducible"
    assert torch.isfinite(g1).all() and g1.abs().sum().item() > 0
    _, lg0, am0, _ = run(x[:1].contiguous(), t[:1].contiguous())
    assert torch.equal(lg0, lg1[:1]) and torch.equal(am0, am1[:1]), "GroupNorm statistics leak across samples"
    # the level-0 weight gradients of this shape take the streaming kernels (64 -> 64, 160^3 planes)
    xs = torch.randn(1, 8, 160, 160, 64, device=DEV, generator=gen).to(torch.bfloat16)
    dys = torch.randn(1, 8, 160, 160, 64, device=DEV, generator=gen).to(torch.bfloat16)
    dw = torch.empty(64, 64, 3, 3, 3, device=DEV)
    ops.wgrad(xs, dys, dw, ksize=3, Cin=64, Cout=64, grid=(1, 8, 160, 160))
    assert ops.wgrad_last_dispatch()[0] == "k3.3d.ppss"


def test_ping_pong_kernels_beyond_4gib_tensors():
    """The ping-pong kernels address their operands through PER-IMAGE buffer resources with 32-bit offsets: tensors far beyond 4 GiB in total (here 72 images of
    512 x 512 x 128 bf16 = 4.8 GB in, 9.7 GB out / dY) must behave exactly like small batches.  Checked by batch independence (the last and the first two images of the big
    batch against a 2-image run: bit-equal conv outputs) and linearity of the weight gradient (72-image dW == sum of per-slice dWs to fp32 accumulation order)."""
    from mdeical_image_segmentation_amd import ops
    N, H, W, Cin, Cout = 72, 512, 512, 128, 256
    gen = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(N, H, W, Cin, device=DEV, generator=gen).to(torch.bfloat16)
    assert x.numel() * 2 > (1 << 32)
    w = (torch.randn(9, Cout, Cin, device=DEV, generator=gen) * (9 * Cin) ** -0.5).to(torch.bfloat16)
    b = torch.randn(Cout, device=DEV, generator=gen)
    y = torch.empty(N, H, W, Cout, device=DEV, dtype=torch.bfloat16)
    ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, bias=b, relu=True)
    assert ops.conv_last_dispatch() == "k3.2d.ppc8"
    for sl in (slice(0, 2), slice(N - 2, N)):
        ys = torch.empty(2, H, W, Cout, device=DEV, dtype=torch.bfloat16)
        ops.conv_igemm(x[sl].contiguous(), w, ys, ksize=3, Cin=Cin, Cout=Cout, bias=b, relu=True)
        assert torch.equal(ys, y[sl]), "an image's output depends on its position in a > 4 GiB batch"
    # weight gradient over the whole batch vs the sum over 8-image slices (y doubles as dY)
    dw = torch.empty(Cout, Cin, 3, 3, device=DEV)
    ops.wgrad(x, y, dw, ksize=3, Cin=Cin, Cout=Cout)
    assert ops.wgrad_last_dispatch()[0] == "k3.2d.ppst"          # (the streaming form of the row kernel: per-image resources, offsets advanced row by row)
    acc = torch.zeros_like(dw)
    part = torch.empty_like(dw)
    for i in range(0, N, 8):
        ops.wgrad(x[i:i + 8].contiguous(), y[i:i + 8].contiguous(), part, ksize=3, Cin=Cin, Cout=Cout)
        acc += part
    rel = ((dw - acc).norm() / acc.norm()).item()
    assert rel < 1e-5, rel


@pytest.mark.parametrize("kind", ["2d", "3d"])
def test_unsynchronised_steps_with_side_stream_reductions_are_bit_reproducible(kind, monkeypatch):
    """ADVICE r2 (csrc/wgrad.hip wg_finish): the split-K slab reductions run on a second stream, ordered behind the MFMA kernel by an event from a process-lifetime ring.
    K back-to-back UNSYNCHRONISED train steps (nothing waits between them, so a reduction of step i may still be running when step i+1's kernels are enqueued; the two
    workspaces alternate) must end in bit-identical parameters and gradients on every repetition.  2-D: side-stream reductions are opt-in (MISAMD_SIDE_REDUCE=1);
    the 3-D engines use them by default - bf16 here, i.e. the ping-pong weight-gradient kernels with one slab per persistent block."""
    monkeypatch.setenv("MISAMD_SIDE_REDUCE", "1")
    gen = torch.Generator().manual_seed(3)
    if kind == "2d":
        from mdeical_image_segmentation_amd.engine2d import UNet2DEngine
        x = torch.randn(16, 1, 256, 256, generator=gen).to(DEV)
        y = torch.randint(0, 2, (16, 256, 256), generator=gen).to(DEV)
        make = lambda: UNet2DEngine(1, 2, dtype=torch.bfloat16, device=DEV, seed=0, lr=1e-4)      # noqa: E731
    else:
        from mdeical_image_segmentation_amd.engine3d import UNet3DEngine
        x = torch.randn(2, 1, 64, 64, 64, generator=gen).to(DEV)
        y = (torch.rand(2, 3, 64, 64, 64, generator=gen) > 0.5).float().to(DEV)
        make = lambda: UNet3DEngine(1, 3, dtype=torch.bfloat16, device=DEV, seed=0, lr=1e-4)      # noqa: E731
    finals = []
    for rep in range(3):
        eng = make()
        assert eng.side_reduce
        for _ in range(6):
            eng.train_step(x, y)
        torch.cuda.synchronize()
        finals.append((eng.flat.p.clone(), eng.flat.g.clone()))
    for p, g in finals[1:]:
        assert torch.equal(p, finals[0][0]) and torch.equal(g, finals[0][1])
