"""hipGraph replay of a fused train step (graph.GraphedTrainStep) against the same steps launched eagerly: losses, gradient norms and parameters
after several steps on changing data, 2-D (CE head) and 3-D (BCE+Dice head, GroupNorm) engines."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(make_engine, batches, graphed):
    from mdeical_image_segmentation_amd.graph import GraphedTrainStep
    eng = make_engine()
    losses = []
    if graphed:
        step = GraphedTrainStep(eng, batches[0][0], batches[0][1])
        for x, y in batches:
            losses.append(step(x, y).item())
        step.release()
    else:
        for x, y in batches:
            eng.forward(x, y, train=True)
            eng.backward()
            eng.optimizer_step()
            losses.append(eng.loss_buf[0].item())
    return losses, eng.flat.p.clone(), eng.step_count, eng.gradnorm.item()


@pytest.mark.parametrize("kind", ["2d", "3d"])
def test_graph_replay_equals_eager_steps(kind):
    gen = torch.Generator(device=DEV).manual_seed(11)
    if kind == "2d":
        from mdeical_image_segmentation_amd.engine2d import UNet2DEngine

        def make():
            return UNet2DEngine(1, 2, dtype=torch.float32, device=DEV, seed=0)
        batches = [(torch.randn(2, 1, 32, 48, device=DEV, generator=gen), torch.randint(0, 2, (2, 32, 48), device=DEV, generator=gen)) for _ in range(4)]
    else:
        from mdeical_image_segmentation_amd.engine3d import UNet3DEngine

        def make():
            return UNet3DEngine(1, 3, f_maps=(64, 128, 256), dtype=torch.float32, device=DEV, seed=0)
        batches = [(torch.randn(1, 1, 16, 16, 16, device=DEV, generator=gen), (torch.rand(1, 3, 16, 16, 16, device=DEV, generator=gen) > 0.5).float())
                   for _ in range(4)]
    le, pe, se, ge = _run(make, batches, False)
    lg, pg, sg, gg = _run(make, batches, True)
    assert se == sg == 4
    for a, b in zip(le, lg):
        assert abs(a - b) <= 2e-6 * max(1.0, abs(a)), (le, lg)
    assert abs(ge - gg) <= 1e-5 * max(1.0, abs(ge))
    assert ((pe - pg).norm() / pe.norm()).item() < 1e-5


def test_graph_replay_timing_report():
    """reports eager vs replay time of a tiny step (measured on MI355X / ROCm 7.2: 2.7 ms eager vs 3.0 ms replay at bs 1 x 64²: the ~150 dependent
    kernels are bound by the GPU's per-kernel dispatch, which a graph replay does not remove - see DESIGN.md); asserts only that both run"""
    from mdeical_image_segmentation_amd.engine2d import UNet2DEngine
    from mdeical_image_segmentation_amd.graph import GraphedTrainStep
    gen = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(1, 1, 64, 64, device=DEV, generator=gen)
    y = torch.randint(0, 2, (1, 64, 64), device=DEV, generator=gen)
    eng = UNet2DEngine(1, 2, dtype=torch.bfloat16, device=DEV, seed=0)

    def eager():
        eng.forward(x, y, train=True)
        eng.backward()
        eng.optimizer_step()

    def timeit(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    te = timeit(eager)
    step = GraphedTrainStep(eng, x, y)
    tg = timeit(lambda: step())
    print(f"bs 1 64x64 bf16 train step: eager {te * 1e3:.2f} ms, hipGraph replay {tg * 1e3:.2f} ms")
    assert tg > 0 and te > 0
