"""Data-parallel path on the GPU box with two ranks sharing cuda:0 (gloo transport for CUDA tensors, since RCCL
refuses two ranks on one device): the bucketed, stream-overlapped GradReducer fed by the engine's backward stage
callbacks must reproduce the single-process gradient (same data on both ranks, 1/world folded into the loss grad)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, kind="2d"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mdeical_image_segmentation_amd.ddp import GradReducer
    torch.cuda.set_device(0)
    g = torch.Generator().manual_seed(3)
    if kind == "2d":
        from mdeical_image_segmentation_amd.engine2d import UNet2DEngine
        eng = UNet2DEngine(1, 2, dtype=torch.float32, device="cuda:0", seed=0)
        images = torch.randn(2, 1, 32, 32, generator=g).cuda()
        labels = torch.randint(0, 2, (2, 32, 32), generator=g).cuda()
    else:       # the 3-D engine (GroupNorm parameters, side-stream slab reductions joined per stage)
        from mdeical_image_segmentation_amd.engine3d import UNet3DEngine
        eng = UNet3DEngine(1, 3, f_maps=(64, 128, 256), dtype=torch.float32, device="cuda:0", seed=0)
        images = torch.randn(1, 1, 16, 16, 16, generator=g).cuda()
        labels = (torch.rand(1, 3, 16, 16, 16, generator=g) > 0.5).float().cuda()
    # single-process reference
    eng.forward(images, labels, train=True, grad_scale=1.0)
    eng.backward()
    ref = eng.flat.g.clone()
    # data-parallel: every rank sees the same shard, loss gradient scaled by 1/world, buckets all-reduced during backward
    red = GradReducer(eng.flat)
    seen = []
    eng.forward(images, labels, train=True, grad_scale=1.0 / world)
    eng.backward(stage_cb=lambda names: (seen.extend(names), red.stage_done(names)))
    red.finish()
    torch.cuda.synchronize()
    err = (eng.flat.g - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
    # every parameter belongs to a module prefix that was handed to the reducer (2-D: middle_conv goes out as middle_conv.second / middle_conv.first since round 5)
    all_covered = all(any(n.startswith(p + ".") for p in seen) for n, _ in eng.specs)
    out[rank] = (err, all_covered and len(seen) == len(set(seen)))
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["2d", "3d"])
def test_grad_reducer_two_ranks_one_gpu(kind):
    world = 2
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(world, port, out, kind), nprocs=world, join=True)
        res = dict(out)
    assert set(res) == {0, 1}
    for r, (err, all_stages) in res.items():
        assert err < 1e-5, (r, err)
        assert all_stages, "a module never reported its gradients to the reducer"


@pytest.mark.gpu
def test_native_rccl_communicator_single_rank():
    """The RCCL communicator behind the C ABI (mis_comm_unique_id / mis_comm_init / mis_allreduce_bucket / mis_comm_finalize) on the one GPU of this box:
    a 1-rank communicator really goes through librccl (ncclCommInitRank + ncclAllReduce on a side stream); the sum over one rank is the identity, the
    bucket schedule of GradReducer(backend="native") must leave the engine's gradients bit-identical, and the explicit init / finalize protocol holds."""
    import torch

    from mdeical_image_segmentation_amd import MisError, _lib
    from mdeical_image_segmentation_amd.ddp import GradReducer, native_comm_finalize, native_comm_init
    from mdeical_image_segmentation_amd.engine2d import UNet2DEngine
    lib = _lib.load()
    assert lib.mis_comm_world() == 0
    eng = UNet2DEngine(1, 2, dtype=torch.float32, device="cuda", seed=0)
    with pytest.raises(MisError):
        GradReducer(eng.flat, backend="native")                # no communicator yet
    assert native_comm_init() == 1 and lib.mis_comm_world() == 1
    with pytest.raises(MisError):
        native_comm_init()                                      # one communicator per process
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 1, 32, 32, generator=g).cuda()
    y = torch.randint(0, 2, (2, 32, 32), generator=g).cuda()
    eng.forward(x, y, train=True)
    eng.backward()
    torch.cuda.synchronize()
    ref = eng.flat.g.clone()
    red = GradReducer(eng.flat, backend="native", timing=True)
    eng.forward(x, y, train=True)
    eng.backward(stage_cb=red.stage_done)
    red.finish()
    torch.cuda.synchronize()
    assert red.buckets_per_step >= 8
    assert torch.equal(eng.flat.g, ref)
    ar, exposed = red.timing_ms()
    assert ar > 0.0
    native_comm_finalize()
    assert lib.mis_comm_world() == 0
    buf = torch.ones(8, device="cuda")
    assert lib.mis_allreduce_bucket(buf.data_ptr(), 8, None) != 0      # finalized: loud error, no crash


def _nccl1_worker(rank, port, out):
    import torch
    import torch.distributed as dist

    from mdeical_image_segmentation_amd.ddp import GradReducer
    from mdeical_image_segmentation_amd.engine2d import UNet2DEngine
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        eng = UNet2DEngine(1, 2, dtype=torch.float32, device="cuda", seed=0)
        g = torch.Generator().manual_seed(11)
        x = torch.randn(2, 1, 32, 32, generator=g).cuda()
        y = torch.randint(0, 2, (2, 32, 32), generator=g).cuda()
        eng.forward(x, y, train=True)
        eng.backward()
        torch.cuda.synchronize()
        ref = eng.flat.g.clone()
        red = GradReducer(eng.flat, timing=True)                 # backend "torch" = ProcessGroupNCCL = RCCL
        for _ in range(2):                                       # twice: the persistent events / bucket slots are reused
            eng.forward(x, y, train=True)
            eng.backward(stage_cb=red.stage_done)
            red.finish()
        dist.barrier()
        torch.cuda.synchronize()
        ar, _ = red.timing_ms()
        out["res"] = (dist.get_backend(), red.buckets_per_step, bool(torch.equal(eng.flat.g, ref)), ar > 0.0)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_torch_rccl_process_group_single_rank():
    """GradReducer's default backend on the real transport: torch.distributed 'nccl' (= RCCL) with a one-rank group on this box's GPU - the same calls an 8-GPU
    launch makes (init with device_id, all_reduce per bucket under the side stream, barrier), in a child process so that the group's lifetime is its own"""
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_nccl1_worker, args=(port, out), nprocs=1, join=True)
        res = out["res"]
    assert res[0] == "nccl" and res[1] >= 8 and res[2] and res[3], res


def _torch_ddp_worker(rank, world, port, out):
    """the drop-in nn.Module under torch's own DistributedDataParallel (what HF Trainer wraps it in under torchrun): its parameters alias the engine's flat fp32 buffer"""
    import torch
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel

    from mdeical_image_segmentation_amd.model.unet2d import UNetConfig, UNetModel
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        g = torch.Generator().manual_seed(5)
        xs = torch.randn(world, 2, 1, 32, 32, generator=g)
        ys = torch.randint(0, 2, (world, 2, 32, 32), generator=g)

        def make():
            torch.manual_seed(0)
            return UNetModel(UNetConfig(in_channels=1, out_channels=2, unet_type="UNet")).cuda()

        # reference in this process: per-shard gradients averaged by hand, stock AdamW
        ref = make()
        opt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=1e-2)
        for step in range(2):
            grads = None
            for r in range(world):
                ref.zero_grad()
                ref(images=xs[r].cuda(), labels=ys[r].cuda()).loss.backward()
                gr = [p.grad.detach().clone() for p in ref.parameters()]
                grads = gr if grads is None else [a + b for a, b in zip(grads, gr)]
            for p, gsum in zip(ref.parameters(), grads):
                p.grad = gsum / world
            opt.step()
        want = [p.detach().clone() for p in ref.parameters()]
        del ref, opt
        # the same two steps under DistributedDataParallel, one shard per rank
        model = make()
        ddp = DistributedDataParallel(model, device_ids=[0])
        opt = torch.optim.AdamW(ddp.parameters(), lr=1e-3, weight_decay=1e-2)
        for step in range(2):
            opt.zero_grad()
            ddp(images=xs[rank].cuda(), labels=ys[rank].cuda()).loss.backward()
            opt.step()
        torch.cuda.synchronize()
        eng = model.unet._engine if hasattr(model, "unet") and hasattr(model.unet, "_engine") else None
        worst = max(((a.detach() - b).norm() / b.norm().clamp_min(1e-30)).item() for a, b in zip(model.parameters(), want))
        aliased = eng is None or all(p.data_ptr() == eng.P[n].data_ptr() for n, p in model.unet.named_parameters())
        out[rank] = (worst, bool(aliased))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_unet_model_under_torch_ddp_two_ranks():
    """VERDICT r1 (weak 7): `UNetModel` wrapped in torch DistributedDataParallel, two ranks sharing this box's GPU over gloo - DDP's parameter broadcast, gradient buckets
    and the optimizer's in-place updates must leave the parameters aliased to the engine's flat buffer and reproduce hand-averaged gradients + AdamW to fp32 rounding"""
    world = 2
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_torch_ddp_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert set(res) == {0, 1}
    for r, (worst, aliased) in res.items():
        assert worst < 2e-5, (r, worst)
        assert aliased, "a parameter no longer aliases the engine's flat buffer after DDP + optimizer steps"


def _exact_dice_worker(rank, world, port, out):
    """two ranks on cuda:0 over gloo, each with ONE volume of a 2-volume batch: with exact_dice the summed gradients and the loss are those of the single-process
    run on the full batch (global Dice sums); without it they are not (per-rank Dice)"""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mdeical_image_segmentation_amd.ddp import GradReducer
    from mdeical_image_segmentation_amd.engine3d import UNet3DEngine
    torch.cuda.set_device(0)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 1, 16, 16, 16, generator=g).cuda()
    t = (torch.rand(2, 3, 16, 16, 16, generator=g) > (0.3 + 0.4 * torch.arange(2).view(2, 1, 1, 1, 1))).float().cuda()      # unequal foreground per sample: per-rank Dice != global Dice
    res = {}
    for exact in (True, False):
        eng = UNet3DEngine(1, 3, f_maps=(64, 128, 256), dtype=torch.float32, device="cuda:0", seed=0, exact_dice=exact)
        loss, _, _ = eng.forward(x, t, train=True)                       # single process, full batch: the reference semantics
        full_loss = loss.item()
        eng.backward()
        torch.cuda.synchronize()
        ref = eng.flat.g.clone()
        red = GradReducer(eng.flat)
        xs, ts = x[rank:rank + 1].contiguous(), t[rank:rank + 1].contiguous()
        # (the full-batch call above took the non-distributed branch only because its batch is what one rank would hold; the group IS initialised:
        #  with exact=True it already all-reduced its sums over two ranks that hold the same data - which doubles I, P, T alike and leaves the Dice ratio unchanged)
        loss, _, _ = eng.forward(xs, ts, train=True, grad_scale=1.0 / world)
        eng.backward(stage_cb=red.stage_done)
        red.finish()
        torch.cuda.synchronize()
        lt = loss.clone()
        if not exact:
            dist.all_reduce(lt)
            lt /= world
        res[exact] = ((eng.flat.g - ref).abs().max().item() / ref.abs().max().item(), abs(lt.item() - full_loss))
    out[rank] = res
    dist.destroy_process_group()


def test_exact_dice_two_ranks_equals_the_full_batch():
    """SURVEY §8e / reference model/unet3d/trainer.py:312-318: with `exact_dice` the 36-byte all-reduce of the Dice partial sums makes a 2-rank step equal to the
    single-process step on the gathered batch (loss and every gradient); the default per-rank Dice (DDP semantics) measurably is not"""
    world = 2
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_exact_dice_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert set(res) == {0, 1}
    for r, d in res.items():
        gerr, lerr = d[True]
        assert gerr < 2e-5 and lerr < 2e-6, (r, "exact", gerr, lerr)
        gerr, lerr = d[False]
        assert gerr > 1e-4 or lerr > 1e-5, (r, "per-rank Dice should differ from the global-batch Dice on this data", gerr, lerr)


@pytest.mark.parametrize("workload", ["2d", "3d"])
def test_bench_launch_at_two_ranks_prints_one_line_with_a_comm_report(workload):
    """the N > 1 leg of bench.py - launched exactly as the driver launches it (python -m torch.distributed.run ... bench.py --gpus N), as a FRESH child
    process (nothing in it has touched the GPU before the launcher starts the ranks) - rehearsed on the one-GPU box: two ranks share the card over gloo
    (MISAMD_BENCH_REHEARSAL=gloo).  Checks what the first real 8-GPU run will be read for: exactly one JSON line on stdout, the comm report (ranks, per-bucket list in issue
    order with the head first and the biases last, exposed time), parameters / gradients bit-identical across ranks, and for the 3-D workload the global-batch Dice."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MISAMD_BENCH_REHEARSAL="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    size = ["--batch", "2", "--size", "64"] if workload == "2d" else ["--workload", "3d", "--dtype", "f32", "--batch", "1", "--size", "32"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-kernel-timing"] + size
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must be ONE JSON line, got {len(lines)}: {r.stdout[:500]}"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    comm = out["comm"]
    assert comm["rccl_ranks"] == 2 and comm["params_identical_across_ranks"] and comm["grads_identical_across_ranks"]
    assert "exposed_comm_ms_per_step" in comm and "allreduce_ms_per_step" in comm
    pb = comm["per_bucket"]
    assert len(pb) == comm["buckets_per_step"] >= 4 and [b["bucket"] for b in pb] == list(range(len(pb)))
    assert sum(b["bytes"] for b in pb) >= comm["allreduce_bytes_per_step"] * 0.99          # every gradient byte travels (ranges are padded to 256 bytes)
    assert all("exposed_ms" in b and "allreduce_ms" in b for b in pb)
    if workload == "2d":
        # issue order: head first (final_conv.weight: 2 x 64 floats, padded to 64-float granules), ..., biases last; middle_conv goes out as TWO buckets, the 1024 x 1024 x 9
        # layer (37.7 MB) ahead of the 512 -> 1024 one (18.9 MB)
        sizes = [b["bytes"] for b in pb]
        assert sizes[0] == 128 * 4
        i2 = sizes.index(1024 * 1024 * 9 * 4)
        assert sizes[i2 + 1] == 1024 * 512 * 9 * 4
    else:
        assert comm["exact_dice"] is True
