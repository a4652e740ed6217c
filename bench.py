#!/usr/bin/env python3
"""Headline benchmark: 2-D U-Net train step (fwd + CE + bwd + clip + AdamW) on synthetic 512x512 images.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: unet2d 1-ch -> 2-class, bs=32 per GPU, 512x512, bf16 activations / fp32 master
(weak scaling: 32 images per rank, gradients all-reduced over RCCL, overlapped with backward).
Prints ONE JSON line (rank 0) with the throughput, the roofline of the dominant kernel (HIP-event timed in the
timed region) and, at N=1, the CPU oracle timed on the host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_IMAGE_512 = 1154.0e9      # conv/convT MACs x2, fwd + wgrad + dgrad, UNet(1,2)
PEAK_BF16_TFLOPS = 2500.0          # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_F32_TFLOPS = 157.3


def cpu_baseline(batch, size, steps=2, warmup=1):
    """The CPU oracle (a port of the reference's ATen graph) on a bounded sample of the same workload."""
    from oracle import unet2d_oracle as o2
    p = o2.init_params(1, 2, seed=0)
    opt = o2.AdamW(p)
    g = torch.Generator().manual_seed(0)
    images = torch.randn(batch, 1, size, size, generator=g)
    labels = torch.randint(0, 2, (batch, size, size), generator=g)
    for _ in range(warmup):
        o2.train_step(p, opt, images, labels)
    t0 = time.perf_counter()
    for _ in range(steps):
        o2.train_step(p, opt, images, labels)
    dt = (time.perf_counter() - t0) / steps
    return batch / dt, dt


def cpu_baseline3d(size, steps=1, warmup=1):
    """The CPU oracle of the 3-D path (UNet3D(1,3) + BCE-Dice forward/backward + clip + AdamW in stock PyTorch) on ONE volume of the benchmark size."""
    from oracle import unet3d_oracle as o3
    p = o3.init_params(1, 3, seed=0)
    params = [v.requires_grad_(True) for v in p.values()]
    opt = torch.optim.AdamW(params, lr=5e-3, weight_decay=1e-3)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 1, size, size, size, generator=g)
    t = (torch.rand(1, 3, size, size, size, generator=g) > 0.5).float()

    def step():
        opt.zero_grad(set_to_none=True)
        o3.bce_dice_loss(o3.unet3d_forward(p, x, 4), t).backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()

    for _ in range(warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    return 1.0 / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--layers", action="store_true", help="print a per-layer table of the MFMA kernels to stderr")
    ap.add_argument("--net", default="1x2", choices=["1x2", "3x4"],
                    help="2-D net: 1-channel -> 2 classes (BASELINE configs[1], default at every N) or 3 -> 4 (configs[2])")
    ap.add_argument("--workload", default="2d", choices=["2d", "3d"],
                    help="2d = BASELINE configs[1] (headline); 3d = configs[3]: UNet3D(1,3) bs=2 128^3 (use --dtype f32)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from mdeical_image_segmentation_amd import ops
    from mdeical_image_segmentation_amd.ddp import GradReducer
    from mdeical_image_segmentation_amd.engine2d import UNet2DEngine

    if args.workload == "3d":
        return bench3d(args, rank, world, dev, dist)

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    cin, ncls = (1, 2) if args.net == "1x2" else (3, 4)
    eng = UNet2DEngine(cin, ncls, dtype=dtype, device=dev, seed=0)  # identical init on every rank
    reducer = GradReducer(eng.flat) if world > 1 else None
    g = torch.Generator().manual_seed(1000 + rank)                    # per-rank data shard
    images = torch.randn(args.batch, cin, args.size, args.size, generator=g).to(dev)
    labels = torch.randint(0, ncls, (args.batch, args.size, args.size), generator=g).to(dev)

    def step():
        eng.forward(images, labels, train=True, grad_scale=1.0 / world)
        if reducer is None:
            eng.backward()
        else:
            eng.backward(stage_cb=reducer.stage_done)
            reducer.finish()
        eng.optimizer_step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if not args.no_kernel_timing:
        ops.PROFILE = []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss = eng.loss_buf[0].item()

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.batch * args.steps / dt
        out = {
            "metric": "images/sec (2D 512x512 U-Net train step)", "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"unet2d {cin}-ch->{ncls}-class, bs={args.batch}/GPU {args.size}x{args.size}, "
                                   "fwd+CE loss+bwd+clip_grad_norm(1.0)+AdamW, random-init weights",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}", "final_loss": round(loss, 5)},
        }
        flop_img = FLOP_PER_IMAGE_512 * (args.size / 512.0) ** 2
        out["model_tflops"] = round(value * flop_img / 1e12, 1)
        if prof:
            agg, layers = {}, {}
            for key, flops, e0, e1 in prof:
                la = layers.setdefault(key, [0.0, 0.0, 0])
                la[0] += flops
                la[1] += e0.elapsed_time(e1) * 1e-3
                la[2] += 1
                key = key[:5]
                a = agg.setdefault(key, [0.0, 0.0, 0])
                a[0] += flops
                a[1] += e0.elapsed_time(e1) * 1e-3
                a[2] += 1
            dom = max(agg.items(), key=lambda kv: kv[1][1])
            key, (fl, sec, cnt) = dom
            peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
            ach = fl / sec / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(ach / peak, 4), "traffic": None, "kernel": "/".join(k for k in key if k),
                               "launches": cnt, "avg_launch_ms": round(sec / cnt * 1e3, 4),
                               "alg_gflop_per_launch": round(fl / cnt / 1e9, 2)}
            # HBM traffic of the dominant kernel cannot be counted from inside this process: it comes from the committed rocprofv3 PMC
            # passes of this same command (profiles/README.md), per launch like `achieved`
            try:
                import json as _json
                tr = _json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_h_traffic.json")))
                t = tr.get(out["roofline"]["kernel"])
                if t is not None and args.batch == 32 and args.size == 512:
                    out["roofline"]["traffic"] = t["hbm_read_bytes_per_launch"] + t["hbm_write_bytes_per_launch"]
                    out["roofline"]["traffic_unit"] = "bytes/launch (PMC: FETCH_SIZE x2 + WRITE_SIZE)"
            except (OSError, ValueError, KeyError):
                pass
            out["kernels"] = {"/".join(k for k in key if k): {"tflops": round(v[0] / v[1] / 1e12, 1), "ms_per_step": round(v[1] / args.steps * 1e3, 3),
                                                             "launches_per_step": v[2] // args.steps}
                              for key, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
            out["mfma_kernel_ms_per_step"] = round(sum(v[1] for v in agg.values()) / args.steps * 1e3, 3)
        if prof and args.layers:
            for key, v in sorted(layers.items(), key=lambda kv: -kv[1][1]):
                print(f"{v[1] / args.steps * 1e3:8.3f} ms/step {v[0] / v[1] / 1e12:7.1f} TF/s x{v[2] // args.steps}  {' '.join(k for k in key if k)}",
                      file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            cb, cs = 2, 512
            v, sdt = cpu_baseline(cb, cs)
            out["cpu_baseline"] = {"value": round(v, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": f"oracle (stock PyTorch CPU restatement of the reference) fp32 train step, bs={cb} {cs}x{cs}, "
                                             f"1 warm-up + 2 timed steps, {sdt:.2f} s/step"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


FLOP_PER_VOLUME_128 = 11359.7e9


def bench3d(args, rank, world, dev, dist):
    from mdeical_image_segmentation_amd import ops
    from mdeical_image_segmentation_amd.ddp import GradReducer
    from mdeical_image_segmentation_amd.engine3d import UNet3DEngine
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    batch = args.batch if args.batch != 32 else 2
    size = args.size if args.size != 512 else 128
    eng = UNet3DEngine(1, 3, dtype=dtype, device=dev, seed=0)
    reducer = GradReducer(eng.flat) if world > 1 else None
    g = torch.Generator().manual_seed(1000 + rank)
    x = torch.randn(batch, 1, size, size, size, generator=g).to(dev)
    t = (torch.rand(batch, 3, size, size, size, generator=g) > 0.5).float().to(dev)

    # on-device augmentation inside the timed step: flip + rot90 + rotate (+-30 deg, reflect; cubic spline
    # order 3 on raw, order 0 on targets) on raw and targets in lock-step, contrast (p=1) + Gaussian noise (p=1) on raw; parameters from the reference's streams
    import numpy as np
    from mdeical_image_segmentation_amd.augment.unet3d_augment import transforms as tr
    tr.GLOBAL_RANDOM_STATE = np.random.RandomState(47 + rank)
    def geo_(order):
        return [{"name": "RandomFlip"}, {"name": "RandomRotate90"},
                {"name": "RandomRotate", "axes": [[2, 1]], "angle_spectrum": 30, "mode": "reflect", "order": order}]
    geo = geo_(0)
    tf = tr.Transformer({"raw": geo_(3) + [{"name": "RandomContrast", "execution_probability": 1.0},
                                       {"name": "AdditiveGaussianNoise", "execution_probability": 1.0, "scale": [0.0, 0.1]}],
                         "label": geo}, {"mean": 0.0, "std": 1.0})
    rt, lt = tf.raw_transform(), tf.label_transform()
    xa, ta = torch.empty_like(x), torch.empty_like(t)

    def step():
        for b in range(batch):
            xa[b, 0] = rt(x[b, 0])
            ta[b] = lt(t[b])
        eng.forward(xa, ta, train=True, grad_scale=1.0 / world)
        if reducer is None:
            eng.backward()
        else:
            eng.backward(stage_cb=reducer.stage_done)
            reducer.finish()
        eng.optimizer_step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if not args.no_kernel_timing:
        ops.PROFILE = []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    if rank == 0:
        value = world * batch * args.steps / dt
        out = {"metric": f"volumes/sec (3D {size}^3 U-Net train step)", "value": round(value, 3), "unit": "volumes/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"unet3d 1-ch->3-class bs={batch}/GPU {size}^3, on-device augment (flip+rot90+rotate[order 3 raw / 0 targets]+contrast+noise) + fwd+BCEDice+bwd+clip+AdamW, random-init weights",
                          "global_batch": world * batch, "parallelism": f"dp{world}", "final_loss": round(eng.loss_buf[0].item(), 5)}}
        out["model_tflops"] = round(value * FLOP_PER_VOLUME_128 * (size / 128.0) ** 3 / 1e12, 1)
        if prof:
            agg, layers = {}, {}
            for key, flops, e0, e1 in prof:
                la = layers.setdefault(key, [0.0, 0.0, 0])
                la[0] += flops; la[1] += e0.elapsed_time(e1) * 1e-3; la[2] += 1
                a = agg.setdefault(key[:5], [0.0, 0.0, 0])
                a[0] += flops; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
            key, (fl, sec, cnt) = max(agg.items(), key=lambda kv: kv[1][1])
            peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
            out["roofline"] = {"bound": "mfma", "achieved": round(fl / sec / 1e12, 1), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(fl / sec / 1e12 / peak, 4), "traffic": None, "kernel": "/".join(k for k in key if k),
                               "launches": cnt, "avg_launch_ms": round(sec / cnt * 1e3, 3)}
            out["mfma_kernel_ms_per_step"] = round(sum(v[1] for v in agg.values()) / args.steps * 1e3, 2)
            if args.layers:
                for k, v in sorted(layers.items(), key=lambda kv: -kv[1][1]):
                    print(f"{v[1] / args.steps * 1e3:9.3f} ms/step {v[0] / v[1] / 1e12:7.1f} TF/s x{v[2] // args.steps}  {' '.join(x for x in k if x)}", file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            v, sdt = cpu_baseline3d(size)
            out["cpu_baseline"] = {"value": round(v, 4), "unit": "volumes/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": f"oracle (stock PyTorch CPU restatement of the reference) fp32 train step on one {size}^3 volume, "
                                             f"1 warm-up + 1 timed step, {sdt:.2f} s/step (no augmentation)"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
