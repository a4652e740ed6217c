#!/usr/bin/env python3
"""Headline benchmark: 2-D U-Net train step (fwd + CE + bwd + clip + AdamW) on synthetic 512x512 images.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                      # cfg2 shape per rank (default at every N)
    ... bench.py --gpus 8 --net 3x4                                  # BASELINE configs[2]: UNet(3,4), bs 32/GPU = 256 global
    ... bench.py --gpus 8 --workload 3d --dtype bf16 --size 160      # BASELINE configs[4]: UNet3D(1,3), 2 volumes/GPU of 160^3
    python bench.py --workload 3d --dtype f32                        # BASELINE configs[3]: UNet3D(1,3) bs=2 128^3 fp32 + on-device augment

Workload = BASELINE.json configs[1]: unet2d 1-ch -> 2-class, bs=32 per GPU, 512x512, bf16 activations / fp32 master
(weak scaling: 32 images per rank, gradients all-reduced over RCCL, overlapped with backward).
Prints ONE JSON line (rank 0) with the throughput, the roofline of the dominant kernel (HIP-event timed in the
timed region), the state of the network during the timed steps (per-step loss, live-activation fractions: the run
FAILS if the net has collapsed) and, at N=1, the CPU oracle timed on the host cores with the cfg1 protocol of
SURVEY.md §8(d) plus driver-timed secondary legs (2-D fp32 parity mode, cfg4 3-D fp32).
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
FLOP_PER_IMAGE_512_3X4 = 1155.4e9  # ... UNet(3,4)
FLOP_PER_VOLUME_128 = 11359.7e9
PEAK_BF16_TFLOPS = 2500.0          # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_F32_TFLOPS = 157.3
MIN_LIVE_FRACTION = 0.20           # a timed step with fewer live (non-zero) activations than this is a collapsed network: invalid


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only): the oracle = stock-PyTorch restatement of the reference's ATen graph
# ------------------------------------------------------------------------------------------------------------------
def host_cores():
    """(logical CPUs this process may run on, physical cores among them)"""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    cores, proc, phys = set(), None, 0
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                proc = int(line.split(":")[1])
            elif line.startswith("physical id"):
                phys = int(line.split(":")[1])
            elif line.startswith("core id") and proc in allowed:
                cores.add((phys, int(line.split(":")[1])))
    except OSError:
        pass
    return len(allowed), (len(cores) if cores else len(allowed))


def _cpu_steps(threads, steps, warmup=1):
    from oracle import unet2d_oracle as o2
    torch.set_num_threads(threads)
    p = o2.init_params(1, 2, seed=0)
    opt = o2.AdamW(p)                      # lr 5e-3, wd 1e-3 on the non-bias parameters (HF split), clip 1.0 inside train_step
    g = torch.Generator().manual_seed(0)
    images = torch.randn(4, 1, 256, 256, generator=g)
    labels = torch.randint(0, 2, (4, 256, 256), generator=g)
    for _ in range(warmup):
        o2.train_step(p, opt, images, labels)
    t0 = time.perf_counter()
    for _ in range(steps):
        o2.train_step(p, opt, images, labels)
    return (time.perf_counter() - t0) / steps


def cpu_baseline():
    """cfg1 protocol of SURVEY.md §8(d) / BASELINE.md §4: UNet(1,2), 4 x 1 x 256 x 256 N(0,1) (seed 0), int64 labels, CE, AdamW + clip,
    fp32; 1 warm-up + 5 timed steps on all physical cores of the host (count stated) and 1 + 3 on 8 threads (the survey's figure)."""
    before = torch.get_num_threads()
    logical, physical = host_cores()
    s_all = _cpu_steps(physical, 5)
    s_8 = _cpu_steps(min(8, logical), 3) if physical != 8 else s_all
    torch.set_num_threads(before)
    # `value` = the FASTER of the two thread counts (oneDNN does not scale a batch-4 256^2 step over 128 cores: 8 threads beat 128 on the GPU box); both are reported
    best, cores = (s_all, physical) if s_all <= s_8 else (s_8, min(8, logical))
    img256 = 4.0 / best
    return {"value": round(img256 / 4.0, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"cfg1 protocol: oracle (stock-PyTorch CPU restatement of the reference) UNet(1,2) fp32 fwd+CE+bwd+clip+AdamW on 4x1x256x256; value = the faster of "
                      f"[1 warm-up + 5 timed steps on {physical} threads = physical cores of the host ({logical} logical CPUs available): {s_all:.2f} s/step] and "
                      f"[1 + 3 steps on {min(8, logical)} threads: {s_8:.2f} s/step (survey container: 3.17 s/step)] = {img256:.3f} 256x256-images/s on {cores} threads = value x 4 "
                      f"(value is in 512x512-equivalents: same per-pixel work)",
            "s_per_step": round(best, 3), "images256_per_s": round(img256, 4), "all_cores_s_per_step": round(s_all, 3), "physical_cores": physical,
            "threads8_s_per_step": round(s_8, 3), "threads8_images256_per_s": round(4.0 / s_8, 4), "logical_cpus": logical}


def cpu_baseline3d_entry(size):
    """the `cpu_baseline` object of a 3-D line: the oracle's train step on ONE volume (B = 1: SURVEY.md §8d), 1 warm-up + 1 timed step on the host's physical cores"""
    v, sdt, cores = cpu_baseline3d(size)
    return {"value": round(v, 4), "unit": "volumes/s", "cores": cores, "kind": "port", "s_per_step": round(sdt, 2),
            "sample": f"oracle (stock PyTorch CPU restatement of the reference: UNet3D(1,3) 'gcr' + BCEDiceLoss, reference model/unet3d/model.py:125-151, losses.py:167-178) fp32 "
                      f"fwd + loss + bwd + clip + AdamW on one {size}^3 volume (B = 1), 1 warm-up + 1 timed step on {cores} threads = physical cores of the host, "
                      f"{sdt:.2f} s/step (no augmentation)"}


def cpu_baseline3d(size, steps=1, warmup=1):
    """The CPU oracle of the 3-D path (UNet3D(1,3) + BCE-Dice forward/backward + clip + AdamW in stock PyTorch) on ONE volume of the benchmark size."""
    from oracle import unet3d_oracle as o3
    logical, physical = host_cores()
    before = torch.get_num_threads()
    torch.set_num_threads(physical)
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
    torch.set_num_threads(before)
    return 1.0 / dt, dt, physical


# ------------------------------------------------------------------------------------------------------------------
def live_fractions(tensors):
    """fraction of non-zero elements of post-ReLU activation buffers (one small reduction each, outside the timed region)"""
    return {k: round(float((t > 0).float().mean().item()), 4) for k, t in tensors.items()}


# dispatch configuration (mis_conv_last_dispatch / mis_wgrad_last_dispatch) -> the kernel symbol rocprofv3 shows for it
SYMBOLS = {
    "k3.2d.pp256": "conv_pp_kernel<8>", "k3.2d.pp128": "conv_pp_kernel<4>",
    "k3.2d.pp64": "conv_pp_kernel<2>", "k3.2d.ws64": "conv64_ws_kernel", "k3.2d.rs64": "conv_pp_rs64_kernel",
    "k3.2d.ppw": "wgrad_pp_wide_kernel<false, false>", "k3.2d.pps": "wgrad_pp_wide_kernel<true, false>", "k3.2d.ppwr": "wgrad_pp_row_kernel<false, false>",
    "k3.2d.ppsr": "wgrad_pp_row_kernel<true, false>", "k1.2d.ppg": "wgrad1_pp_kernel", "k3.2d.ppst": "wgrad_pp_stream_kernel<false, false>", "k3.2d.ppss": "wgrad_pp_stream_kernel<true, false>", "k3.3d.ppst": "wgrad_pp_stream_kernel<false, true>", "k3.3d.ppss": "wgrad_pp_stream_kernel<true, true>", "k3.3d.ppw": "wgrad_pp_wide_kernel<false, true>", "k3.3d.pps": "wgrad_pp_wide_kernel<true, true>",
    "k3.3d.ppwr": "wgrad_pp_row_kernel<false, true>", "k3.3d.ppsr": "wgrad_pp_row_kernel<true, true>", "k3.2d.pp": "wgrad_pp_kernel<2>",
    "k3.3d.f32pp64": "conv3d_f32_kernel<2>", "k3.3d.f32pp128": "conv3d_f32_kernel<4>", "k3.3d.f32pp32": "conv3d_f32_kernel<1>",
    "k3.3d.f32s": "wgrad_f32_stream_kernel<true, false>", "k3.2d.f32s": "wgrad_f32_stream_kernel<false, false>", "k3.3d.f32s32": "wgrad_f32_stream_kernel<true, true>",
    "k3.2d.f32s32": "wgrad_f32_stream_kernel<false, true>",
}
# the column-segment kernels: one instantiation per epilogue mask path (template argument 0 = none, 1 = bf16 mask ".mask", 2 = ReLU bits ".bits")
for _t, _k in (("k3.2d.ppc8", "conv_ppc_kernel<8, 4"), ("k3.2d.ppc8n2", "conv_ppc_kernel<8, 2"), ("k3.3d.ppc5", "conv3d_ppc_kernel<5, 4"),
               ("k3.3d.ppc8", "conv3d_ppc_kernel<8, 4"), ("k3.3d.ppc5n6", "conv3d_ppc_kernel<5, 6"), ("k3.3d.ppc5n2", "conv3d_ppc_kernel<5, 2"),
               ("k3.3d.ppc8n2", "conv3d_ppc_kernel<8, 2"), ("k3.3d.ppc10n2", "conv3d_ppc_kernel<10, 2"), ("k3.3d.ppc10", "conv3d_ppc_kernel<10, 4")):
    for _sfx, _em in (("", 0), (".mask", 1), (".bits", 2), (".gn", 3)):          # .gn: GroupNorm backward in the epilogue (3-D dgrads, MisConvDesc.gn_p)
        SYMBOLS[_t + _sfx] = f"{_k}, {_em}>"
for _sfx, _em in (("", 0), (".mask", 1), (".bits", 2)):
    SYMBOLS["k1.2d.pp" + _sfx] = f"gemm1_pp_kernel<{_em}>"
    SYMBOLS["k3.2d.ppd8" + _sfx] = f"conv_ppd_kernel<{_em}>"
    SYMBOLS["k3.2d.pps" + _sfx] = f"conv_pps_kernel<{_em}>"
    SYMBOLS["k3.2d.ppc2" + _sfx] = f"conv_ppc2_kernel<{_em}>"
SYMBOLS["k3.2d.ppd8.head"] = "conv_ppd_head_kernel<C>"          # (C = classes: 2 for the headline net, 4 for cfg3's)


def kernel_tables(prof, steps, peak):
    prof, steps = prof          # (events, number of steps whose launches were bracketed)
    agg, layers, tags = {}, {}, {}
    # flops = ALGORITHMIC work of the launch (the layer's real channel counts); xflops = what the kernel executed (more where an operand is zero-padded to the tile width:
    # the bf16 3-D engine's encoders.0 SingleConv2 dgrad / weight gradient) - achieved / frac / tflops are priced on the first, the second is reported beside it
    for key, flops, e0, e1, tag, xflops in prof:
        ms = e0.elapsed_time(e1) * 1e-3
        if tag.startswith("k3.3d.f32pp"):          # the fp32 3-D launcher picks the tile width by grid size too (64 columns on underfilled grids): key by what ran
            key = key[:4] + ("bn" + tag[len("k3.3d.f32pp"):],) + key[5:]
        for table, k in ((layers, key), (agg, key[:5])):
            a = table.setdefault(k, [0.0, 0.0, 0, 0.0])
            a[0] += flops
            a[1] += ms
            a[2] += 1
            a[3] += xflops
        t = tags.setdefault(key[:5], {})
        t[tag] = t.get(tag, 0) + 1
    key, (fl, sec, cnt, xfl) = max(agg.items(), key=lambda kv: kv[1][1])
    ach = fl / sec / 1e12
    # the dominant key's launches by kernel symbol (conv_igemm.hip's template kernels print as conv_igemm_kernel<...> in rocprofv3: the configuration tag stands for them)
    syms = {SYMBOLS.get(t, f"conv_igemm_kernel<{t}>" if key[0] == "conv_igemm" else f"wgrad_kernel<{t}>"): c for t, c in sorted(tags[key].items(), key=lambda kv: -kv[1])}
    roof = {"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
            "kernel": " + ".join(syms), "kernel_launches_by_symbol": syms, "key": "/".join(k for k in key if k), "launches": cnt,
            "avg_launch_ms": round(sec / cnt * 1e3, 4), "alg_gflop_per_launch": round(fl / cnt / 1e9, 2), "executed_gflop_per_launch": round(xfl / cnt / 1e9, 2)}
    def _syms(k):          # the key's launches (bracketed steps) by kernel symbol - tests/test_profiles_consistency.py holds them against the rocprofv3 CSV of the same run
        return {SYMBOLS.get(t, f"conv_igemm_kernel<{t}>" if k[0] == "conv_igemm" else f"wgrad_kernel<{t}>"): c for t, c in sorted(tags[k].items(), key=lambda kv: -kv[1])}
    kernels = {"/".join(k for k in key if k): {"tflops": round(v[0] / v[1] / 1e12, 1), "ms_per_step": round(v[1] / steps * 1e3, 3),
                                              "launches_per_step": v[2] // steps, **({"tflops_executed": round(v[3] / v[1] / 1e12, 1)} if v[3] != v[0] else {}),
                                              "launches_by_symbol": _syms(key)}
               for key, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
    total = round(sum(v[1] for v in agg.values()) / steps * 1e3, 3)
    return roof, kernels, total, layers, steps


def attach_traffic(roof, batch, size, fname="traffic.json", sources=None):
    """HBM bytes per launch of the dominant kernel come from the rocprofv3 PMC passes of THIS command (profiles/README.md), which cannot run
    inside the process.  The figure is only reported when the committed summary was taken on the same kernel sources (hash match) and shape.
    fname / sources: the traffic file and the source list its hash covers (2-D headline: traffic.json / _lib.TRAFFIC_SOURCES; cfg4: traffic_3d_f32.json)."""
    from mdeical_image_segmentation_amd import _lib
    path = os.path.join(ROOT, "profiles", fname)
    try:
        tr = json.load(open(path))
    except (OSError, ValueError):
        return
    ent = tr.get("kernels", {}).get(roof["key"])
    roof["traffic_source_hash"] = tr.get("source_hash")
    if ent is None or tr.get("source_hash") != _lib.source_hash(sources or _lib.TRAFFIC_SOURCES) or tr.get("batch") != batch or tr.get("size") != size:
        roof["traffic_note"] = f"profiles/{fname} was collected on other kernel sources or another shape: not reported"
        return
    roof["traffic"] = ent["hbm_read_bytes_per_launch"] + ent["hbm_write_bytes_per_launch"]
    roof["traffic_unit"] = (f"bytes/launch, FROM THE COMMITTED PROFILE profiles/{tr.get('profile')} of this command on these kernel sources (hash-checked), not measured "
                            "in this run (PMC: FETCH_SIZE x2 on gfx950 + WRITE_SIZE; separate --pmc passes)")
    roof["traffic_profile"] = tr.get("profile")


DDP_ON = False


class _StdoutToStderr:
    """RCCL 2.26 prints a version banner ("RCCL version : ...", five lines) on STDOUT when rank 0 creates its first communicator; the contract of this file is ONE JSON line on
    stdout.  While the process group (and the C-ABI communicator) come up, file descriptor 1 points at stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="units per GPU (default 32 images / 2 volumes)")
    ap.add_argument("--size", type=int, default=None, help="edge length (default 512 / 128)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--lr", type=float, default=1e-5,
                    help="constant learning rate of the timed steps.  The reference's 5e-3 (train.py:98) makes THIS synthetic task (random labels) "
                         "diverge and collapse to dead ReLUs within ~10 steps (loss = ln 2, activations 0-2 %% live): AdamW does the same work at any "
                         "lr, so the benchmark keeps the network in its initial, live state instead of timing MFMAs on zeros")
    ap.add_argument("--comm", default="torch", choices=["torch", "native"],
                    help="gradient exchange at N > 1: torch.distributed all_reduce (ProcessGroupNCCL = RCCL) or the RCCL communicator behind the C ABI "
                         "(mis_comm_init / mis_allreduce_bucket, csrc/comm.cpp); same bucket schedule either way")
    ap.add_argument("--persist-cus", type=int, default=None,
                    help="grid of the persistent kernels (default 256 = one block per CU).  At N > 1 an RCCL kernel shares the device with the backward kernels: "
                         "e.g. 248 leaves 8 CUs free for it (MIS_PERSIST_CUS, csrc/dispatch_cfg.hpp) - the A/B switch for the multi-GPU runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary driver-timed legs (2-D fp32, cfg4 3-D fp32) of the default N=1 run")
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
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run "
                  f"(python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...)",
                  file=sys.stderr)
        sys.exit(2)

    # the CPU baseline runs FIRST (GPU idle), so that the GPU legs form one contiguous busy window at the end of the run
    cpu = cpu3 = None
    if world == 1 and not args.no_cpu_baseline and args.workload == "2d":
        cpu = cpu_baseline()
        default_shape_ = (args.batch or 32) == 32 and (args.size or 512) == 512 and args.net == "1x2" and args.dtype == "bf16"
        if default_shape_ and not args.no_extra:
            cpu3 = cpu_baseline3d_entry(128)          # the 3-D half of the metric (cfg4) gets its CPU figure beside it too (last row): ~35 s

    import torch.distributed as dist
    if os.environ.get("MISAMD_BENCH_REHEARSAL"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # MISAMD_BENCH_REHEARSAL=nccl1 (with --gpus 1): the whole N > 1 code path - RCCL process group, bucketed all-reduces on the side stream under the backward,
    # barriers, max-over-ranks, comm report - with a communicator of ONE rank, which is what a one-GPU box can run of it; never a number
    global DDP_ON
    DDP_ON = world > 1 or os.environ.get("MISAMD_BENCH_REHEARSAL") == "nccl1"
    if DDP_ON:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # MISAMD_BENCH_REHEARSAL=gloo: several ranks SHARING one GPU over gloo - exercises the N > 1 code path (buckets, comm report) on a one-GPU box; never a number
        backend = os.environ.get("MISAMD_BENCH_REHEARSAL", "nccl")
        if backend == "nccl1":
            backend = "nccl"
        with _StdoutToStderr():
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)                      # (the communicator exists after this call whichever way the backend initialises it)
            torch.cuda.synchronize()
        if dist.get_world_size() != args.gpus or dist.get_backend() != backend or (world > 1 and not os.environ.get("MISAMD_BENCH_REHEARSAL") and dist.get_backend() != "nccl"):
            # (not an assert: python -O must not turn a gloo / wrong-size run into a number)
            raise SystemExit(f"bench.py: N > 1 needs the RCCL process group of {args.gpus} ranks: got backend {dist.get_backend()!r}, world size {dist.get_world_size()}")
        if args.comm == "native":
            from mdeical_image_segmentation_amd.ddp import native_comm_init
            with _StdoutToStderr():
                n_native = native_comm_init()          # (not inside an assert: python -O would strip the CALL)
            if n_native != args.gpus:
                raise SystemExit(f"bench.py: the C-ABI communicator came up with {n_native} ranks, {args.gpus} wanted")

    if args.persist_cus is not None:
        from mdeical_image_segmentation_amd import ops
        ops.dispatch_override("MIS_PERSIST_CUS", args.persist_cus)

    if args.workload == "3d":
        out = run3d(args, rank, world, dev, dist, dtype=args.dtype, batch=args.batch or 2, size=args.size or 128, steps=args.steps,
                    warmup=args.warmup, timing=not args.no_kernel_timing, layers=args.layers)
        if rank == 0:
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline3d_entry(args.size or 128)
            print(json.dumps(out), flush=True)
    else:
        out = run2d(args, rank, world, dev, dist, dtype=args.dtype, batch=args.batch or 32, size=args.size or 512, steps=args.steps,
                    warmup=args.warmup, timing=not args.no_kernel_timing, layers=args.layers)
        if rank == 0:
            if cpu is not None:
                out["cpu_baseline"] = cpu
            default_shape = (args.batch or 32) == 32 and (args.size or 512) == 512 and args.net == "1x2" and args.dtype == "bf16"
            if world == 1 and default_shape and not args.no_extra:
                # secondary legs, so that the driver's clock covers them too: the fp32 parity mode of the same workload and cfg4 (3-D fp32 + augment)
                ex = {}
                torch.cuda.empty_cache()
                o = run2d(args, rank, world, dev, dist, dtype="f32", batch=32, size=512, steps=3, warmup=1, timing=True, layers=False)
                ex["unet2d_f32_parity_mode"] = {k: o[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "model_tflops", "roofline", "loss_per_step",
                                                                   "act_nonzero_frac", "clock")}
                torch.cuda.empty_cache()
                o = run3d(args, rank, world, dev, dist, dtype="f32", batch=2, size=128, steps=5, warmup=2, timing=True, layers=False)
                ex["unet3d_cfg4_f32_128"] = {k: o[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "model_tflops", "roofline",
                                                                "kernels", "mfma_kernel_ms_per_step", "loss_per_step", "config", "clock")}
                if cpu3 is not None:
                    ex["unet3d_cfg4_f32_128"]["cpu_baseline"] = cpu3
                # the per-GPU shapes of the two 8-GPU configurations (BASELINE configs[2] and configs[4]), so that they are driver-timed at N = 1 as well
                torch.cuda.empty_cache()
                a3 = argparse.Namespace(**{**vars(args), "net": "3x4"})
                o = run2d(a3, rank, world, dev, dist, dtype="bf16", batch=32, size=512, steps=5, warmup=2, timing=True, layers=False)
                ex["unet2d_cfg3_3x4_bf16_per_gpu_shape"] = {k: o[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "model_tflops", "roofline",
                                                                               "loss_per_step", "act_nonzero_frac", "config", "clock")}
                del o
                torch.cuda.empty_cache()
                o = run3d(args, rank, world, dev, dist, dtype="bf16", batch=2, size=160, steps=5, warmup=2, timing=True, layers=False)
                ex["unet3d_cfg5_bf16_160_per_gpu_shape"] = {k: o[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "model_tflops",
                                                                               "roofline", "kernels", "loss_per_step", "config", "clock")}
                del o
                torch.cuda.empty_cache()
                o = run_dropin2d(args, dev, batch=32, size=512, steps=5, warmup=2)
                o["vs_engine_loop_ms"] = round(o["ms_per_step"] / out["ms_per_step"], 4)
                ex["unet2d_dropin_UNetModel_autograd_torch_adamw"] = o
                out["extra"] = ex
            print(json.dumps(out), flush=True)
    if DDP_ON:
        if args.comm == "native":
            from mdeical_image_segmentation_amd.ddp import native_comm_finalize
            torch.cuda.synchronize()
            native_comm_finalize()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------
LAST_STEP_TRACE = None
LAST_CLOCK = None


class ClockSampler:
    """Shader clock and socket power of the device WHILE the timed steps run ("settle the clock"): a host thread reads the amdgpu hwmon files of the card
    (freq1_input = sclk in Hz, power1_input = socket power in uW; plain sysfs reads, no GPU runtime call, nothing inside any kernel) every `period` seconds.  The dense
    MFMA peaks of MI355X_MICROARCH.md are quoted at the 2.4 GHz boost clock; under the bf16 MFMA load of this benchmark the part holds less (PMC: GRBM_GUI_ACTIVE / duration =
    1.93 GHz in round 5) - the line reports what it held and the peak at THAT clock beside the nominal one.  Never fails the benchmark: no readable hwmon -> no `clock`."""

    def __init__(self, dev, period=0.02):
        import glob
        import threading
        self.period, self.samples, self._stop, self._thr = period, [], threading.Event(), None
        cands = []
        for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")):
            h = os.path.dirname(f)
            if os.access(f, os.R_OK) and os.access(os.path.join(h, "power1_input"), os.R_OK):
                cands.append((os.path.basename(os.path.realpath(os.path.join(h, "..", ".."))), h))          # (PCI address of the card, hwmon directory)
        self.hwmon = None
        if cands:
            want = None
            try:
                p = torch.cuda.get_device_properties(dev)
                want = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            except Exception:
                pass
            match = [h for a, h in cands if a == want]
            self.hwmon = match[0] if match else (cands[0][1] if len(cands) == 1 else None)
        self.cap_w = None
        if self.hwmon:
            try:
                self.cap_w = int(open(os.path.join(self.hwmon, "power1_cap")).read()) / 1e6
            except (OSError, ValueError):
                pass

    def _run(self):
        fq, pw = os.path.join(self.hwmon, "freq1_input"), os.path.join(self.hwmon, "power1_input")
        while not self._stop.is_set():
            try:
                self.samples.append((time.perf_counter(), int(open(fq).read()) / 1e6, int(open(pw).read()) / 1e6))
            except (OSError, ValueError):
                pass
            self._stop.wait(self.period)

    def start(self):
        if self.hwmon:
            import threading
            self._thr = threading.Thread(target=self._run, daemon=True)
            self._thr.start()

    def stop(self):
        if self._thr is None:
            return None
        self._stop.set()
        self._thr.join()
        s = self.samples[1:] if len(self.samples) > 3 else self.samples          # (the first sample may predate the first kernel)
        if not s:
            return None
        mhz, w = [v[1] for v in s], [v[2] for v in s]
        return {"sclk_mhz_mean": round(sum(mhz) / len(mhz), 1), "sclk_mhz_min": round(min(mhz), 1), "sclk_mhz_max": round(max(mhz), 1),
                "socket_power_w_mean": round(sum(w) / len(w), 1), "socket_power_w_max": round(max(w), 1), "power_cap_w": self.cap_w, "samples": len(s),
                "source": f"amdgpu hwmon freq1_input / power1_input every {int(self.period * 1e3)} ms during the timed steps (host thread, sysfs reads only)"}


# What the matrix pipe of an MI355X of this pool DELIVERS on a pure MFMA stream with operands that toggle like real data (scripts/mfma_peak.hip -> profiles/r06_mfma_peak.txt:
# register operands only, eight different pseudo-random pairs, 256 blocks): the part's power management holds 2.17-2.19 GHz at ~1.25 kW on v_mfma_f32_16x16x32_bf16
# (2102-2108 TFLOP/s = 0.84 of the nominal dense peak on two boxes, 2043 at 2.10 GHz on a third; constant operands: 2362-2425 at 2.39 GHz / 0.83 kW) and 2.39 GHz on v_mfma_f32_16x16x4_f32 (154.1-154.7 TFLOP/s = 0.98).
# Reported beside `peak` (the nominal figure of MI355X_MICROARCH.md, which `frac` is priced on); never replaces it.
MEASURED_MFMA_PEAK = {2500.0: 2105.0, 157.3: 154.7}


def attach_clock(roof, nominal_mhz=2400.0):
    """roofline.peak is the dense peak at the 2.4 GHz boost clock; next to it: the peak at the clock the part HELD during these timed steps and the fraction of that, and the
    rate a pure MFMA stream with toggling operands was MEASURED to deliver on this pool (MEASURED_MFMA_PEAK) and the fraction of that"""
    if roof is not None and roof.get("peak") in MEASURED_MFMA_PEAK:
        roof["peak_measured_pure_mfma"] = MEASURED_MFMA_PEAK[roof["peak"]]
        roof["frac_of_measured_peak"] = round(roof["achieved"] / MEASURED_MFMA_PEAK[roof["peak"]], 4)
        roof["peak_measured_source"] = "profiles/r06_mfma_peak.txt (scripts/mfma_peak.hip: register-operand MFMA stream, pseudo-random operands, 256 blocks)"
    if LAST_CLOCK and roof is not None and LAST_CLOCK.get("sclk_mhz_mean"):
        held = roof["peak"] * LAST_CLOCK["sclk_mhz_mean"] / nominal_mhz
        roof["peak_at_held_clock"] = round(held, 1)
        roof["frac_of_held_clock_peak"] = round(roof["achieved"] / held, 4)
        roof["held_sclk_mhz"] = LAST_CLOCK["sclk_mhz_mean"]


def _timed_loop(step, steps, warmup, world, dist, dev, timing):
    from mdeical_image_segmentation_amd import ops

    def fence():
        torch.cuda.synchronize()
        if DDP_ON:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(warmup):
        step(-1)
    # per-launch HIP events cost ~1.5 % of the step when every launch of every step is bracketed: bracket the launches of at most three timed
    # steps (first, middle, last) - still inside the timed region - and leave the others untouched
    timed_steps = sorted({0, steps // 2, steps - 1}) if timing else []
    prof = []
    if timing:
        # the brackets' events are created and recorded once BEFORE the timed region (ops.prepare_events): created inside it, the ~130 events of the first bracketed
        # step made that step host-bound (49-67 ms instead of 33.5: the `step_trace` of a run shows it), i.e. the 10-step default under-reported by 5-9 %
        ops.prepare_events(3 * 2 * 100)
    fence()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]       # one event per step boundary: shows a ramp (clocks, host) if there is one
    host = []
    sampler = ClockSampler(dev) if not os.environ.get("MISAMD_BENCH_NO_CLOCK") else None
    if sampler is not None:
        sampler.start()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        if i in timed_steps:
            ops.PROFILE = prof
        step(i)
        ops.PROFILE = None
        marks[i + 1].record()
        host.append(time.perf_counter())
    fence()
    dt = time.perf_counter() - t0
    global LAST_CLOCK
    LAST_CLOCK = sampler.stop() if sampler is not None else None
    ops.tile_queue_check()          # (outside the timed region) a persistent launch that lost tiles to dirty queue counters would make the figure meaningless: fail loudly
    global LAST_STEP_TRACE
    LAST_STEP_TRACE = {"gpu_ms_each": [round(marks[i].elapsed_time(marks[i + 1]), 2) for i in range(steps)],
                       "host_enqueue_ms_each": [round((host[i] - (host[i - 1] if i else t0)) * 1e3, 2) for i in range(steps)]}
    prof = (prof, len(timed_steps)) if timing else None
    if DDP_ON:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    return dt, prof


def _comm_report(reducer, eng, steps, world, dist, dev):
    """multi-GPU self-description: RCCL really carried the gradients, how long the buckets took, how much of that was exposed, and whether
    every rank ends the run with bit-identical parameters (same init + same summed gradients + same optimizer => must be equal)"""
    rep = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend() + (" + C-ABI communicator (mis_allreduce_bucket)" if reducer.backend == "native" else ""),
           "allreduce_bytes_per_step": int(eng.flat.total * 4),
           "buckets_per_step": reducer.buckets_per_step}
    ar, exposed = reducer.timing_ms()
    rep["allreduce_ms_per_step"] = round(ar / steps, 3)
    rep["exposed_comm_ms_per_step"] = round(exposed / steps, 3)
    rep["per_bucket"] = reducer.timing_breakdown()          # issue order (head first, biases last): bytes, all-reduce ms, exposed ms (the part after the backward ran out)
    from mdeical_image_segmentation_amd import ops
    rep["persistent_grid_blocks"] = ops.dispatch_switch("MIS_PERSIST_CUS")      # 256 = every CU; fewer leaves CUs to the RCCL kernels (csrc/dispatch_cfg.hpp)
    rep["tile_queue"] = not ops.dispatch_switch("MIS_TILEQ_OFF")      # the persistent conv kernels share their tiles dynamically: a collective that holds CUs does not stall a launch
    h = torch.stack([eng.flat.p.double().sum(), eng.flat.p.double().abs().sum(), eng.flat.g.double().sum()])
    hs = [torch.zeros_like(h) for _ in range(world)]
    dist.all_gather(hs, h)
    rep["params_identical_across_ranks"] = bool(all(torch.equal(hs[0][:2], x[:2]) for x in hs))
    rep["grads_identical_across_ranks"] = bool(all(torch.equal(hs[0][2], x[2]) for x in hs))
    return rep


def run_dropin2d(args, dev, *, batch, size, steps, warmup):
    """The drop-in path a user of the reference executes: `UNetModel(UNetConfig(1, 2, "UNet"))` - the nn.Module whose forward / backward are ONE
    torch.autograd.Function over the fused engine - driven like the HF Trainer step CustomTrainer plugs into (reference trainer/MYtrainer.py:6-11, train.py:147-158):
    `loss = model(images=..., labels=...).loss; loss.backward(); clip_grad_norm_(1.0); torch AdamW (the Trainer's decay / no-decay split)`, same batch shape, dtype
    and data as the headline loop, timed the same way.  Reported beside the engine's own loop: the difference is the price of the nn.Module / autograd hand-over."""
    import mdeical_image_segmentation_amd.dropin as dropin
    dropin.install()
    from unet2d import UNetConfig, UNetModel
    torch.manual_seed(0)
    m = UNetModel(UNetConfig(in_channels=1, out_channels=2, unet_type="UNet", compute_dtype="bf16")).to(dev).train()
    decay = [p for n, p in m.named_parameters() if not n.endswith("bias")]
    nodecay = [p for n, p in m.named_parameters() if n.endswith("bias")]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 1e-3}, {"params": nodecay, "weight_decay": 0.0}], lr=args.lr, fused=True)
    g = torch.Generator().manual_seed(1000)
    images = torch.randn(batch, 1, size, size, generator=g).to(dev)
    labels = torch.randint(0, 2, (batch, size, size), generator=g).to(dev)
    losses = torch.zeros(max(steps, 1), dtype=torch.float32, device=dev)

    def step(i):
        opt.zero_grad(set_to_none=True)
        out = m(images=images, labels=labels)
        if i >= 0:
            losses[i:i + 1].copy_(out.loss.detach().reshape(1), non_blocking=True)
        out.loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()

    for _ in range(max(warmup, 1)):
        step(-1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 2), "unit": "images/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "warmup": max(warmup, 1), "dtype": "bf16",
            "loss_per_step": [round(float(v), 5) for v in losses[:steps].cpu().tolist()],
            "config": {"workload": f"UNetModel(UNetConfig(1, 2, 'UNet', compute_dtype='bf16')) bs={batch} {size}x{size}: model(images, labels).loss.backward() + "
                                   f"clip_grad_norm_(1.0) + torch.optim.AdamW(fused, lr {args.lr:g}, wd 1e-3 on weights): the step CustomTrainer / HF Trainer runs per batch"}}


def run2d(args, rank, world, dev, dist, *, dtype, batch, size, steps, warmup, timing, layers):
    from mdeical_image_segmentation_amd.ddp import GradReducer
    from mdeical_image_segmentation_amd.engine2d import UNet2DEngine
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    cin, ncls = (1, 2) if args.net == "1x2" else (3, 4)
    eng = UNet2DEngine(cin, ncls, dtype=tdt, device=dev, seed=0, lr=args.lr)  # identical init on every rank (PyTorch default init, seed 0)
    reducer = GradReducer(eng.flat, timing=True, backend=args.comm) if DDP_ON else None
    g = torch.Generator().manual_seed(1000 + rank)                    # per-rank data shard
    images = torch.randn(batch, cin, size, size, generator=g).to(dev)
    labels = torch.randint(0, ncls, (batch, size, size), generator=g).to(dev)
    losses = torch.zeros(max(steps, 1), dtype=torch.float32, device=dev)

    def step(i):
        eng.forward(images, labels, train=True, grad_scale=1.0 / world)
        if i >= 0:
            losses[i:i + 1].copy_(eng.loss_buf[:1], non_blocking=True)        # device-side, 4 bytes, no synchronisation
        if reducer is None:
            eng.backward()
        else:
            eng.backward(stage_cb=reducer.stage_done)
            reducer.finish()
        eng.optimizer_step()

    def acts():
        return live_fractions({"down_conv.0.second": eng.cat[0][..., 64:], "down_conv.2.second": eng.cat[2][..., 256:],
                               "middle_conv.second": eng.m2, "up_conv.3.first": eng.u1[3]})

    warmup = max(warmup, 1)           # the live-activation probe needs one completed step
    step(-1)
    live0 = acts()
    if reducer is not None:
        reducer.reset_timing()
    dt, prof = _timed_loop(step, steps, warmup - 1, world, dist, dev, timing)
    live1 = acts()
    loss_list = [round(float(v), 5) for v in losses[:steps].cpu().tolist()]
    out = None
    comm = _comm_report(reducer, eng, steps, world, dist, dev) if reducer is not None else None
    if rank == 0:
        ms = dt / steps * 1e3
        value = world * batch * steps / dt
        out = {
            "metric": "images/sec (2D 512x512 U-Net train step)", "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"unet2d {cin}-ch->{ncls}-class, bs={batch}/GPU {size}x{size}, "
                                   f"fwd+CE loss+bwd+clip_grad_norm(1.0)+AdamW(lr {args.lr:g} constant, wd 1e-3 on weights), PyTorch-default init (seed 0), N(0,1) images",
                       "global_batch": world * batch, "parallelism": f"dp{world}", "final_loss": loss_list[-1] if loss_list else None},
            "loss_per_step": loss_list, "step_trace": LAST_STEP_TRACE, "clock": LAST_CLOCK,
            "act_nonzero_frac": {"before_timed_steps": live0, "after_timed_steps": live1},
        }
        out["model_tflops"] = round(value * (FLOP_PER_IMAGE_512 if args.net == "1x2" else FLOP_PER_IMAGE_512_3X4) * (size / 512.0) ** 2 / 1e12, 1)
        if prof:
            peak = PEAK_BF16_TFLOPS if dtype == "bf16" else PEAK_F32_TFLOPS
            roof, kernels, total, ltab, psteps = kernel_tables(prof, steps, peak)
            roof["steps_with_launch_events"] = psteps
            if dtype == "bf16":
                attach_traffic(roof, batch, size)
            attach_clock(roof)
            out["roofline"], out["kernels"], out["mfma_kernel_ms_per_step"] = roof, kernels, total
            if layers:
                for key, v in sorted(ltab.items(), key=lambda kv: -kv[1][1]):
                    print(f"{v[1] / psteps * 1e3:8.3f} ms/step {v[0] / v[1] / 1e12:7.1f} TF/s x{v[2] // psteps}  {' '.join(k for k in key if k)}", file=sys.stderr)
        if comm is not None:
            out["comm"] = comm
        dead = {k: v for k, v in {**live0, **{k + "@end": v for k, v in live1.items()}}.items() if v < MIN_LIVE_FRACTION}
        bad_loss = [v for v in loss_list if not (v == v) or v > 50.0]
        if dead or bad_loss:
            print(json.dumps(out), file=sys.stderr)
            raise SystemExit(f"bench.py: the network collapsed during the timed steps (live fractions {dead}, losses {bad_loss}): number invalid")
    return out


def run3d(args, rank, world, dev, dist, *, dtype, batch, size, steps, warmup, timing, layers):
    import numpy as np

    from mdeical_image_segmentation_amd.augment.unet3d_augment import transforms as tr
    from mdeical_image_segmentation_amd.ddp import GradReducer
    from mdeical_image_segmentation_amd.engine3d import UNet3DEngine
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    eng = UNet3DEngine(1, 3, dtype=tdt, device=dev, seed=0, lr=args.lr)
    reducer = GradReducer(eng.flat, timing=True, backend=args.comm) if DDP_ON else None
    g = torch.Generator().manual_seed(1000 + rank)
    x = torch.randn(batch, 1, size, size, size, generator=g).to(dev)
    t = (torch.rand(batch, 3, size, size, size, generator=g) > 0.5).float().to(dev)

    # on-device augmentation inside the timed step: flip + rot90 + rotate (+-30 deg, reflect; cubic spline
    # order 3 on raw, order 0 on targets) on raw and targets in lock-step, contrast (p=1) + Gaussian noise (p=1) on raw; parameters from the reference's streams
    tr.GLOBAL_RANDOM_STATE = np.random.RandomState(47 + rank)

    def geo_(order):
        return [{"name": "RandomFlip"}, {"name": "RandomRotate90"},
                {"name": "RandomRotate", "axes": [[2, 1]], "angle_spectrum": 30, "mode": "reflect", "order": order}]
    tf = tr.Transformer({"raw": geo_(3) + [{"name": "RandomContrast", "execution_probability": 1.0},
                                           {"name": "AdditiveGaussianNoise", "execution_probability": 1.0, "scale": [0.0, 0.1]}],
                         "label": geo_(0)}, {"mean": 0.0, "std": 1.0})
    rt, lt = tf.raw_transform(), tf.label_transform()
    losses = torch.zeros(max(steps, 1), dtype=torch.float32, device=dev)
    # The augmentation is the input pipeline of the step: it runs on its OWN HIP stream, one batch ahead, into two alternating buffer pairs - every train step still
    # augments exactly one batch inside the timed region, but the transforms' host-side read-backs (the noise transform advances its numpy RandomState from a device
    # result: 2.6 KB) wait for the short augmentation queue only, never for the train step that is still executing on the main stream.
    aug_stream = torch.cuda.Stream(device=dev)
    bufs = [(torch.empty_like(x), torch.empty_like(t)) for _ in range(2)]
    ready = [torch.cuda.Event() for _ in range(2)]
    free = [torch.cuda.Event() for _ in range(2)]
    state = {"n": 0}

    no_aug = bool(os.environ.get("MISAMD_BENCH_NOAUG"))          # diagnostic A/B only (what the on-device augmentation costs the step): the line is then NOT cfg4 / cfg5

    def augment_into(k):
        xa, ta = bufs[k]
        if no_aug:
            aug_stream.wait_event(free[k])
            with torch.cuda.stream(aug_stream):
                xa.copy_(x)
                ta.copy_(t)
                ready[k].record(aug_stream)
            return
        aug_stream.wait_event(free[k])                      # the step that last trained on this pair is past its backward pass
        with torch.cuda.stream(aug_stream):
            for b in range(batch):
                xa[b, 0] = rt(x[b, 0])
                ta[b] = lt(t[b])
            ready[k].record(aug_stream)

    for k in range(2):
        free[k].record(torch.cuda.current_stream(dev))
    augment_into(0)

    def step(i):
        k = state["n"] & 1
        state["n"] += 1
        main = torch.cuda.current_stream(dev)
        main.wait_event(ready[k])
        xa, ta = bufs[k]
        eng.forward(xa, ta, train=True, grad_scale=1.0 / world)
        if i >= 0:
            losses[i:i + 1].copy_(eng.loss_buf[:1], non_blocking=True)
        if reducer is None:
            eng.backward()
        else:
            eng.backward(stage_cb=reducer.stage_done)
            reducer.finish()
        free[k].record(main)                                # (the first layer's weight gradient read the augmented volume)
        eng.optimizer_step()
        # the next step's batch, on the side stream - enqueued AFTER this step's kernels: the transforms block the host on small read-backs (the noise transform advances
        # its numpy RandomState from a device result), and with the augmentation enqueued first the host sat in those waits while the main stream ran dry - the device
        # idled ~3 ms at every step boundary (`step_trace`: host enqueue 52 ms per step = the whole step).  Now the host waits while the device works on this step.
        augment_into(k ^ 1)

    warmup = max(warmup, 1)
    step(-1)
    if reducer is not None:
        reducer.reset_timing()
    dt, prof = _timed_loop(step, steps, warmup - 1, world, dist, dev, timing)
    loss_list = [round(float(v), 5) for v in losses[:steps].cpu().tolist()]
    comm = _comm_report(reducer, eng, steps, world, dist, dev) if reducer is not None else None
    if comm is not None:
        comm["exact_dice"] = bool(eng.exact_dice and world > 1)          # the Dice term of the GLOBAL batch (36-byte all-reduce inside the head), as the reference's DataParallel
    out = None
    if rank == 0:
        value = world * batch * steps / dt
        out = {"metric": f"volumes/sec (3D {size}^3 U-Net train step)", "value": round(value, 3), "unit": "volumes/s", "n_gpus": world,
               "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 2), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
               "config": {"workload": ("[DIAGNOSTIC: augmentation replaced by a copy, MISAMD_BENCH_NOAUG] " if no_aug else "") + f"unet3d 1-ch->3-class bs={batch}/GPU {size}^3, on-device augment (flip+rot90+rotate[order 3 raw / 0 targets]+contrast+"
                                      f"Gaussian noise from the reference's own MT19937 stream; one batch ahead on a second HIP stream) + "
                                      f"fwd+BCEDice+bwd+clip+AdamW(lr {args.lr:g}), random-init weights",
                          "global_batch": world * batch, "parallelism": f"dp{world}", "final_loss": loss_list[-1] if loss_list else None},
               "loss_per_step": loss_list, "step_trace": LAST_STEP_TRACE, "clock": LAST_CLOCK}
        out["model_tflops"] = round(value * FLOP_PER_VOLUME_128 * (size / 128.0) ** 3 / 1e12, 1)
        if prof:
            peak = PEAK_BF16_TFLOPS if dtype == "bf16" else PEAK_F32_TFLOPS
            roof, kernels, total, ltab, psteps = kernel_tables(prof, steps, peak)
            roof["steps_with_launch_events"] = psteps
            if dtype == "f32" and world == 1:
                from mdeical_image_segmentation_amd import _lib
                attach_traffic(roof, batch, size, "traffic_3d_f32.json", _lib.TRAFFIC_SOURCES_3D_F32)
            attach_clock(roof, 2400.0)
            out["roofline"], out["kernels"], out["mfma_kernel_ms_per_step"] = roof, kernels, total
            if layers:
                for k, v in sorted(ltab.items(), key=lambda kv: -kv[1][1]):
                    print(f"{v[1] / psteps * 1e3:9.3f} ms/step {v[0] / v[1] / 1e12:7.1f} TF/s x{v[2] // psteps}  {' '.join(x for x in k if x)}", file=sys.stderr)
        if comm is not None:
            out["comm"] = comm
    return out


if __name__ == "__main__":
    main()
