"""What a concurrent kernel that holds a few CUs (an RCCL all-reduce on the side stream) does to the persistent MFMA kernels on the main stream:
    bash scripts/hog_probe.sh            (builds scripts/cu_hog.hip into a scratch library, then runs this file)
For each layer: ms per launch alone, and with `H` CUs held by the hog for the whole measurement."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops  # noqa: E402

hog = ctypes.CDLL(os.environ["CU_HOG_LIB"])
hog.cu_hog.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
ops.load()
dev = "cuda"
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
BF = torch.bfloat16
HOGS = [int(v) for v in os.environ.get("HOG_CUS", "0,8,16").split(",")]


def timed(fn, hog_cus, iters=6):
    fn()
    torch.cuda.synchronize()
    if hog_cus:
        # ~100 MHz counter on gfx950's s_memrealtime; __builtin_readcyclecounter = shader clock (~2 GHz): 40 M cycles ~ 20 ms, longer than the measurement
        hog.cu_hog(hog_cus, 40_000_000, sink.data_ptr(), side.cuda_stream)
        torch.cuda._sleep(2_000_000)          # let the hog become resident before the first launch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


g = torch.Generator(device=dev).manual_seed(1)
for H, Cin, Cout in ((64, 512, 512), (256, 128, 128), (512, 64, 64), (512, 64, 128)):
    x = torch.randn(32, H, H, Cin, device=dev, generator=g).to(BF)
    w = (torch.randn(9, Cout, Cin, device=dev, generator=g) * (9 * Cin) ** -0.5).to(BF)
    b = torch.randn(Cout, device=dev, generator=g)
    y = torch.empty(32, H, H, Cout, device=dev, dtype=BF)
    dy = torch.randn(32, H, H, Cout, device=dev, generator=g).to(BF)
    dw = torch.empty(Cout, Cin, 3, 3, device=dev)
    conv = lambda: ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, bias=b, relu=True)      # noqa: E731
    wg = lambda: ops.wgrad(x, dy, dw, ksize=3, Cin=Cin, Cout=Cout)      # noqa: E731
    rc = [timed(conv, h) for h in HOGS]
    ctag = ops.conv_last_dispatch()
    rw = [timed(wg, h) for h in HOGS]
    wtag = ops.wgrad_last_dispatch()[0]
    print(f"{H:4d}^2 {Cin:4d}->{Cout:<4d} conv [{ctag}] " + "  ".join(f"hog {h:2d}: {t:6.3f} ms" for h, t in zip(HOGS, rc)), flush=True)
    print(f"{'':16s} wgrad [{wtag}] " + "  ".join(f"hog {h:2d}: {t:6.3f} ms" for h, t in zip(HOGS, rw)), flush=True)
