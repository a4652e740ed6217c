"""Diagnostic: where the waves of conv_pp spend their cycles (needs a libmisamd.so built with -DMIS_PP_STAMPS; see csrc/conv_pp.hip).
    python scripts/pp_stamps.py H Cin Cout"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops  # noqa: E402

H, Cin, Cout = (int(v) for v in sys.argv[1:4])
N = 32
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(N, H, H, Cin, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(9, Cout, Cin, device="cuda", generator=g) * (9 * Cin) ** -0.5).to(torch.bfloat16)
b = torch.randn(Cout, device="cuda", generator=g)
y = torch.empty(N, H, H, Cout, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, bias=b, relu=True)
torch.cuda.synchronize()
lib = ops.load()
buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
assert getattr(lib, os.environ.get("STAMP_SYMBOL", "mis_debug_pp_stamps"))(buf.ctypes.data_as(C.c_void_p)) == 0
st = buf.reshape(256, 8, 8).astype(np.float64)
names = ["R rest/drain", "R barrier", "M work", "M barrier", "epilogue", "R dma issue", "R read issue", "R lgkm wait"]
print(ops.conv_last_dispatch(), f"{H}^2 {Cin}->{Cout}")
for grp, sl in (("group 0 (waves 0-3)", slice(0, 4)), ("group 1 (waves 4-7)", slice(4, 8))):
    m = st[:, sl].mean((0, 1))
    tot = m.sum()
    print(f"  {grp}: total {tot:12.0f} cycles/wave: " + "  ".join(f"{n} {100 * v / tot:5.1f}%" for n, v in zip(names, m)))
