import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops
N, H, W = 32, 512, 512
x = torch.randn(N, H, W, 64, device="cuda").bfloat16()
w = (torch.randn(9, 64, 64, device="cuda") * 0.04).bfloat16()
b = torch.randn(64, device="cuda")
y = torch.empty(N, H, W, 64, device="cuda", dtype=torch.bfloat16)
m = torch.randn(N, H, W, 64, device="cuda").bfloat16()
def run(mask):
    for _ in range(3):
        ops.conv_igemm(x, w, y, ksize=3, Cin=64, Cout=64, bias=b, relu=True, mask=mask)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv_igemm(x, w, y, ksize=3, Cin=64, Cout=64, bias=b, relu=True, mask=mask)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * N * H * W * 9 * 64 * 64
    return ms, fl / ms / 1e9
print(os.environ.get("MIS_WS64_DEBUG", "0"), "fwd  %.3f ms %.0f TF" % run(None), "| dgrad+mask %.3f ms %.0f TF" % run(m))
