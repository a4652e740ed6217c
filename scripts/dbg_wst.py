import os, sys, torch
sys.path.insert(0, os.getcwd())
from mdeical_image_segmentation_amd import ops
torch.manual_seed(0)
N,H,W,Cin,Cout = 2,64,64,64,128
x = torch.randn(N,H,W,Cin,device='cuda').to(torch.bfloat16)
dy = torch.randn(N,H,W,Cout,device='cuda').to(torch.bfloat16)
dw0 = torch.empty(Cout,Cin,3,3,device='cuda'); dw1 = torch.empty_like(dw0)
ops.dispatch_override("MIS_WGRAD_PP_NOSTREAM", 1)
ops.wgrad(x,dy,dw0,ksize=3,Cin=Cin,Cout=Cout); print(ops.wgrad_last_dispatch())
ops.dispatch_override("MIS_WGRAD_PP_NOSTREAM", -1)
ops.wgrad(x,dy,dw1,ksize=3,Cin=Cin,Cout=Cout); print(ops.wgrad_last_dispatch())
d = (dw1-dw0).abs()
print("per tap max err:", d.amax((0,1)).cpu())
print("per tap ref max:", dw0.abs().amax((0,1)).cpu())
print("per co-block(16) max err:", d.view(8,16,Cin,9).amax((1,2,3)).cpu())
print("per ci-block(16) max err:", d.view(Cout,4,16,9).amax((0,2,3)).cpu())
# locality: x nonzero only in one row
for row in (0, 3, 4, 7, 8, 60):
    x2 = torch.zeros_like(x); x2[:, row] = x[:, row]
    ops.dispatch_override("MIS_WGRAD_PP_NOSTREAM", 1); ops.wgrad(x2,dy,dw0,ksize=3,Cin=Cin,Cout=Cout)
    ops.dispatch_override("MIS_WGRAD_PP_NOSTREAM", -1); ops.wgrad(x2,dy,dw1,ksize=3,Cin=Cin,Cout=Cout)
    print("x row", row, "per tap err:", [round(v, 3) for v in (dw1-dw0).abs().amax((0,1)).cpu().flatten().tolist()])
for col in (0, 1, 30, 31, 32, 33, 63):
    x2 = torch.zeros_like(x); x2[:, 0, col] = x[:, 0, col]
    ops.dispatch_override("MIS_WGRAD_PP_NOSTREAM", 1); ops.wgrad(x2,dy,dw0,ksize=3,Cin=Cin,Cout=Cout)
    ops.dispatch_override("MIS_WGRAD_PP_NOSTREAM", -1); ops.wgrad(x2,dy,dw1,ksize=3,Cin=Cin,Cout=Cout)
    print("x row 0 col", col, "per tap err:", [round(v, 3) for v in (dw1-dw0).abs().amax((0,1)).cpu().flatten().tolist()])
