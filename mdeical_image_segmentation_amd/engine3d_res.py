"""Fused residual 3-D U-Net engine on libmisamd (MI355X): the reference's `ResidualUNet3D` (model/unet3d/model.py:197-232) with its
defaults - `ResNetBlock` basic module (buildingblocks.py:255-325) with layer order 'gcr': a 1x1x1 conv (bias) when the channel count changes,
SingleConv 'gcr' (GroupNorm -> Conv3d -> ReLU), SingleConv 'gc', `out += residual`, ReLU; MaxPool3d between encoders; decoders = transposed
conv upsampling (k3, s2, p1, resized to the encoder grid), SUM joining with the encoder features, ResNetBlock without the 1x1x1 conv
(buildingblocks.py:498-534); 1x1x1 head, BCE+Dice in the head kernel.

Reuses the GroupNorm-folded conv / wgrad machinery, the transposed-conv GEMM + gathers and the optimizer of engine3d.UNet3DEngine; new here
are the residual joins (`mis_add_act`) and the first block's single-input-channel 1x1x1 conv (`mis_expand1_*`)."""
import math

import os

import torch

from . import ops
from ._lib import MisError
from .engine2d import FlatParams
from .engine3d import UNet3DEngine, _SC, _ct_name
from .ops import View


def _se_specs(pre, co):
    """parameters of `se_module` = ChannelSpatialSELayer3D(co, reduction_ratio=1) in registration order (se.py:18-116)"""
    m = f"{pre}.se_module"
    return [(f"{m}.cSE.fc1.weight", (co, co)), (f"{m}.cSE.fc1.bias", (co,)), (f"{m}.cSE.fc2.weight", (co, co)), (f"{m}.cSE.fc2.bias", (co,)),
            (f"{m}.sSE.conv.weight", (1, co, 1, 1, 1)), (f"{m}.sSE.conv.bias", (1,))]


def resunet3d_param_specs(in_channels, out_channels, f_maps, se=False):
    specs = []
    for i, co in enumerate(f_maps):
        ci = in_channels if i == 0 else f_maps[i - 1]
        pre = f"encoders.{i}.basic_module"
        specs += [(f"{pre}.conv1.weight", (co, ci, 1, 1, 1)), (f"{pre}.conv1.bias", (co,))]
        for k in (2, 3):
            specs += [(f"{pre}.conv{k}.groupnorm.weight", (co,)), (f"{pre}.conv{k}.groupnorm.bias", (co,)), (f"{pre}.conv{k}.conv.weight", (co, co, 3, 3, 3))]
        if se:
            specs += _se_specs(pre, co)
    rf = list(reversed(f_maps))
    for j in range(len(rf) - 1):
        co = rf[j + 1]
        specs.append((_ct_name(j), (rf[j], co, 3, 3, 3)))
        pre = f"decoders.{j}.basic_module"
        for k in (2, 3):
            specs += [(f"{pre}.conv{k}.groupnorm.weight", (co,)), (f"{pre}.conv{k}.groupnorm.bias", (co,)), (f"{pre}.conv{k}.conv.weight", (co, co, 3, 3, 3))]
        if se:
            specs += _se_specs(pre, co)
    specs += [("final_conv.weight", (out_channels, f_maps[0], 1, 1, 1)), ("final_conv.bias", (out_channels,))]
    return specs


def default_init_res_(params, seed=None):
    """Same RNG stream as `torch.manual_seed(seed); ResidualUNet3D(...)`: every conv / transposed conv draws kaiming_uniform(a=sqrt(5)) for
    its weight and, when it has a bias, U(+-1/sqrt(fan_in)); GroupNorm draws nothing."""
    if seed is not None:
        torch.manual_seed(seed)
    for name, p in params.items():
        if name.endswith("groupnorm.weight"):
            p.fill_(1.0)
        elif name.endswith("groupnorm.bias"):
            p.zero_()
        elif name.endswith(".weight"):
            w = torch.empty(p.shape)
            torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            p.copy_(w)
            bname = name[:-len("weight")] + "bias"
            if bname in params:
                fan_in = w[0].numel()
                b = torch.empty(params[bname].shape)
                bound = 1.0 / math.sqrt(fan_in)
                torch.nn.init.uniform_(b, -bound, bound)
                params[bname].copy_(b)


class ResidualUNet3DEngine(UNet3DEngine):
    SE = False

    def __init__(self, in_channels=1, out_channels=3, f_maps=(64, 128, 256, 512, 1024), num_groups=8, dtype=torch.float32, device="cuda", seed=None,
                 lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3, max_grad_norm=1.0, alpha=1.0, beta=1.0):
        if in_channels != 1:
            raise MisError("ResidualUNet3DEngine: in_channels must be 1")
        if not (1 <= out_channels <= 4):
            raise MisError("ResidualUNet3DEngine: out_channels must be 1..4")
        f_maps = list(f_maps)
        if f_maps[0] != 64 or any(f % 64 for f in f_maps):
            raise MisError("ResidualUNet3DEngine: f_maps must start at 64 and be multiples of 64")
        ops.load()
        with torch.cuda.device(torch.device(device)):
            ops.tile_queue_init()          # the tile queue's counter pool exists before the first launch (and before any graph capture)
        self.deconv = True
        self.cin, self.cout, self.f_maps, self.G = in_channels, out_channels, f_maps, num_groups
        self.dtype, self.device = dtype, torch.device(device)
        self.side_reduce = os.environ.get("MISAMD_NO_SIDE_REDUCE") is None
        self.materialize = dtype == torch.bfloat16 and os.environ.get("MISAMD_GN_FOLD") is None      # see UNet3DEngine
        self.gn_from_dw = False                                                                      # (the residual engines keep the statistics kernel)
        self.levels = len(f_maps)
        self.specs = resunet3d_param_specs(in_channels, out_channels, f_maps, se=self.SE)
        self.flat = FlatParams(self.specs, self.device, lambda n: not n.endswith("bias"))
        self.P, self.Gr = self.flat.param, self.flat.grad
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.alpha, self.beta = alpha, beta
        self.step_count = 0
        host = {n: torch.empty(s) for n, s in self.specs}
        default_init_res_(host, seed)
        for n in host:
            self.P[n].copy_(host[n])
        rf = list(reversed(f_maps))
        self.ct = []
        for j in range(len(rf) - 1):
            t = _SC()
            t.name, t.cin, t.cout = _ct_name(j), rf[j], rf[j + 1]
            t.w2d = torch.empty(27 * t.cout, t.cin, 1, device=self.device)
            t.wf = torch.empty(1, 27 * t.cout, t.cin, dtype=dtype, device=self.device)
            t.wd = torch.empty(1, t.cin, 27 * t.cout, dtype=dtype, device=self.device)
            t.dw2d = torch.empty(27 * t.cout, t.cin, device=self.device)
            self.ct.append(t)
        self.sc = {}
        blocks = [(f"encoders.{i}.basic_module", co) for i, co in enumerate(f_maps)] + [(f"decoders.{j}.basic_module", rf[j + 1]) for j in range(len(rf) - 1)]
        for pre, co in blocks:
            for k in (2, 3):
                s = _SC()
                s.name, s.cin, s.cout, s.cin_pad = f"{pre}.conv{k}", co, co, co
                s.groups = 1 if co < num_groups else num_groups
                s.first = False
                s.relu = k == 2                                   # conv3 is 'gc': the non-linearity comes after the residual add
                s.wf = torch.empty(27, co, co, dtype=dtype, device=self.device)
                s.wd = torch.empty(27, co, co, dtype=dtype, device=self.device)
                s.wpad = s.dwpad = None
                self.sc[s.name] = s
        self.c1 = {}                                              # 1x1x1 convs of encoders 1.. (encoder 0 has a single input channel)
        for i in range(1, self.levels):
            c = _SC()
            c.name, c.cin, c.cout = f"encoders.{i}.basic_module.conv1", f_maps[i - 1], f_maps[i]
            c.wf = torch.empty(1, c.cout, c.cin, dtype=dtype, device=self.device)
            c.wd = torch.empty(1, c.cin, c.cout, dtype=dtype, device=self.device)
            self.c1[i] = c
        self.partials = torch.zeros(ops.sumsq_npartials(self.flat.total), dtype=torch.float32, device=self.device)
        self.gradnorm = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.loss_buf = torch.zeros(32, dtype=torch.float32, device=self.device)
        self._shape = None
        self.repack()

    def repack(self):
        super().repack()
        for c in self.c1.values():
            ops.pack_conv_weight(self.P[c.name + ".weight"].view(c.cout, c.cin, 1), c.wf, c.wd)

    def _alloc(self, N, D, H, W):
        if self._shape == (N, D, H, W):
            return
        L, fm = self.levels, self.f_maps
        div = 1 << (L - 1)
        if D % div or H % div or W % div:
            raise MisError(f"the fused residual 3-D engine needs D, H, W divisible by {div}; got {D}x{H}x{W}")
        dt, dev = self.dtype, self.device

        def buf(l, c):
            return torch.empty(N, D >> l, H >> l, W >> l, c, dtype=dt, device=dev)

        self.r, self.t, self.u, self.e, self.pooled = [], [], [], [], []
        self.g_r, self.g_t, self.g_e, self.g_pooled = [], [], [], []
        for l in range(L):
            for lst in (self.r, self.t, self.u, self.e, self.g_r, self.g_t, self.g_e):
                lst.append(buf(l, fm[l]))
            if l < L - 1:
                self.pooled.append(buf(l + 1, fm[l]))
                self.g_pooled.append(buf(l + 1, fm[l]))
        self.up, self.g_up, self.xj, self.t_dec, self.u_dec, self.d, self.g_d, self.g_t_dec = ([] for _ in range(8))
        for j in range(L - 1):
            l = L - 2 - j
            for lst in (self.up, self.g_up, self.xj, self.t_dec, self.u_dec, self.d, self.g_d, self.g_t_dec):
                lst.append(buf(l, fm[l]))
        for s in self.sc.values():
            cs = s.cin
            s.scale, s.shift = torch.zeros(N, cs, device=dev), torch.zeros(N, cs, device=dev)
            s.mean, s.rstd = torch.zeros(N, s.groups, device=dev), torch.zeros(N, s.groups, device=dev)
            s.S1, s.S2 = torch.zeros(N, cs, device=dev), torch.zeros(N, cs, device=dev)
            s.p, s.q, s.r = torch.zeros(N, cs, device=dev), torch.zeros(N, cs, device=dev), torch.zeros(N, cs, device=dev)
            s.sum0, s.sq0 = torch.zeros(N, cs, device=dev), torch.zeros(N, cs, device=dev)
            s.sum1, s.sq1 = torch.zeros(N, cs, device=dev), torch.zeros(N, cs, device=dev)
            s.xn = buf(self._level(s.name), s.cin_pad) if self.materialize else None
        self.dyn = {}
        for s in self.sc.values():
            key = (self._level(s.name), s.cin_pad)
            if key not in self.dyn:
                self.dyn[key] = buf(key[0], key[1])
        self.cols_elems = max(N * (D >> (l + 1)) * (H >> (l + 1)) * (W >> (l + 1)) * 27 * fm[l] for l in range(L - 1))
        self.cols = torch.empty(self.cols_elems, dtype=dt, device=dev)
        self.logits = torch.empty(N, self.cout, D, H, W, dtype=torch.float32, device=dev)
        self.argmax = torch.empty(N, D, H, W, dtype=torch.uint8, device=dev)
        self._shape = (N, D, H, W)

    # ---- forward ---------------------------------------------------------------------------------------
    def _block_fwd(self, pre, r, t, u, out):
        """ResNetBlock after its conv1: out = relu(conv3(GN(relu(conv2(GN(r))))) + r)"""
        c = r.shape[-1]
        self._sc_fwd(self.sc[pre + ".conv2"], r, c, t)
        self._sc_fwd(self.sc[pre + ".conv3"], t, c, u)
        ops.add_act(u, r, out, relu=True)

    def forward(self, x, target=None, train=True, grad_scale=1.0):
        if x.dtype != torch.float32 or not x.is_contiguous() or x.device.type != "cuda" or x.dim() != 5 or x.shape[1] != 1:
            raise MisError("x must be a contiguous fp32 CUDA tensor (N, 1, D, H, W)")
        N, _, D, H, W = x.shape
        self._alloc(N, D, H, W)
        self._x = x
        P, L = self.P, self.levels
        ops.expand1_fwd(x, P["encoders.0.basic_module.conv1.weight"].view(-1), P["encoders.0.basic_module.conv1.bias"], self.r[0])
        self._block_fwd("encoders.0.basic_module", self.r[0], self.t[0], self.u[0], self.e[0])
        for l in range(1, L):
            ops.maxpool2_fwd(self.e[l - 1], self.pooled[l - 1])
            c = self.c1[l]
            ops.conv_igemm(self.pooled[l - 1], c.wf, self.r[l], ksize=1, Cin=c.cin, Cout=c.cout, bias=P[c.name + ".bias"])
            self._block_fwd(f"encoders.{l}.basic_module", self.r[l], self.t[l], self.u[l], self.e[l])
        low = self.e[L - 1]
        for j in range(L - 1):
            l = L - 2 - j
            self._ct_fwd(j, low)                                        # -> self.up[j]
            ops.add_act(self.e[l], self.up[j], self.xj[j], relu=False)   # sum joining (buildingblocks.py:546-550)
            self._block_fwd(f"decoders.{j}.basic_module", self.xj[j], self.t_dec[j], self.u_dec[j], self.d[j])
            low = self.d[j]
        wh = P["final_conv.weight"].view(self.cout, self.f_maps[0])
        bh = P["final_conv.bias"]
        feat = self.d[L - 2]
        if target is None:
            ops.head_loss(feat, wh, bh, loss=ops.LOSS_NONE, logits=self.logits, argmax=self.argmax)
            return None, self.logits, self.argmax
        if target.dtype != torch.float32 or tuple(target.shape) != (N, self.cout, D, H, W) or not target.is_contiguous():
            raise MisError("target must be contiguous fp32 (N, C, D, H, W)")
        kw = dict(loss=ops.LOSS_BCEDICE, labels=target, logits=self.logits, argmax=self.argmax, loss_out=self.loss_buf, alpha=self.alpha, beta=self.beta)
        if train:
            kw.update(dy=self.g_d[L - 2], dw=self.Gr["final_conv.weight"], db=self.Gr["final_conv.bias"], grad_scale=grad_scale)
        ops.head_loss(feat, wh, bh, **kw)
        return self.loss_buf[:1], self.logits, self.argmax

    # ---- backward --------------------------------------------------------------------------------------
    def _block_bwd(self, pre, g_pre, g_t, g_r):
        """g_pre = dL/d(pre-activation of the block output); returns in g_r the total dL/d(r) (conv path + residual path)"""
        self._sc_bwd(self.sc[pre + ".conv3"], g_pre, g_t, mask0=True)              # input t = a ReLU output
        self._sc_bwd(self.sc[pre + ".conv2"], g_t, g_r, mask0=False, add0=g_pre)   # r is not a ReLU output; + residual branch

    def backward(self, stage_cb=None):
        cb = self._stage_cb(stage_cb)
        L = self.levels
        cb(["final_conv"])
        for j in range(L - 2, -1, -1):
            self._block_bwd(f"decoders.{j}.basic_module", self.g_d[j], self.g_t_dec[j], self.g_up[j])
            # g_up[j] = dL/d(e_l + up_j): it is the gradient of the transposed-conv output AND the decoder's share of e_l's gradient
            self._ct_bwd(j, self.g_e[L - 1] if j == 0 else self.g_d[j - 1])
            cb([f"decoders.{j}"])
        for l in range(L - 1, -1, -1):
            if l < L - 1:
                j = L - 2 - l
                # g_e[l] <- relu_mask(e_l) * (scatter(g_pooled[l]) + decoder share)
                ops.maxpool2_bwd(self.e[l], self.g_pooled[l], self.g_e[l], add=self.g_up[j], relu_mask=True)
            self._block_bwd(f"encoders.{l}.basic_module", self.g_e[l], self.g_t[l], self.g_r[l])
            if l > 0:
                c = self.c1[l]
                ops.wgrad(self.pooled[l - 1], self.g_r[l], self.Gr[c.name + ".weight"].view(c.cout, c.cin, 1), ksize=1, Cin=c.cin, Cout=c.cout,
                          dbias=self.Gr[c.name + ".bias"])
                ops.conv_igemm(self.g_r[l], c.wd, self.g_pooled[l - 1], ksize=1, Cin=c.cout, Cout=c.cin)
            else:
                ops.expand1_bwd(self._x, self.g_r[0], self.Gr["encoders.0.basic_module.conv1.weight"].view(-1),
                                self.Gr["encoders.0.basic_module.conv1.bias"])
            cb([f"encoders.{l}"])
        ops.wgrad_join(self.device)


class ResidualUNetSE3DEngine(ResidualUNet3DEngine):
    """The reference's `ResidualUNetSE3D` (model/unet3d/model.py:235-280): ResidualUNet3D whose blocks end in a concurrent spatial-and-channel
    squeeze & excitation (`ResNetBlockSE`, se_module 'scse', buildingblocks.py:326-362) - csrc/se3d.hip.  The block's ReLU output goes to a side
    buffer and the SE output takes its place, so every consumer (pooling, joins, transposed convs, head) is unchanged; in the backward the
    consumers' masked gradient is turned into the block's pre-activation gradient in place before the residual block's own backward."""
    SE = True

    def _alloc(self, N, D, H, W):
        if self._shape == (N, D, H, W):
            return
        super()._alloc(N, D, H, W)
        L, fm, dev = self.levels, self.f_maps, self.device
        self.se = {}
        blocks = [(f"encoders.{l}.basic_module", l, fm[l]) for l in range(L)] + [(f"decoders.{j}.basic_module", L - 2 - j, fm[L - 2 - j]) for j in range(L - 1)]
        for pre, l, c in blocks:
            st = _SC()
            S = (D >> l) * (H >> l) * (W >> l)
            st.raw = torch.empty(N, D >> l, H >> l, W >> l, c, dtype=self.dtype, device=dev)
            for name in ("sum", "sq", "mean", "z1", "a", "da", "cross"):
                setattr(st, name, torch.zeros(N, c, device=dev))
            st.bgate, st.dq = torch.zeros(N, S, device=dev), torch.zeros(N, S, device=dev)
            self.se[pre] = st

    def _se_params(self, pre, table):
        m = pre + ".se_module"
        return (table[m + ".cSE.fc1.weight"], table[m + ".cSE.fc1.bias"], table[m + ".cSE.fc2.weight"], table[m + ".cSE.fc2.bias"],
                table[m + ".sSE.conv.weight"].view(-1), table[m + ".sSE.conv.bias"])

    def _block_fwd(self, pre, r, t, u, out):
        st = self.se[pre]
        super()._block_fwd(pre, r, t, u, st.raw)
        W1, b1, W2, b2, w, b0 = self._se_params(pre, self.P)
        ops.se_fwd(st.raw, out, W1, b1, W2, b2, w, b0, st)

    def _block_bwd(self, pre, g_pre, g_t, g_r):
        st = self.se[pre]
        W1, _, W2, _, w, _ = self._se_params(pre, self.P)
        dW1, db1, dW2, db2, dw, db0 = self._se_params(pre, self.Gr)
        ops.se_bwd(g_pre, st.raw, W1, W2, w, st, dW1, db1, dW2, db2, dw, db0)
        super()._block_bwd(pre, g_pre, g_t, g_r)
