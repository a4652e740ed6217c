"""Fused 2-D U-Net train-step engine on libmisamd (MI355X).

Topology, parameter names/shapes and semantics mirror the reference exactly
(model/unet2d/unet.py:42-128 `UNet`, :1184-1213 `UNetModel` criterion, trainer/MYtrainer.py:6-11 + the HF
Trainer inner step), but the execution is MI355X-first:

  * activations live channels-last (NHWC) in bf16 (or fp32 = the parity mode) and never leave HBM layouts the
    MFMA kernels consume; `torch.cat` / `center_crop` / `MaxPool2d` / `ConvTranspose2d` become addressing:
    the skip conv and the up-conv write straight into the two halves of one concat buffer;
  * forward = 23 implicit-GEMM launches + 4 pools + 1 fused head/loss/argmax(/backward) kernel;
  * backward = dgrad through the same implicit-GEMM kernel (mirrored weight pack, ReLU mask in the epilogue,
    the up-conv gradient is written pixel-UNshuffled so the transposed conv's dgrad/wgrad are plain GEMMs),
    wgrad = split-K MFMA GEMM with deterministic slab reduction, pool-bwd fused with the skip-gradient add;
  * optimizer = one global sum-of-squares + two AdamW launches over flat fp32 master buffers (weights | biases).

PyTorch is only the allocator and stream provider here.  No CPU fallback exists.
"""
import math

import os

import torch

from . import ops
from ._lib import OUT_PLAIN, OUT_SHUFFLE2, OUT_UNSHUFFLE2, MisError
from .ops import View

FEATS = (64, 128, 256, 512)


def unet2d_param_specs(in_channels, out_channels):
    """(name, shape) in the reference's registration order (model/unet2d/unet.py:47-89)."""
    specs = []

    def dc(prefix, ci, co):
        specs.extend([(f"{prefix}.first.weight", (co, ci, 3, 3)), (f"{prefix}.first.bias", (co,)),
                      (f"{prefix}.second.weight", (co, co, 3, 3)), (f"{prefix}.second.bias", (co,))])

    for i, (ci, co) in enumerate([(in_channels, 64), (64, 128), (128, 256), (256, 512)]):
        dc(f"down_conv.{i}", ci, co)
    dc("middle_conv", 512, 1024)
    for i, (ci, co) in enumerate([(1024, 512), (512, 256), (256, 128), (128, 64)]):
        specs.extend([(f"up_sample.{i}.up.weight", (ci, co, 2, 2)), (f"up_sample.{i}.up.bias", (co,))])
    for i, (ci, co) in enumerate([(1024, 512), (512, 256), (256, 128), (128, 64)]):
        dc(f"up_conv.{i}", ci, co)
    specs.extend([("final_conv.weight", (out_channels, 64, 1, 1)), ("final_conv.bias", (out_channels,))])
    return specs


def default_init_(params, seed=None):
    """PyTorch default Conv init (kaiming_uniform a=sqrt(5); bias U(+-1/sqrt(fan_in))) in registration order,
    i.e. the same RNG stream as `torch.manual_seed(seed); UNet(in, out)` in the reference."""
    if seed is not None:
        torch.manual_seed(seed)
    names = list(params.keys())
    for wn, bn in zip(names[0::2], names[1::2]):
        w, b = params[wn], params[bn]
        wc = torch.empty(w.shape, dtype=torch.float32)
        torch.nn.init.kaiming_uniform_(wc, a=math.sqrt(5))
        fan_in = wc.size(1) * wc[0][0].numel()
        bc = torch.empty(b.shape, dtype=torch.float32)
        bound = 1.0 / math.sqrt(fan_in)
        torch.nn.init.uniform_(bc, -bound, bound)
        w.copy_(wc)
        b.copy_(bc)


class FlatParams:
    """fp32 master parameters / grads / Adam moments in two contiguous regions: [weights | biases]."""

    def __init__(self, specs, device, decay_fn):
        self.specs = specs
        al = 64
        off = 0
        self.offsets = {}
        order = [s for s in specs if decay_fn(s[0])] + [s for s in specs if not decay_fn(s[0])]
        self.n_decay = 0
        for name, shape in order:
            n = 1
            for d in shape:
                n *= d
            self.offsets[name] = (off, n, shape)
            off += (n + al - 1) // al * al
            if decay_fn(name):
                self.n_decay = off
        self.total = off
        self.p = torch.zeros(off, dtype=torch.float32, device=device)
        self.g = torch.zeros(off, dtype=torch.float32, device=device)
        self.m = torch.zeros(off, dtype=torch.float32, device=device)
        self.v = torch.zeros(off, dtype=torch.float32, device=device)
        self.param = {n: self._view(self.p, n) for n, _ in specs}
        self.grad = {n: self._view(self.g, n) for n, _ in specs}

    def _view(self, flat, name):
        off, n, shape = self.offsets[name]
        return flat[off:off + n].view(shape)


class UNet2DEngine:
    def __init__(self, in_channels, out_channels, dtype=torch.bfloat16, device="cuda", seed=None,
                 lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3, max_grad_norm=1.0):
        if dtype not in (torch.bfloat16, torch.float32):
            raise MisError("dtype must be torch.bfloat16 or torch.float32")
        if not (1 <= in_channels <= 4):
            raise MisError("in_channels must be 1..4")
        if not (1 <= out_channels <= 4):
            raise MisError("out_channels must be 1..4")
        ops.load()
        with torch.cuda.device(torch.device(device)):
            ops.tile_queue_init()          # the tile queue's counter pool exists before the first launch (and before any graph capture)
        self.cin, self.cout = in_channels, out_channels
        # wgrad's slab reductions can run on a second stream under the following dgrad kernel (MISAMD_SIDE_REDUCE=1).  Off by default since round 2: with
        # one slab per persistent block the reductions are small, and the measured step is the same either way (817.2 vs 817.4 img/s) - one stream less to order.
        self.side_reduce = os.environ.get("MISAMD_SIDE_REDUCE") is not None
        # round 4: the split-K slab reductions of a whole stage (decoder level / encoder level: 1-3 layers) run as TWO launches (ops.wgrad_reduce_batch) instead of three small
        # kernels per layer: 59 -> 20 launches, 0.83 -> 0.3 ms per step.  MISAMD_REDUCE_PER_LAYER=1: the per-layer kernels (A/B switch).
        self.batch_reduce = not self.side_reduce and os.environ.get("MISAMD_REDUCE_PER_LAYER") is None
        self._red = []
        self.dtype = dtype
        self.device = torch.device(device)
        self.specs = unet2d_param_specs(in_channels, out_channels)
        self.flat = FlatParams(self.specs, self.device, lambda n: not n.endswith("bias"))
        self.P, self.G = self.flat.param, self.flat.grad
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.step_count = 0
        self.loss_kind = ops.LOSS_CE if out_channels > 1 else ops.LOSS_BCE
        host = {n: torch.empty(s) for n, s in self.specs}
        default_init_(host, seed)
        for n in host:
            self.P[n].copy_(host[n])
        # packed MFMA operands
        self.wf, self.wd_ = {}, {}
        for n, s in self.specs:
            if not n.endswith("weight") or n in ("down_conv.0.first.weight", "final_conv.weight"):
                continue
            base = n[:-len(".weight")]
            if n.startswith("up_sample"):
                cin, cq = s[0], s[1]
                self.wf[base] = torch.empty(4 * cq, cin, dtype=dtype, device=self.device)
                self.wd_[base] = torch.empty(cin, 4 * cq, dtype=dtype, device=self.device)
            else:
                co, ci = s[0], s[1]
                self.wf[base] = torch.empty(9, co, ci, dtype=dtype, device=self.device)
                self.wd_[base] = torch.empty(9, ci, co, dtype=dtype, device=self.device)
        self.partials = torch.zeros(ops.sumsq_npartials(self.flat.total), dtype=torch.float32, device=self.device)
        self.gradnorm = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.loss_buf = torch.zeros(16, dtype=torch.float32, device=self.device)
        self._shape = None
        self.repack()

    # ---- parameters ------------------------------------------------------------------------------------
    def state_dict(self):
        return {n: self.P[n].detach().clone() for n, _ in self.specs}

    def load_state_dict(self, sd, prefix=""):
        for n, s in self.specs:
            t = sd[prefix + n]
            if tuple(t.shape) != tuple(s):
                raise MisError(f"{n}: shape {tuple(t.shape)} != {s}")
            self.P[n].copy_(t.to(torch.float32))
        self.repack()

    def repack(self):
        """fp32 master weights -> the packed MFMA operands of every layer, ONE launch (the flat parameter buffer and the operand buffers never move, so the table of
        their addresses is built once; MISAMD_REPACK_PER_LAYER=1: one launch per layer, the pre-batching path)"""
        if os.environ.get("MISAMD_REPACK_PER_LAYER"):
            for base, wf in self.wf.items():
                w = self.P[base + ".weight"]
                if base.startswith("up_sample"):
                    ops.pack_convt_weight(w, wf, self.wd_[base])
                else:
                    ops.pack_conv_weight(w, wf, self.wd_[base])
            return
        if getattr(self, "_pack_table", None) is None:
            self._pack_table = ops.PackTable([(self.P[base + ".weight"], wf, self.wd_[base], 1 if base.startswith("up_sample") else 0)
                                              for base, wf in self.wf.items()], self.device)
        ops.pack_batch(self._pack_table)

    # ---- buffers ---------------------------------------------------------------------------------------
    def _alloc(self, N, H, W):
        if self._shape == (N, H, W):
            return
        if H % 16 or W % 16:
            raise MisError("the fused engine needs H and W divisible by 16 (4 pooling levels); "
                           f"got {H}x{W}")
        dt, dev = self.dtype, self.device

        def buf(h, w, c):
            return torch.empty(N, h, w, c, dtype=dt, device=dev)

        self.t1, self.cat, self.pooled = [], [], []
        self.g_t1, self.g_skip, self.g_pooled = [], [], []
        for l, c in enumerate(FEATS):
            h, w = H >> l, W >> l
            self.t1.append(buf(h, w, c))
            self.cat.append(buf(h, w, 2 * c))
            self.pooled.append(buf(h // 2, w // 2, c))
            self.g_t1.append(buf(h, w, c))
            self.g_skip.append(buf(h, w, c))
            self.g_pooled.append(buf(h // 2, w // 2, c))
        hm, wm = H >> 4, W >> 4
        self.m1, self.m2 = buf(hm, wm, 1024), buf(hm, wm, 1024)
        self.g_m1, self.g_m2 = buf(hm, wm, 1024), buf(hm, wm, 1024)
        self.u1, self.u2, self.g_u1, self.g_u2, self.dys = [], [], [], [], []
        for j in range(4):
            l = 3 - j
            c = FEATS[l]
            h, w = H >> l, W >> l
            self.u1.append(buf(h, w, c))
            self.u2.append(buf(h, w, c))
            self.g_u1.append(buf(h, w, c))
            self.g_u2.append(buf(h, w, c))
            self.dys.append(buf(h // 2, w // 2, 4 * c))
        self.logits = torch.empty(N, self.cout, H, W, dtype=torch.float32, device=dev)
        self.argmax = torch.empty(N, H, W, dtype=torch.uint8, device=dev)
        # ReLU bits (csrc/relu_bits.hpp) of every activation whose sign the backward pass needs as a ReLU mask: written by the producing convolution's epilogue, read by
        # the masked dgrad instead of the bf16 tensor (1/16 of the bytes; MISAMD_BF16_MASK=1: the bf16 masks, the pre-round-3 path)
        self.use_bits = dt == torch.bfloat16 and not os.environ.get("MISAMD_BF16_MASK")
        self.rb = {}
        if self.use_bits:
            def bits(t):
                n_, h_, w_, c_ = t.shape
                return torch.empty(ops.relu_bits_bytes(n_, h_, w_, c_), dtype=torch.uint8, device=dev)
            for l in range(4):
                self.rb[("t1", l)] = bits(self.t1[l])
            for j in range(4):
                self.rb[("u1", j)] = bits(self.u1[j])
                if j < 3:
                    self.rb[("u2", j)] = bits(self.u2[j])
            self.rb["m1"], self.rb["m2"] = bits(self.m1), bits(self.m2)
        # "pool bits" (mis_maxpool2_fwd_pb): arg-max position + input sign per pooled element, so that the pooling backward does not read the skip tensor again
        self.pb = [None] * 4 if os.environ.get("MISAMD_NO_POOL_BITS") else [torch.empty(N, (H >> l) // 2, (W >> l) // 2, c, dtype=torch.uint8, device=dev)
                                                                           for l, c in enumerate(FEATS)]
        self._shape = (N, H, W)

    # ---- forward ---------------------------------------------------------------------------------------
    def _conv(self, x, name, y, cin, cout, rb=None):
        ops.conv_igemm(x, self.wf[name], y, ksize=3, Cin=cin, Cout=cout, bias=self.P[name + ".bias"], relu=True, relu_bits=self.rb.get(rb))

    def forward(self, images, labels=None, train=True, grad_scale=1.0):
        """images: fp32 NCHW on the device; labels: int64 (N,H,W) for CE / fp32 (N,1,H,W) for BCE.
        Returns (loss[1] device tensor or None, logits fp32 NCHW, argmax uint8)."""
        if images.dtype != torch.float32 or not images.is_contiguous() or images.device.type != "cuda":
            raise MisError("images must be a contiguous fp32 CUDA tensor (N, C, H, W)")
        N, Cin, H, W = images.shape
        if Cin != self.cin:
            raise MisError(f"expected {self.cin} input channels, got {Cin}")
        self._alloc(N, H, W)
        self._images = images
        ops.tile_queue_reset()          # a step never inherits tile-queue counters from an earlier launch (captured as a memset node)
        P = self.P
        for l, c in enumerate(FEATS):
            skip = View(self.cat[l], c, c)
            if l == 0:
                ops.first_conv_fwd(images, P["down_conv.0.first.weight"], P["down_conv.0.first.bias"], self.t1[0], relu_bits=self.rb.get(("t1", 0)))
            else:
                self._conv(self.pooled[l - 1], f"down_conv.{l}.first", self.t1[l], FEATS[l - 1], c, rb=("t1", l))
            self._conv(self.t1[l], f"down_conv.{l}.second", skip, c, c)
            # the pool bits are written by EVERY forward: `logits = model(images)` + an external loss + backward() (head_backward) is a supported path, and the
            # pooling backward reads these bytes instead of the skip tensor (they used to be written only with labels)
            ops.maxpool2_fwd(skip, self.pooled[l], pbits=self.pb[l])
        self._conv(self.pooled[3], "middle_conv.first", self.m1, 512, 1024, rb="m1")
        self._conv(self.m1, "middle_conv.second", self.m2, 1024, 1024, rb="m2")
        x = self.m2
        for j in range(4):
            l = 3 - j
            c = FEATS[l]
            ops.conv_igemm(x, self.wf[f"up_sample.{j}.up"], View(self.cat[l], 0, c), ksize=1, Cin=2 * c, Cout=4 * c,
                           bias=P[f"up_sample.{j}.up.bias"], relu=False, y0_mode=OUT_SHUFFLE2)
            self._conv(self.cat[l], f"up_conv.{j}.first", self.u1[j], 2 * c, c, rb=("u1", j))
            if j == 3 and labels is not None and train and self._fused_head(labels, N, H, W, grad_scale):
                return self.loss_buf[:1], self.logits, self.argmax
            self._conv(self.u1[j], f"up_conv.{j}.second", self.u2[j], c, c, rb=("u2", j))
            x = self.u2[j]
        self.features_valid = True
        wh = P["final_conv.weight"].view(self.cout, 64)
        bh = P["final_conv.bias"]
        if labels is None:
            ops.head_loss(self.u2[3], wh, bh, loss=ops.LOSS_NONE, logits=self.logits, argmax=self.argmax)
            return None, self.logits, self.argmax
        self._check_labels(labels, N, H, W)
        if train:
            ops.head_loss(self.u2[3], wh, bh, loss=self.loss_kind, labels=labels, logits=self.logits, argmax=self.argmax,
                          loss_out=self.loss_buf, dy=self.g_u2[3], dw=self.G["final_conv.weight"], db=self.G["final_conv.bias"],
                          grad_scale=grad_scale)
        else:
            ops.head_loss(self.u2[3], wh, bh, loss=self.loss_kind, labels=labels, logits=self.logits, argmax=self.argmax,
                          loss_out=self.loss_buf)
        return self.loss_buf[:1], self.logits, self.argmax

    def _fused_head(self, labels, N, H, W, grad_scale):
        """up_conv.3.second + final_conv + loss + their backward as ONE kernel (csrc/conv_ppd_head.hip, round 4): the last feature map is never written - dL/dfeatures
        (g_u2[3]) takes its place - and the head's pass over it disappears (0.58 ms of a 30.6 ms step).  bf16, 1 .. 4 classes (BCE / cross entropy), training form;
        everything else (and MISAMD_HEAD_UNFUSED=1 / MIS_HEAD_UNFUSED) runs the convolution and mis_head_loss separately."""
        if self.dtype != torch.bfloat16 or os.environ.get("MISAMD_HEAD_UNFUSED"):
            return False
        self._check_labels(labels, N, H, W)
        name = "up_conv.3.second"
        ok = ops.conv3x3_head_fused(self.u1[3], self.wf[name], self.P[name + ".bias"], self.g_u2[3], self.P["final_conv.weight"].view(self.cout, 64),
                                    self.P["final_conv.bias"], Cin=64, loss=self.loss_kind, labels=labels, logits=self.logits, argmax=self.argmax,
                                    loss_out=self.loss_buf, dw=self.G["final_conv.weight"], db=self.G["final_conv.bias"], grad_scale=grad_scale)
        if ok:
            self.features_valid = False          # u2[3] was not written: head_backward (an external gradient through the logits) regenerates it first
        return ok

    def _check_labels(self, labels, N, H, W):
        if self.loss_kind == ops.LOSS_CE:
            ok = labels.dtype == torch.int64 and tuple(labels.shape) == (N, H, W)
        else:
            ok = labels.dtype == torch.float32 and tuple(labels.shape) == (N, self.cout, H, W)
        if not ok or not labels.is_contiguous() or labels.device.type != "cuda":
            raise MisError("labels must be contiguous CUDA int64 (N,H,W) for C>1 or fp32 (N,C,H,W) for C==1; "
                           f"got {labels.dtype} {tuple(labels.shape)}")

    def head_backward(self, dlogits):
        """backward entry for an EXTERNAL loss: dlogits = dL/dlogits, fp32 (N, C, H, W); then call backward()."""
        wh = self.P["final_conv.weight"].view(self.cout, 64)
        if not getattr(self, "features_valid", True):          # the forward ran the fused head: the last feature map was never stored
            self._conv(self.u1[3], "up_conv.3.second", self.u2[3], 64, 64, rb=("u2", 3))
            self.features_valid = True
        ops.head_loss(self.u2[3], wh, self.P["final_conv.bias"], loss=ops.LOSS_EXTERNAL, labels=dlogits.contiguous(),
                      dy=self.g_u2[3], dw=self.G["final_conv.weight"], db=self.G["final_conv.bias"])

    # ---- backward --------------------------------------------------------------------------------------
    def _mask(self, t, key):
        """the ReLU mask of activation t as conv_igemm keywords: its bits when the forward pass wrote them, else the bf16 tensor itself"""
        b = self.rb.get(key)
        return dict(mask=t) if b is None else dict(mask_bits=b)

    def _defer(self, name):
        # the slabs of a deferred layer live until the stage's mis_wgrad_reduce_batch (same stream, a few launches later): one workspace per POSITION within the stage
        # (<= 3), shared by every engine on the device - keyed by (engine id, layer) they were 23 grow-only buffers per engine that nothing ever freed
        return dict(defer=self._red, ws_tag=f"defer:{len(self._red)}") if self.batch_reduce else {}

    def _bwd_conv(self, x, dy, name, cin, cout, dx=None, mask=None, dx1=None, cout0=None, dx_mode=OUT_PLAIN):
        """grads of y = relu(conv3x3(x) + b) given dy = dL/d(pre-activation); mask: (activation, key of its ReLU bits) of the layer below."""
        ops.wgrad(x, dy, self.G[name + ".weight"], ksize=3, Cin=cin, Cout=cout, dbias=self.G[name + ".bias"], side=self.side_reduce, **self._defer(name))
        if dx is not None:
            ops.conv_igemm(dy, self.wd_[name], dx, ksize=3, Cin=cout, Cout=cin, y0_mode=dx_mode, y1=dx1,
                           Cout0=cout0, **({} if mask is None else self._mask(*mask)))

    def backward(self, stage_cb=None):
        """Run after forward(train=True): fills self.G (reference-layout fp32 grads).
        stage_cb(module_prefixes) is called as soon as the gradients of those modules have been enqueued
        (used by ddp.GradReducer to start their all-reduce while the rest of backward still runs)."""
        def cb(names):
            if self._red:
                ops.wgrad_reduce_batch(self._red)          # the stage's weight gradients: their slabs summed by two launches
            if stage_cb is not None:
                ops.wgrad_join(self.device)      # the stage's weight gradients are final only after their side-stream reductions
                stage_cb(names)

        cb(["final_conv"])
        for j in range(3, -1, -1):
            l = 3 - j
            c = FEATS[l]
            x_in = self.m2 if j == 0 else self.u2[j - 1]
            g_in = self.g_m2 if j == 0 else self.g_u2[j - 1]
            self._bwd_conv(self.u1[j], self.g_u2[j], f"up_conv.{j}.second", c, c, dx=self.g_u1[j], mask=(self.u1[j], ("u1", j)))
            # first conv of the block reads the concat buffer: gradient of the up-sampled half goes out
            # pixel-unshuffled (N, h/2, w/2, 4c), gradient of the skip half goes to g_skip[l]
            self._bwd_conv(self.cat[l], self.g_u1[j], f"up_conv.{j}.first", 2 * c, c, dx=self.dys[j], dx1=self.g_skip[l],
                           cout0=c, dx_mode=OUT_UNSHUFFLE2)
            up = f"up_sample.{j}.up"
            ops.wgrad(x_in, self.dys[j], self.G[up + ".weight"], ksize=1, Cin=2 * c, Cout=4 * c, dw_layout=1,
                      dbias=self.G[up + ".bias"], side=self.side_reduce, **self._defer(up))
            ops.conv_igemm(self.dys[j], self.wd_[up], g_in, ksize=1, Cin=4 * c, Cout=2 * c, **self._mask(x_in, "m2" if j == 0 else ("u2", j - 1)))
            cb([f"up_conv.{j}", f"up_sample.{j}"])
        self._bwd_conv(self.m1, self.g_m2, "middle_conv.second", 1024, 1024, dx=self.g_m1, mask=(self.m1, "m1"))
        if stage_cb is not None:
            # data parallel: middle_conv is 56.6 MB of gradients, the largest bucket of the step - hand its 37.7 MB half to the reducer while middle_conv.first's
            # kernels (and the whole encoder backward) are still to run; without a reducer the stage stays one reduction batch
            cb(["middle_conv.second"])
        self._bwd_conv(self.pooled[3], self.g_m1, "middle_conv.first", 512, 1024, dx=self.g_pooled[3])
        cb(["middle_conv.first"] if stage_cb is not None else ["middle_conv"])
        for l in range(3, -1, -1):
            c = FEATS[l]
            skip = View(self.cat[l], c, c)
            ops.maxpool2_bwd(skip, self.g_pooled[l], self.g_skip[l], add=self.g_skip[l], relu_mask=True, pbits=self.pb[l])
            self._bwd_conv(self.t1[l], self.g_skip[l], f"down_conv.{l}.second", c, c, dx=self.g_t1[l], mask=(self.t1[l], ("t1", l)))
            if l > 0:
                self._bwd_conv(self.pooled[l - 1], self.g_t1[l], f"down_conv.{l}.first", FEATS[l - 1], c, dx=self.g_pooled[l - 1])
            else:
                ops.first_conv_wgrad(self._images, self.g_t1[0], self.G["down_conv.0.first.weight"], self.G["down_conv.0.first.bias"])
            cb([f"down_conv.{l}"])
        ops.wgrad_join(self.device)

    # ---- optimizer -------------------------------------------------------------------------------------
    def optimizer_step_dev(self):
        """optimizer_step with the step counter and the learning rate read from DEVICE memory (self.opt_step / self.lr_dev), so that the call can be
        captured in a hipGraph and replayed (graph.GraphedTrainStep); the host-side step_count is advanced by the caller of the replay"""
        f = self.flat
        if not hasattr(self, "opt_step"):
            self.opt_step = torch.full((1,), self.step_count, dtype=torch.int32, device=self.device)
            self.lr_dev = torch.full((1,), float(self.lr), dtype=torch.float32, device=self.device)
            self.opt_hyper = torch.zeros(8, dtype=torch.float32, device=self.device)
        ops.sumsq(f.g, self.partials)
        nd = f.n_decay
        common = dict(partials=self.partials, max_norm=self.max_norm, lr_dev=self.lr_dev, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                      step_dev=self.opt_step)
        ops.adamw_step_dev(f.p[:nd], f.g[:nd], f.m[:nd], f.v[:nd], weight_decay=self.wd, advance=True, hyper=self.opt_hyper[:4],
                           gradnorm_out=self.gradnorm, **common)
        ops.adamw_step_dev(f.p[nd:], f.g[nd:], f.m[nd:], f.v[nd:], weight_decay=0.0, advance=False, hyper=self.opt_hyper[4:], **common)
        self.repack()

    def optimizer_step(self, lr=None):
        """clip_grad_norm_(max_norm) + AdamW (decay on weights only), then refresh the packed operands."""
        lr = self.lr if lr is None else lr
        self.step_count += 1
        f = self.flat
        ops.sumsq(f.g, self.partials)
        nd = f.n_decay
        common = dict(partials=self.partials, max_norm=self.max_norm, lr=lr, beta1=self.betas[0], beta2=self.betas[1],
                      eps=self.eps, step=self.step_count)
        ops.adamw_step(f.p[:nd], f.g[:nd], f.m[:nd], f.v[:nd], weight_decay=self.wd, gradnorm_out=self.gradnorm, **common)
        ops.adamw_step(f.p[nd:], f.g[nd:], f.m[nd:], f.v[nd:], weight_decay=0.0, **common)
        self.repack()

    def train_step(self, images, labels, lr=None):
        loss, logits, am = self.forward(images, labels, train=True)
        self.backward()
        self.optimizer_step(lr)
        return loss
