"""Fused 3-D U-Net (pytorch-3dunet flavour) train-step engine on libmisamd (MI355X).

Mirrors the reference's `UNet3D` (model/unet3d/model.py:13-194) with its defaults: layer order 'gcr'
(GroupNorm -> Conv3d(k3, p1, no bias) -> ReLU, buildingblocks.py:14-159), DoubleConv channel plan
(buildingblocks.py:202-215), MaxPool3d(2) between encoders (:409-418), nearest upsampling to the encoder's size and
`cat((encoder_features, x), 1)` in the decoders (:546-548, :671-673), 1x1x1 head with bias (model.py:111), logits out,
and `BCEDiceLoss(1, 1)` (losses.py:167-178).

MI355X mapping: NDHWC activations; GroupNorm never materialises its output - per-(sample, channel) scale/shift are
computed from per-channel sums and applied by the conv / wgrad kernels while they stage their input tile; the decoder's
concat + nearest upsample are two-source addressing inside the same kernels (channels >= C_enc read the half-resolution
tensor at (d>>1, h>>1, w>>1)); the GroupNorm backward is 2 reductions + one fused elementwise pass that also applies
the ReLU mask, accumulates skip gradients and sums the 8 children of each coarse voxel for the upsampled source.
"""
import math

import os

import torch

from . import ops
from ._lib import MisError
from .engine2d import FlatParams
from .ops import View


def layer_plan(in_channels, f_maps, upsample="default"):
    enc = []
    for i, out in enumerate(f_maps):
        cin = in_channels if i == 0 else f_maps[i - 1]
        c1 = max(out // 2, cin)
        enc.append([(cin, c1), (c1, out)])
    dec = []
    rf = list(reversed(f_maps))
    for i in range(len(rf) - 1):
        # 'deconv' (buildingblocks.py:604-626): the transposed conv halves the channels first, the DoubleConv sees rf[i]
        dec.append([(rf[i] + rf[i + 1] if upsample != "deconv" else rf[i], rf[i + 1]), (rf[i + 1], rf[i + 1])])
    return enc, dec


def _ct_name(j):
    return f"decoders.{j}.upsampling.upsample.conv_transposed.weight"


def unet3d_param_specs(in_channels, out_channels, f_maps, upsample="default"):
    enc, dec = layer_plan(in_channels, f_maps, upsample)
    rf = list(reversed(f_maps))
    specs = []
    for grp, plan in (("encoders", enc), ("decoders", dec)):
        for i, convs in enumerate(plan):
            if grp == "decoders" and upsample == "deconv":     # registered before basic_module (buildingblocks.py:504-534)
                specs.append((_ct_name(i), (rf[i], rf[i + 1], 3, 3, 3)))
            for j, (ci, co) in enumerate(convs):
                pre = f"{grp}.{i}.basic_module.SingleConv{j + 1}"
                specs += [(f"{pre}.groupnorm.weight", (ci,)), (f"{pre}.groupnorm.bias", (ci,)), (f"{pre}.conv.weight", (co, ci, 3, 3, 3))]
    specs += [("final_conv.weight", (out_channels, f_maps[0], 1, 1, 1)), ("final_conv.bias", (out_channels,))]
    return specs


def default_init3d_(params, seed=None):
    """Same RNG stream as `torch.manual_seed(seed); UNet3D(...)` (GroupNorm init draws nothing)."""
    if seed is not None:
        torch.manual_seed(seed)
    for name, p in params.items():
        if name.endswith("groupnorm.weight"):
            p.fill_(1.0)
        elif name.endswith("groupnorm.bias"):
            p.zero_()
        elif name.endswith("conv.weight") or name.endswith("conv_transposed.weight") or name == "final_conv.weight":
            w = torch.empty(p.shape)
            torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            p.copy_(w)
        elif name == "final_conv.bias":
            fan_in = params["final_conv.weight"].shape[1]
            b = torch.empty(p.shape)
            bound = 1.0 / math.sqrt(fan_in)
            torch.nn.init.uniform_(b, -bound, bound)
            p.copy_(b)


class _SC:
    """book-keeping of one SingleConv ('gcr')"""
    pass


class UNet3DEngine:
    exact_dice, dice_group = False, None          # class defaults (the residual engines do not take the options)
    narrow32 = epi_stats = False
    _ystats = {}                                  # (never written while epi_stats is off)

    def __init__(self, in_channels=1, out_channels=3, f_maps=(64, 128, 256, 512), num_groups=8, dtype=torch.float32, device="cuda",
                 seed=None, lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3, max_grad_norm=1.0, alpha=1.0, beta=1.0,
                 upsample="default", exact_dice=None, dice_group=None):
        """exact_dice: under data parallelism (an initialised torch.distributed group) the Dice term is that of the GLOBAL batch - the 3*C per-class sums
        (I_c, P_c, T_c: 36 bytes for C = 3) are all-reduced between the forward and the gradient part of the head, as the reference computes the loss once on the
        gathered batch (model/unet3d/trainer.py:312-318, nn.DataParallel).  Default (None, round 4 / ADVICE r3): ON whenever a process group with more than one rank
        exists - the reference's semantics; False = the DistributedDataParallel semantics (per-rank loss, averaged gradients: a different loss and different gradients)."""
        self.exact_dice, self.dice_group = (True if exact_dice is None else bool(exact_dice)), dice_group
        if upsample not in ("default", "nearest", "deconv"):
            raise MisError(f"UNet3DEngine: upsample must be 'default'/'nearest' or 'deconv', got {upsample!r}")
        self.deconv = upsample == "deconv"
        if in_channels != 1:
            raise MisError("UNet3DEngine: in_channels must be 1 (the direct first-layer kernel)")
        if not (1 <= out_channels <= 4):
            raise MisError("UNet3DEngine: out_channels must be 1..4")
        f_maps = list(f_maps)
        if f_maps[0] != 64 or any(f % 64 for f in f_maps):
            raise MisError("UNet3DEngine: f_maps must start at 64 and be multiples of 64")
        ops.load()
        with torch.cuda.device(torch.device(device)):
            ops.tile_queue_init()          # the tile queue's counter pool exists before the first launch (and before any graph capture)
        self.cin, self.cout, self.f_maps, self.G = in_channels, out_channels, f_maps, num_groups
        self.dtype, self.device = dtype, torch.device(device)
        # split-K slab reductions of the weight-gradient kernels run on a second stream under the next MFMA kernel (joined per DDP stage / at the end)
        self.side_reduce = os.environ.get("MISAMD_NO_SIDE_REDUCE") is None
        # bf16: the GroupNorm output is WRITTEN once per SingleConv (mis_gn_apply: same arithmetic and rounding as the operand-staging fold) and feeds both the forward
        # convolution and the weight gradient as a plain single-source tensor - which is what lets them run on the all-DMA ping-pong kernels (conv3d_pp.hip,
        # wgrad_pp.hip); fp32 keeps the fold (its lock-step kernels sit at 0.79 of the f32 MFMA peak).  MISAMD_GN_FOLD=1: the fold in bf16 too (A/B switch).
        # (the materialised route lives on the ping-pong kernels - per-sample split-K ranges, 32-channel K chunks: with one of their A/B switches set in the environment the
        #  engine takes the operand-fold route of the lock-step kernels instead of failing in the middle of a step)
        pp_off = dtype == torch.bfloat16 and any(os.environ.get(k) for k in ("MIS_WGRAD_NOPP", "MIS_WGRAD3D_NOPP", "MIS_CONV3D_NOPP", "MIS_WGRAD_NO_TR"))
        # round 5: fp32 materialises too - its convolutions and weight gradients now run on all-DMA kernels of their own (conv3d_f32.hip, wgrad_f32.hip), which need a
        # plain single-source operand exactly like the bf16 ones; the written tensor equals what the operand fold fed the MFMAs bit for bit (fmaf, no rounding step in fp32)
        self.materialize = os.environ.get("MISAMD_GN_FOLD") is None and not pp_off
        # ... and with xn at hand the GroupNorm backward statistics (sum dyn, sum dyn * x per sample and channel) follow from the per-sample weight gradients and the
        # border sums of g_y (mis_gn_bwd_stats_from_dw) instead of a pass over dyn and x (2 x 1-3 GB per full-resolution layer).  MISAMD_GN_STATS_KERNEL=1: that pass.
        # (bf16 only: the fp32 parity mode keeps the direct statistics pass - exact at any gamma / beta, 2.6 ms of its 180 ms step)
        self.gn_from_dw = self.materialize and dtype == torch.bfloat16 and os.environ.get("MISAMD_GN_STATS_KERNEL") is None
        # ... and with the statistics known BEFORE the dgrad runs, the single-source layers (10 of 13) continue their dgrad through the GroupNorm and the ReLU in its
        # epilogue (MisConvDesc.gn_p, round 4): dx = mask * (p * acc + q * x + r) straight from the fp32 accumulator - dL/d(normalised operand) is never written and
        # mis_gn_bwd_apply's pass over three tensors disappears.  MISAMD_GN_BWD_UNFUSED=1: the separate pass (A/B switch; the two-source decoder layers always take it).
        self.fuse_gn_bwd = self.gn_from_dw and os.environ.get("MISAMD_GN_BWD_UNFUSED") is None
        self.levels = len(f_maps)
        self.specs = unet3d_param_specs(in_channels, out_channels, f_maps, upsample)
        self.flat = FlatParams(self.specs, self.device, lambda n: not n.endswith("bias"))
        self.P, self.Gr = self.flat.param, self.flat.grad
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.alpha, self.beta = alpha, beta
        self.step_count = 0
        host = {n: torch.empty(s) for n, s in self.specs}
        default_init3d_(host, seed)
        for n in host:
            self.P[n].copy_(host[n])
        self.ck = 64 if dtype == torch.bfloat16 else 32
        self.c1 = max(f_maps[0] // 2, in_channels)                 # 32
        # the first SingleConv's 32 output channels live in a 64-channel buffer when the consumer folds GroupNorm into a 64-channel K chunk (fp32 / MISAMD_GN_FOLD);
        # with the materialised operand (bf16) only the 64-wide buffers xn / dyn carry the padding: every pass over t_enc[0] / g_t_enc[0] moves half the bytes
        self.c1p = self.c1 if (self.materialize and self.c1 % 32 == 0) else (self.c1 + 63) // 64 * 64
        # SingleConv descriptors
        enc, dec = layer_plan(in_channels, f_maps, upsample)
        self.ct = []                                               # transposed-conv upsamplers ('deconv')
        if self.deconv:
            rf = list(reversed(f_maps))
            for j in range(len(rf) - 1):
                t = _SC()
                t.name, t.cin, t.cout = _ct_name(j), rf[j], rf[j + 1]
                t.w2d = torch.empty(27 * t.cout, t.cin, 1, device=self.device)             # row k*Cout + co, fp32
                t.wf = torch.empty(1, 27 * t.cout, t.cin, dtype=dtype, device=self.device)  # GEMM operand of the forward
                t.wd = torch.empty(1, t.cin, 27 * t.cout, dtype=dtype, device=self.device)  # ... of the input gradient
                t.dw2d = torch.empty(27 * t.cout, t.cin, device=self.device)
                self.ct.append(t)
        self.sc = {}
        # round 6: the fp32 all-DMA kernels have 32-column (dgrad) and 32-input-channel (weight gradient) tiles, so encoders.0 SingleConv2 (32 -> 64,
        # buildingblocks.py:202-211) keeps its 32 real channels everywhere - operand, dgrad output and weight gradient are not padded to 64 (half of that layer's backward
        # was multiplications by zero: 928 GFLOP of cfg4's step).  With the lock-step kernels selected (MIS_CONV3D_F32_NOPP / MIS_WGRAD_F32_NOPP) the padded plan stays.
        self.narrow32 = (dtype == torch.float32 and self.materialize and not ops.dispatch_switch("MIS_CONV3D_F32_NOPP")
                         and not ops.dispatch_switch("MIS_WGRAD_F32_NOPP") and os.environ.get("MISAMD_F32_PAD64") is None)
        # ... and a statistics epilogue (MisConvDesc.st_mode): the dgrad leaves the two reductions of the GroupNorm backward (sum dyn, sum dyn * x: mis_gn_bwd_stats' pass over
        # both tensors, 2.6 ms of cfg4's step) and the forward convolution the sums the NEXT GroupNorm needs (mis_chanstats' pass, 0.9 ms) as per-tile partial rows, reduced in
        # a fixed order.  MISAMD_NO_EPI_STATS=1: the separate passes (A/B switch).
        self.epi_stats = self.narrow32 and os.environ.get("MISAMD_NO_EPI_STATS") is None
        self._ystats, self._epi_ok = {}, {}
        for grp, plan in (("encoders", enc), ("decoders", dec)):
            for i, convs in enumerate(plan):
                for j, (ci, co) in enumerate(convs):
                    s = _SC()
                    s.name = f"{grp}.{i}.basic_module.SingleConv{j + 1}"
                    s.cin, s.cout = ci, co
                    s.cin_pad = ci if (self.narrow32 and ci % 32 == 0) else ((ci + 63) // 64 * 64 if ci > 1 else 1)
                    s.groups = 1 if ci < num_groups else num_groups
                    s.first = (grp == "encoders" and i == 0 and j == 0)
                    if not s.first:
                        s.wf = torch.empty(27, co, s.cin_pad, dtype=dtype, device=self.device)
                        s.wd = torch.empty(27, s.cin_pad, co, dtype=dtype, device=self.device)
                        s.wpad = torch.zeros(co, s.cin_pad, 3, 3, 3, device=self.device) if s.cin_pad != ci else None
                        # forward operand over the REAL input channels where the ping-pong kernel's 32-channel K chunks allow it (encoders.0 SingleConv2: 32 of 64):
                        # half the MFMA work of that launch; the dgrad / weight-gradient operands stay padded (64-column / 64-channel tiles)
                        s.wf_real = torch.empty(27, co, ci, dtype=dtype, device=self.device) if (self.materialize and s.cin_pad != ci and ci % 32 == 0) else None
                        s.dwpad = torch.zeros(co, s.cin_pad, 3, 3, 3, device=self.device) if s.cin_pad != ci else None
                    self.sc[s.name] = s
        self.partials = torch.zeros(ops.sumsq_npartials(self.flat.total), dtype=torch.float32, device=self.device)
        self.gradnorm = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.loss_buf = torch.zeros(32, dtype=torch.float32, device=self.device)
        self._shape = None
        # conditioning guard of the statistics-from-dW route (csrc/groupnorm.hip: mis_gn_cond): a layer with a channel whose |gamma| < GN_COND_RATIO * |beta| takes
        # the direct pass over dyn and x (and the unfused GroupNorm backward) instead.  The flags are computed on the device after every optimizer step and read back
        # asynchronously (pinned buffer + event): a backward pass uses the newest flags that have ARRIVED - at most a step or two old, which the margin of the ratio covers
        # (the direct route is exact at any ratio; the route through dW is within ~2^-9 / GN_COND_RATIO of it at the threshold).
        self._gn_layers = [s for s in self.sc.values() if not s.first]
        for s in self._gn_layers:
            s.gn_direct = False
        if self.gn_from_dw:
            self._gn_tab = ops.GnCondTable(self.flat.p, [(self.P[s.name + ".groupnorm.weight"], self.P[s.name + ".groupnorm.bias"]) for s in self._gn_layers])
            self._gn_flags = torch.zeros(len(self._gn_layers), dtype=torch.int32, device=self.device)
        self._gn_ring, self._gn_free = [], []
        self.repack()
        self.refresh_gn_flags(sync=True)

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
        self.refresh_gn_flags(sync=True)

    GN_COND_RATIO = 2.0 ** -4

    def refresh_gn_flags(self, sync=False):
        """recompute which layers need the direct GroupNorm-backward statistics (see __init__).  Called by repack(), i.e. at EVERY parameter-change site (optimizer steps incl.
        the device-side one of the graphed step, load_state_dict, the nn.Module's parameter sync, graph restore).  sync=True: wait for the result (construction,
        load_state_dict, the module's first parameter copy).  The read-back is asynchronous otherwise: each call queues one (event, pinned buffer) pair in a small ring and
        _poll_gn_flags applies the NEWEST pair whose event has completed without dropping the pending ones - a free-running loop, where the host stays a step or more ahead
        of the device, therefore sees flags at most GN_RING - 1 steps old; when the ring is full the oldest pending pair is waited for (a bounded wait on work that is
        GN_RING steps behind the host).  Under graph capture nothing is queued: the captured step keeps the routes it was captured with, and GraphedTrainStep re-checks the
        flags between replays (graph.py)."""
        if not self.gn_from_dw:
            return
        if torch.cuda.is_current_stream_capturing():
            return
        ops.gn_cond(self._gn_tab, self.GN_COND_RATIO, self._gn_flags)
        ring = self._gn_ring
        if len(ring) >= self.GN_RING:                        # bounded wait: the pair queued GN_RING refreshes ago
            ev, buf = ring.pop(0)
            ev.synchronize()
            self._apply_gn_flags(buf)
            self._gn_free.append(buf)
        buf = self._gn_free.pop() if self._gn_free else torch.zeros(len(self._gn_layers), dtype=torch.int32).pin_memory()
        buf.copy_(self._gn_flags, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        ring.append((ev, buf))
        if sync:
            ev.synchronize()
            self._poll_gn_flags()

    GN_RING = 4

    def _apply_gn_flags(self, buf):
        for s, f in zip(self._gn_layers, buf.tolist()):
            s.gn_direct = bool(f)

    def _poll_gn_flags(self):
        ring = getattr(self, "_gn_ring", None)
        if not ring or torch.cuda.is_current_stream_capturing():
            return
        done = -1
        for i, (ev, _) in enumerate(ring):                   # events complete in queue order
            if ev.query():
                done = i
            else:
                break
        if done >= 0:
            self._apply_gn_flags(ring[done][1])
            for ev, buf in ring[:done + 1]:
                self._gn_free.append(buf)
            del ring[:done + 1]

    def gn_routes(self):
        """the per-layer GroupNorm-backward routes in force (True = direct statistics pass): what a captured graph was recorded with"""
        return tuple(bool(s.gn_direct) for s in self._gn_layers)

    def repack(self):
        """fp32 master weights -> the packed MFMA operands of every SingleConv in ONE launch (mis_pack_batch: the flat parameter buffer and the operand buffers never
        move, so the table of their addresses is built once; round 2 launched one pack kernel per layer: 17 launches, 0.5 ms per 160^3 step)"""
        if getattr(self, "_pack_table", None) is None:
            entries = []
            for s in self.sc.values():
                if s.first:
                    continue
                entries.append((self.P[s.name + ".conv.weight"] if s.wpad is None else s.wpad, s.wf, s.wd, 0))
                if getattr(s, "wf_real", None) is not None:
                    entries.append((self.P[s.name + ".conv.weight"], s.wf_real, None, 0))
            self._pack_table = ops.PackTable(entries, self.device) if entries else False
        for s in self.sc.values():
            if not s.first and s.wpad is not None:
                s.wpad[:, :s.cin] = self.P[s.name + ".conv.weight"]
        if self._pack_table:
            ops.pack_batch(self._pack_table)
        for t in self.ct:       # W [Cin][Cout][27] -> [27*Cout][Cin] (row k*Cout + co), then the two packed GEMM operands
            t.w2d.view(27, t.cout, t.cin).copy_(self.P[t.name].view(t.cin, t.cout, 27).permute(2, 1, 0))
            ops.pack_conv_weight(t.w2d, t.wf, t.wd)
        # the parameters have (possibly) changed: so may the conditioning of the GroupNorm layers (every parameter-change site goes through here)
        if getattr(self, "_gn_ring", None) is not None:
            self.refresh_gn_flags()

    # ---- buffers ---------------------------------------------------------------------------------------
    def _alloc(self, N, D, H, W):
        if self._shape == (N, D, H, W):
            return
        if self.epi_stats:
            self._ystats = {}                     # keyed by buffer address: the buffers are about to be replaced
        div = 1 << (self.levels - 1)
        if D % div or H % div or W % div:
            raise MisError(f"the fused 3-D engine needs D, H, W divisible by {div}; got {D}x{H}x{W}")
        dt, dev = self.dtype, self.device

        def buf(l, c):
            return torch.empty(N, D >> l, H >> l, W >> l, c, dtype=dt, device=dev)

        L = self.levels
        fm = self.f_maps
        self.t_enc, self.e, self.pooled = [], [], []
        self.g_t_enc, self.g_e, self.g_pooled = [], [], []
        for l in range(L):
            c_mid = self.sc[f"encoders.{l}.basic_module.SingleConv1"].cout
            c_mid_p = self.c1p if l == 0 else c_mid
            self.t_enc.append(buf(l, c_mid_p))
            self.e.append(buf(l, fm[l]))
            self.g_t_enc.append(buf(l, c_mid_p))
            self.g_e.append(buf(l, fm[l]))
            if l < L - 1:
                self.pooled.append(buf(l + 1, fm[l]))
                self.g_pooled.append(buf(l + 1, fm[l]))
        self.t_dec, self.d, self.g_t_dec, self.g_d = [], [], [], []
        self.up, self.g_up = [], []
        for j in range(L - 1):
            l = L - 2 - j
            if self.deconv:
                self.up.append(buf(l, fm[l]))
                self.g_up.append(buf(l, fm[l]))
            self.t_dec.append(buf(l, fm[l]))
            self.d.append(buf(l, fm[l]))
            self.g_t_dec.append(buf(l, fm[l]))
            self.g_d.append(buf(l, fm[l]))
        # per-SingleConv GroupNorm state and the dgrad buffer
        for s in self.sc.values():
            l = self._level(s.name)
            cp = 4 if s.first else s.cin_pad
            s.scale = torch.zeros(N, cp, device=dev)
            s.shift = torch.zeros(N, cp, device=dev)
            s.mean = torch.zeros(N, s.groups, device=dev)
            s.rstd = torch.zeros(N, s.groups, device=dev)
            cs = 4 if s.first else s.cin
            s.S1, s.S2 = torch.zeros(N, cs, device=dev), torch.zeros(N, cs, device=dev)
            s.p, s.q, s.r = torch.zeros(N, cs, device=dev), torch.zeros(N, cs, device=dev), torch.zeros(N, cs, device=dev)
            s.sum0, s.sq0 = torch.zeros(N, cs, device=dev), torch.zeros(N, cs, device=dev)
            s.sum1, s.sq1 = torch.zeros(N, cs, device=dev), torch.zeros(N, cs, device=dev)
            s.dgam, s.dbet = torch.zeros(cs, device=dev), torch.zeros(cs, device=dev)
            s.xn = buf(l, s.cin_pad) if (self.materialize and not s.first) else None
            if s.xn is not None and getattr(self, "gn_from_dw", False):
                s.dwn = torch.empty(N, s.cout, s.cin_pad, 3, 3, 3, device=dev)
                s.gysum = torch.empty(N, s.cout, device=dev)
            if s.xn is not None and s.cin_pad != s.cin:
                s.xn.zero_()                                       # the padding channels are never written: they must read as 0 in the weight gradient
        self.dyn = {}      # dgrad outputs, keyed by (level, channels): shared between SingleConvs of equal shape
        for s in self.sc.values():
            if s.first:
                continue
            key = (self._level(s.name), s.cin_pad)
            if key not in self.dyn:
                self.dyn[key] = buf(key[0], key[1])
        if self.deconv:     # one column buffer (N, d, h, w, 27*Cout) shared by all levels and by forward / backward
            self.cols_elems = max(N * (D >> (l + 1)) * (H >> (l + 1)) * (W >> (l + 1)) * 27 * fm[l] for l in range(L - 1))
            self.cols = torch.empty(self.cols_elems, dtype=dt, device=dev)
        self.logits = torch.empty(N, self.cout, D, H, W, dtype=torch.float32, device=dev)
        self.argmax = torch.empty(N, D, H, W, dtype=torch.uint8, device=dev)
        self._shape = (N, D, H, W)

    def _level(self, name):
        grp, i = name.split(".")[0], int(name.split(".")[1])
        return i if grp == "encoders" else self.levels - 2 - i

    # ---- forward ---------------------------------------------------------------------------------------
    def _epi(self, x, y, cin, cout, grid):
        """does the convolution x -> y of this shape run on a kernel with the statistics epilogue?  (asked once per shape)"""
        if not self.epi_stats:
            return False
        key = (x.shape[-1], y.shape[-1], cin, cout, grid)
        ok = self._epi_ok.get(key)
        if ok is None:
            ok = self._epi_ok[key] = bool(ops.conv_stats_supported(x, y, Cin=cin, Cout=cout, grid=grid))
        return ok

    def _src_stats(self, src, c, own_sum, own_sq):
        """per-(sample, channel) sum / sum of squares of a GroupNorm input: left behind by the convolution that produced it (statistics epilogue), or one pass over it"""
        st = self._ystats.get(src.data_ptr())
        if st is not None and st[2] == (tuple(src.shape), c):
            return st[0], st[1]
        ops.chanstats(View(src, 0, c), own_sum, own_sq)
        return own_sum, own_sq

    def _gn_fwd(self, s, src0, c0, src1=None, c1=0):
        N = src0.shape[0]
        count = src0.shape[1] * src0.shape[2] * src0.shape[3]
        sum0, sq0 = self._src_stats(src0, c0, s.sum0, s.sq0)
        sum1 = sq1 = None
        if src1 is not None:
            sum1, sq1 = self._src_stats(src1, c1, s.sum1, s.sq1)
        mult1 = 8.0 if (src1 is not None and src1.shape[1] != src0.shape[1]) else 1.0     # nearest-upsampled source: 8 children per voxel
        ops.gn_fwd_finalize(sum0, sq0, c0, 1.0, sum1, sq1, c1, mult1,
                            N, s.groups, count, self.P[s.name + ".groupnorm.weight"], self.P[s.name + ".groupnorm.bias"], s.cin_pad,
                            s.scale, s.shift, s.mean, s.rstd)

    def _sc_fwd(self, s, src0, c0, y, src1=None, c1=0):
        self._gn_fwd(s, src0, c0, src1, c1)
        s.src0, s.c0, s.src1, s.c1 = src0, c0, src1, c1
        grid = (src0.shape[0], src0.shape[1], src0.shape[2], src0.shape[3])
        if s.xn is not None:
            if src1 is None:
                ops.gn_apply(src0, src0.shape[-1], False, grid, s.scale, s.shift, s.cin_pad, 0, s.xn)
            else:
                ops.gn_apply(View(src0, 0, c0), c0, False, grid, s.scale, s.shift, s.cin_pad, 0, s.xn)
                ops.gn_apply(View(src1, 0, c1), c1, src1.shape[1] != src0.shape[1], grid, s.scale, s.shift, s.cin_pad, c0, s.xn)
            if getattr(s, "wf_real", None) is not None:
                ops.conv_igemm(View(s.xn, 0, s.cin), s.wf_real, y, ksize=3, Cin=s.cin, Cout=s.cout, grid=grid, relu=getattr(s, "relu", True))
            else:
                st = None
                if y.shape[-1] == s.cout and self._epi(s.xn, y, s.cin_pad, s.cout, grid):      # the output's statistics for the GroupNorm that reads it next
                    ent = self._ystats.get(y.data_ptr())
                    if ent is None or ent[2] != (tuple(y.shape), s.cout):
                        ent = self._ystats[y.data_ptr()] = (torch.zeros(grid[0], s.cout, device=self.device), torch.zeros(grid[0], s.cout, device=self.device),
                                                            (tuple(y.shape), s.cout))
                    st = dict(mode=2, S1=ent[0], S2=ent[1])
                ops.conv_igemm(s.xn, s.wf, y, ksize=3, Cin=s.cin_pad, Cout=s.cout, grid=grid, relu=getattr(s, "relu", True), stats=st, real=(s.cin, s.cout))
            return
        ops.conv_igemm(View(src0, 0, src0.shape[-1] if src1 is None else c0), s.wf, y, ksize=3, Cin=s.cin_pad, Cout=s.cout, grid=grid,
                       x1=None if src1 is None else View(src1, 0, c1), relu=getattr(s, "relu", True), in_scale=s.scale, in_shift=s.shift)

    def _cols_view(self, low, cout):
        n, d, h, w = low.shape[:4]
        return self.cols[:n * d * h * w * 27 * cout].view(n, d, h, w, 27 * cout)

    def _ct_fwd(self, j, low):
        """ConvTranspose3d(k3, s2, p1) of `low` as a 27*Cout-column GEMM + gather, resized to the encoder grid -> self.up[j]"""
        t = self.ct[j]
        t.src = low
        cols = self._cols_view(low, t.cout)
        ops.conv_igemm(low, t.wf, cols, ksize=1, Cin=t.cin, Cout=27 * t.cout)
        ops.convt3_col2im(cols, self.up[j])
        return self.up[j]

    def _ct_bwd(self, j, low_grad):
        t = self.ct[j]
        low = t.src
        gcols = self._cols_view(low, t.cout)
        ops.convt3_im2col(self.g_up[j], gcols)
        ops.wgrad(low, gcols, t.dw2d, ksize=1, Cin=t.cin, Cout=27 * t.cout)
        self.Gr[t.name].view(t.cin, t.cout, 27).copy_(t.dw2d.view(27, t.cout, t.cin).permute(2, 1, 0))
        ops.conv_igemm(gcols, t.wd, low_grad, ksize=1, Cin=27 * t.cout, Cout=t.cin, mask=low)      # low is a ReLU output

    def forward(self, x, target=None, train=True, grad_scale=1.0):
        """x: fp32 (N,1,D,H,W) on the device; target: fp32 (N,C,D,H,W) in {0,1}. Returns (loss[1] or None, logits, argmax)."""
        if x.dtype != torch.float32 or not x.is_contiguous() or x.device.type != "cuda" or x.dim() != 5 or x.shape[1] != 1:
            raise MisError("x must be a contiguous fp32 CUDA tensor (N, 1, D, H, W)")
        N, _, D, H, W = x.shape
        self._alloc(N, D, H, W)
        self._x = x
        ops.tile_queue_reset()          # a step never inherits tile-queue counters from an earlier launch (captured as a memset node)
        P, L = self.P, self.levels
        npix = D * H * W
        # first SingleConv: GroupNorm(1 group over the single channel) + direct conv
        s = self.sc["encoders.0.basic_module.SingleConv1"]
        xv = x.view(N, 1, 1, npix // 4, 4)
        ops.chanstats(xv, s.sum0, s.sq0)
        g4 = P[s.name + ".groupnorm.weight"].repeat(4)
        b4 = P[s.name + ".groupnorm.bias"].repeat(4)
        s.g4 = g4
        ops.gn_fwd_finalize(s.sum0, s.sq0, 4, 1.0, None, None, 0, 1.0, N, 1, npix // 4, g4, b4, 4, s.scale, s.shift, s.mean, s.rstd)
        ops.first3d_fwd(x, s.scale, s.shift, 4, P[s.name + ".conv.weight"], self.c1, self.t_enc[0], self.c1p)
        self._sc_fwd(self.sc["encoders.0.basic_module.SingleConv2"], self.t_enc[0], self.c1, self.e[0])
        for l in range(1, L):
            ops.maxpool2_fwd(self.e[l - 1], self.pooled[l - 1])
            s1 = self.sc[f"encoders.{l}.basic_module.SingleConv1"]
            self._sc_fwd(s1, self.pooled[l - 1], s1.cin, self.t_enc[l])
            s2 = self.sc[f"encoders.{l}.basic_module.SingleConv2"]
            self._sc_fwd(s2, self.t_enc[l], s2.cin, self.e[l])
        low = self.e[L - 1]
        for j in range(L - 1):
            l = L - 2 - j
            s1 = self.sc[f"decoders.{j}.basic_module.SingleConv1"]
            if self.deconv:
                low = self._ct_fwd(j, low)
            self._sc_fwd(s1, self.e[l], self.f_maps[l], self.t_dec[j], src1=low, c1=low.shape[-1])
            s2 = self.sc[f"decoders.{j}.basic_module.SingleConv2"]
            self._sc_fwd(s2, self.t_dec[j], s2.cin, self.d[j])
            low = self.d[j]
        wh = P["final_conv.weight"].view(self.cout, self.f_maps[0])
        bh = P["final_conv.bias"]
        feat = self.d[L - 2]
        if target is None:
            ops.head_loss(feat, wh, bh, loss=ops.LOSS_NONE, logits=self.logits, argmax=self.argmax)
            return None, self.logits, self.argmax
        if target.dtype != torch.float32 or tuple(target.shape) != (N, self.cout, D, H, W) or not target.is_contiguous():
            raise MisError("target must be contiguous fp32 (N, C, D, H, W)")
        kw = dict(loss=ops.LOSS_BCEDICE, labels=target, logits=self.logits, argmax=self.argmax, loss_out=self.loss_buf,
                  alpha=self.alpha, beta=self.beta)
        if train:
            kw.update(dy=self.g_d[L - 2], dw=self.Gr["final_conv.weight"], db=self.Gr["final_conv.bias"], grad_scale=grad_scale)
        import torch.distributed as dist
        if train and self.exact_dice and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.dice_group) > 1:
            # global-batch Dice: forward part, SUM the per-class Dice sums (and the BCE means) over the ranks, redo the 40-flop loss formula on the summed values,
            # then the gradient part from them.  grad_scale carries 1 / world for the SUM all-reduce of the gradients: the BCE term is a mean of rank means and takes
            # it; the Dice gradient from global sums already is the global derivative, hence beta x world.
            world = dist.get_world_size(self.dice_group)
            ops.head_loss(feat, wh, bh, phase=1, **kw)
            C = self.cout
            sums = self.loss_buf[1:2 + 3 * C]                                # [bce mean, I_c, P_c, T_c]
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.dice_group)
            sums[0:1].div_(world)
            inter, den = sums[1:1 + C], (sums[1 + C:1 + 2 * C] + sums[1 + 2 * C:1 + 3 * C]).clamp_(min=1e-6)
            self.loss_buf[0:1] = self.alpha * sums[0:1] + self.beta * (1.0 - (2.0 * inter / den).mean())
            kw["beta"] = self.beta * world
            ops.head_loss(feat, wh, bh, phase=2, **kw)
        else:
            ops.head_loss(feat, wh, bh, **kw)
        return self.loss_buf[:1], self.logits, self.argmax

    def head_backward(self, dlogits):
        """backward entry for an EXTERNAL loss: dlogits = dL/dlogits, fp32 (N, C, D, H, W); then call backward()."""
        L = self.levels
        wh = self.P["final_conv.weight"].view(self.cout, self.f_maps[0])
        ops.head_loss(self.d[L - 2], wh, self.P["final_conv.bias"], loss=ops.LOSS_EXTERNAL, labels=dlogits.contiguous(),
                      dy=self.g_d[L - 2], dw=self.Gr["final_conv.weight"], db=self.Gr["final_conv.bias"])

    # ---- backward --------------------------------------------------------------------------------------
    def _sc_bwd(self, s, g_y, dx0, mask0, add0=None, dx1=None, up1=True):
        """g_y = dL/d(pre-activation) of s's output.  Produces weight/GroupNorm grads and the input gradients."""
        src0, c0, src1, c1 = s.src0, s.c0, s.src1, s.c1
        N, D, H, W = src0.shape[0], src0.shape[1], src0.shape[2], src0.shape[3]
        grid = (N, D, H, W)
        x0v = View(src0, 0, src0.shape[-1] if src1 is None else c0)
        x1v = None if src1 is None else View(src1, 0, c1)
        dw = self.Gr[s.name + ".conv.weight"] if s.dwpad is None else s.dwpad
        from_dw = s.xn is not None and getattr(s, "dwn", None) is not None and not s.gn_direct
        if s.xn is not None:
            # (per-sample gradients feed this layer's GroupNorm backward right after the dgrad: their reductions stay on the main stream - ~75 MB of slabs per layer -
            #  a side-stream reduction is starved by the persistent dgrad kernel and would be waited for)
            ops.wgrad(s.xn, g_y, dw, ksize=3, Cin=s.cin_pad, Cout=s.cout, grid=grid, side=self.side_reduce and s.dwpad is None and not from_dw,
                      dw_per_sample=s.dwn if from_dw else None, dbias_per_sample=s.gysum if from_dw else None, real=(s.cin, s.cout))
        else:
            ops.wgrad(x0v, g_y, dw, ksize=3, Cin=s.cin_pad, Cout=s.cout, grid=grid, x1=x1v, in_scale=s.scale, in_shift=s.shift,
                      side=self.side_reduce and s.dwpad is None)
        if s.dwpad is not None:
            self.Gr[s.name + ".conv.weight"].copy_(s.dwpad[:, :s.cin])
        ctot = c0 + c1
        if from_dw and self.fuse_gn_bwd and src1 is None and add0 is None:
            ops.gn_bwd_stats_from_dw(g_y, self.P[s.name + ".conv.weight"] if s.wpad is None else s.wpad, s.dwn, s.gysum, s.scale, s.shift, s.mean, s.groups, ctot,
                                     s.S1, s.S2)
            ops.gn_bwd_finalize(s.S1, s.S2, s.mean, s.rstd, self.P[s.name + ".groupnorm.weight"], N, ctot, s.groups, D * H * W,
                                s.p, s.q, s.r, self.Gr[s.name + ".groupnorm.weight"], self.Gr[s.name + ".groupnorm.bias"])
            ops.conv_igemm(g_y, s.wd, View(dx0, 0, c0), ksize=3, Cin=s.cout, Cout=s.cin_pad, Cout0=c0, grid=grid, mask=View(src0, 0, c0),
                           gn_bwd=(s.p, s.q, s.r, mask0), real=(s.cout, s.cin))
            return
        dyn = self.dyn[(self._level(s.name), s.cin_pad)]
        epi = not from_dw and s.cin_pad == ctot and self._epi(g_y, dyn, s.cout, s.cin_pad, grid)
        ops.conv_igemm(g_y, s.wd, dyn, ksize=3, Cin=s.cout, Cout=s.cin_pad, grid=grid, real=(s.cout, s.cin),
                       stats=dict(mode=1, x0=View(src0, 0, c0), x1=x1v, up=up1, S1=s.S1, S2=s.S2) if epi else None)
        if epi:
            pass
        elif from_dw:
            ops.gn_bwd_stats_from_dw(g_y, self.P[s.name + ".conv.weight"] if s.wpad is None else s.wpad, s.dwn, s.gysum, s.scale, s.shift, s.mean, s.groups, ctot,
                                     s.S1, s.S2)
        else:
            ops.gn_bwd_stats(dyn, View(src0, 0, c0), c0, False, grid, s.S1, s.S2, ctot, 0)
            if src1 is not None:
                ops.gn_bwd_stats(dyn, View(src1, 0, c1), c1, up1, grid, s.S1, s.S2, ctot, c0)
        ops.gn_bwd_finalize(s.S1, s.S2, s.mean, s.rstd, self.P[s.name + ".groupnorm.weight"], N, ctot, s.groups, D * H * W,
                            s.p, s.q, s.r, self.Gr[s.name + ".groupnorm.weight"], self.Gr[s.name + ".groupnorm.bias"])
        ops.gn_bwd_apply(dyn, View(src0, 0, c0), c0, False, grid, s.p, s.q, s.r, ctot, 0, View(dx0, 0, c0), relu_mask=mask0, add=add0)
        if src1 is not None:
            # nearest source = a ReLU output on the half grid (mask here); 'deconv' source = the linear transposed-conv output
            ops.gn_bwd_apply(dyn, View(src1, 0, c1), c1, up1, grid, s.p, s.q, s.r, ctot, c0, dx1, relu_mask=up1)

    def _stage_cb(self, stage_cb):
        if stage_cb is None:
            return lambda names: None

        def cb(names):
            ops.wgrad_join(self.device)      # a stage's weight gradients are final only after their side-stream reductions
            stage_cb(names)
        return cb

    def backward(self, stage_cb=None):
        cb = self._stage_cb(stage_cb)
        L = self.levels
        self._poll_gn_flags()
        cb(["final_conv"])
        for j in range(L - 2, -1, -1):
            l = L - 2 - j
            s2 = self.sc[f"decoders.{j}.basic_module.SingleConv2"]
            self._sc_bwd(s2, self.g_d[j], self.g_t_dec[j], mask0=True)
            s1 = self.sc[f"decoders.{j}.basic_module.SingleConv1"]
            low_grad = self.g_e[L - 1] if j == 0 else self.g_d[j - 1]
            # encoder features also feed the pooling path: leave their gradient raw (masked + accumulated in pool-bwd),
            # except the deepest-but-one level... every e[l], l < L-1, is pooled, so never mask here
            if self.deconv:
                self._sc_bwd(s1, self.g_t_dec[j], self.g_e[l], mask0=False, dx1=self.g_up[j], up1=False)
                self._ct_bwd(j, low_grad)
            else:
                self._sc_bwd(s1, self.g_t_dec[j], self.g_e[l], mask0=False, dx1=low_grad)
            cb([f"decoders.{j}"])
        for l in range(L - 1, 0, -1):
            s2 = self.sc[f"encoders.{l}.basic_module.SingleConv2"]
            self._sc_bwd(s2, self.g_e[l], self.g_t_enc[l], mask0=True)
            s1 = self.sc[f"encoders.{l}.basic_module.SingleConv1"]
            self._sc_bwd(s1, self.g_t_enc[l], self.g_pooled[l - 1], mask0=False)
            # g_e[l-1] <- relu_mask(e[l-1]) * (scatter(g_pooled) + g_e[l-1] (from the decoder))
            ops.maxpool2_bwd(self.e[l - 1], self.g_pooled[l - 1], self.g_e[l - 1], add=self.g_e[l - 1], relu_mask=True)
            cb([f"encoders.{l}"])
        s2 = self.sc["encoders.0.basic_module.SingleConv2"]
        self._sc_bwd(s2, self.g_e[0], self.g_t_enc[0], mask0=True)
        # first layer: weight grad + gradient w.r.t. the normalised input (for the 1-channel GroupNorm parameters)
        s = self.sc["encoders.0.basic_module.SingleConv1"]
        x = self._x
        N, _, D, H, W = x.shape
        npix = D * H * W
        # dW plus the 1-channel GroupNorm's dgamma / dbeta straight from the correlation sums (no dL/d(input) pass)
        ops.first3d_bwd(x, s.mean, s.rstd, self.P[s.name + ".groupnorm.weight"], self.P[s.name + ".groupnorm.bias"], self.g_t_enc[0], self.c1p,
                        self.P[s.name + ".conv.weight"], self.c1, self.Gr[s.name + ".conv.weight"], self.Gr[s.name + ".groupnorm.weight"],
                        self.Gr[s.name + ".groupnorm.bias"])
        cb(["encoders.0"])
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
        lr = self.lr if lr is None else lr
        self.step_count += 1
        f = self.flat
        ops.sumsq(f.g, self.partials)
        nd = f.n_decay
        common = dict(partials=self.partials, max_norm=self.max_norm, lr=lr, beta1=self.betas[0], beta2=self.betas[1],
                      eps=self.eps, step=self.step_count)
        ops.adamw_step(f.p[:nd], f.g[:nd], f.m[:nd], f.v[:nd], weight_decay=self.wd, gradnorm_out=self.gradnorm, **common)
        ops.adamw_step(f.p[nd:], f.g[nd:], f.m[nd:], f.v[nd:], weight_decay=0.0, **common)
        self.repack()                                        # (refreshes the GroupNorm conditioning flags)

    def train_step(self, x, target, lr=None):
        loss, _, _ = self.forward(x, target, train=True)
        self.backward()
        self.optimizer_step(lr)
        return loss
