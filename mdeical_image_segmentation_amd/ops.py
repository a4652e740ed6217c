"""Thin Python wrappers over the C ABI (include/misamd.h).  torch is used as the device allocator only:
every wrapper passes raw pointers + sizes and enqueues on torch's current HIP stream."""
import ctypes as C

import os as _os

import torch

from . import _lib
from ._lib import (MIS_BF16, MIS_F32, OUT_PLAIN, OUT_SHUFFLE2, OUT_UNSHUFFLE2, ConvDesc, HeadDesc, MisError, WgradDesc, WgradReduceItem, check,
                   dtype_code, load, stream_ptr)


class View:
    """A channel slice [c0, c0+C) of a channels-last tensor (N, D, H, W, Ctot) or (N, H, W, Ctot)."""

    def __init__(self, t, c0=0, C=None):
        assert t.is_contiguous()
        self.t = t
        self.ld = t.shape[-1]
        self.c0 = c0
        self.C = (t.shape[-1] - c0) if C is None else C
        assert 0 <= c0 and c0 + self.C <= self.ld
        if t.dim() == 4:
            self.N, self.H, self.W = t.shape[0], t.shape[1], t.shape[2]
            self.D = 1
        else:
            self.N, self.D, self.H, self.W = t.shape[0], t.shape[1], t.shape[2], t.shape[3]

    @property
    def ptr(self):
        return self.t.data_ptr() + self.c0 * self.t.element_size()

    @property
    def dtype(self):
        return self.t.dtype

    @property
    def npix(self):
        return self.N * self.D * self.H * self.W


def _v(x):
    return x if isinstance(x, View) else View(x)


# Optional per-launch timing (bench.py): when PROFILE is a list, every MFMA kernel launch is bracketed by HIP
# events on the launch stream and (kernel symbol key, algorithmic FLOPs, start, end) is appended.
PROFILE = None


EVENT_POOL = []          # pre-created, once-recorded timing events for the per-launch brackets (prepare_events)


def prepare_events(n):
    """create n timing events and record each once, OUTSIDE any timed region: the first record of a torch event creates its HIP event, and on this stack the first
    ~130 creations of a process cost 0.3-0.45 ms each - a bench step whose launches are bracketed with fresh events was host-bound for 35-60 ms"""
    while len(EVENT_POOL) < n:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        EVENT_POOL.append(e)
    torch.cuda.synchronize()


def _event():
    return EVENT_POOL.pop() if EVENT_POOL else torch.cuda.Event(enable_timing=True)


class _Timed:
    """flops = ALGORITHMIC work of the launch (the layer's real channel counts); exec_flops = what the kernel executed (>= flops where an operand is zero-padded
    to the tile width; None: the same) - bench.py prices roofline.achieved on the first and reports the second beside it"""

    def __init__(self, key, flops, tag=None, exec_flops=None):
        self.key, self.flops, self.tag = key, flops, tag
        self.exec_flops = flops if exec_flops is None else exec_flops

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = _event()
            self.e1 = _event()
            self.e0.record()

    def __exit__(self, *exc):
        if PROFILE is not None and exc[0] is None:
            self.e1.record()
            lib = load()            # which kernel configuration this launch ran (names the dominant kernel's symbol in bench.py's roofline)
            tag = self.tag or (lib.mis_conv_last_dispatch() if self.key[0] == "conv_igemm" else lib.mis_wgrad_last_dispatch()).decode()
            PROFILE.append((self.key, self.flops, self.e0, self.e1, tag, self.exec_flops))
        return False


def conv_igemm(x0, w_packed, y0, *, ksize, Cin, Cout, grid=None, x1=None, bias=None, relu=False, mask=None,
               y0_mode=OUT_PLAIN, y1=None, y1_mode=OUT_PLAIN, Cout0=None, in_scale=None, in_shift=None, relu_bits=None, mask_bits=None, gn_bwd=None, stats=None,
               real=None):
    """grid = (N, D, H, W) of the GEMM rows; defaults to x0's grid.
    relu_bits (out, uint8 tensor of relu_bits_bytes(N, H, W, Cout) bytes, with relu=True): bit = (output > 0); mask_bits (in): such bits applied instead of `mask`.
    gn_bwd = (p, q, r, relu_mask) with `mask` = the tensor x a GroupNorm in front of this convolution normalised (MisConvDesc.gn_p: bf16 3x3x3 dgrad on the ping-pong
    kernels): out = [relu_mask: (x > 0) *] (p * acc + q * x + r), p / q / r fp32 [N][ld] tables of gn_bwd_finalize; columns >= Cout0 are dropped when y1 is None.
    stats (MisConvDesc.st_mode; fp32 3x3x3 all-DMA kernels, check conv_stats_supported first) = dict(mode=1, x0=View, x1=View | None, up=bool, S1=, S2=): the GroupNorm-backward
    reductions S1 = sum out, S2 = sum out * x per (sample, column) from the epilogue, or dict(mode=2, S1=, S2=): sum out / sum out^2 of the stored output; S1 / S2 fp32 [N][Cout].
    real = (Cin, Cout) of the layer when the launch runs on zero-padded operands: the timing brackets then carry the algorithmic FLOPs (executed ones beside them)."""
    lib = load()
    x0 = _v(x0)
    y0 = _v(y0)
    d = ConvDesc()
    d.dtype = dtype_code(x0.dtype)
    d.ksize = ksize
    if grid is None:
        grid = (x0.N, x0.D, x0.H, x0.W)
    d.N, d.D, d.H, d.W = grid
    d.is3d = 1 if x0.t.dim() == 5 else 0
    d.Cin, d.Cout = Cin, Cout
    d.x0, d.x0_ld, d.x0_D, d.x0_H, d.x0_W = x0.ptr, x0.ld, x0.D, x0.H, x0.W
    d.Cin0 = x0.C if x1 is not None else Cin
    if x1 is not None:
        x1 = _v(x1)
        d.x1, d.x1_ld, d.x1_D, d.x1_H, d.x1_W = x1.ptr, x1.ld, x1.D, x1.H, x1.W
        assert x0.C + x1.C == Cin
    d.in_scale = None if in_scale is None else in_scale.data_ptr()
    d.in_shift = None if in_shift is None else in_shift.data_ptr()
    d.w = w_packed.data_ptr()
    d.bias = None if bias is None else bias.data_ptr()
    d.relu = 1 if relu else 0
    if mask is not None:
        mask = _v(mask)
        d.mask, d.mask_ld = mask.ptr, mask.ld
    d.y0, d.y0_ld, d.y0_mode = y0.ptr, y0.ld, y0_mode
    d.relu_bits = None if relu_bits is None else relu_bits.data_ptr()
    d.mask_bits = None if mask_bits is None else mask_bits.data_ptr()
    d.Cout0 = Cout if Cout0 is None else Cout0
    if y1 is not None:
        y1 = _v(y1)
        d.y1, d.y1_ld, d.y1_mode = y1.ptr, y1.ld, y1_mode
    if gn_bwd is not None:
        gp, gq, gr, grelu = gn_bwd
        assert gp.dtype == gq.dtype == gr.dtype == torch.float32 and gp.shape == gq.shape == gr.shape and gp.dim() == 2 and gp.is_contiguous() and gq.is_contiguous() and gr.is_contiguous()
        d.gn_p, d.gn_q, d.gn_r, d.gn_ld, d.gn_relu = gp.data_ptr(), gq.data_ptr(), gr.data_ptr(), gp.shape[1], 1 if grelu else 0
    taps = 1 if ksize == 1 else (27 if d.is3d else 9)
    key = ("conv_igemm", "bf16" if d.dtype == MIS_BF16 else "f32", f"k{ksize}", "3d" if d.is3d else "2d",
           "bn128" if Cout % 128 == 0 else "bn64", f"{d.N}x{d.D}x{d.H}x{d.W} {Cin}->{Cout}")
    if stats is not None:
        rows = lib.mis_conv_stats_rows(C.byref(d))
        if rows <= 0:
            raise MisError("conv_igemm(stats=...): this launch has no statistics epilogue (fp32 3x3x3 all-DMA path only)")
        part = workspace(d.N * rows * 2 * Cout * 4, y0.t.device, "convstats")[:d.N * rows * 2 * Cout].view(d.N, 1, 1, rows, 2 * Cout)
        d.st_mode, d.st_part = int(stats["mode"]), part.data_ptr()
        if d.st_mode == 1:
            sx0 = _v(stats["x0"])
            d.st_x0, d.st_x0_ld = sx0.ptr, sx0.ld
            if stats.get("x1") is not None:
                sx1 = _v(stats["x1"])
                assert sx0.C + sx1.C == Cout
                d.st_x1, d.st_x1_ld, d.st_c0, d.st_up = sx1.ptr, sx1.ld, sx0.C, 1 if stats.get("up") else 0
    rc_, rco_ = (Cin, Cout) if real is None else real
    with _Timed(key, 2.0 * d.N * d.D * d.H * d.W * taps * rc_ * rco_, exec_flops=2.0 * d.N * d.D * d.H * d.W * taps * Cin * Cout):
        check(lib.mis_conv_igemm(C.byref(d), stream_ptr()), "mis_conv_igemm")
    if stats is not None:
        # the partial rows -> [N][Cout] sums, added in double precision in a fixed order
        S1, S2 = stats["S1"], stats["S2"]
        assert S1.dtype == S2.dtype == torch.float32 and S1.is_contiguous() and S2.is_contiguous() and tuple(S1.shape) == tuple(S2.shape) == (d.N, Cout)
        ws2 = workspace(lib.mis_conv_stats_reduce_workspace_bytes(d.N, Cout), y0.t.device, "convstats.red")
        check(lib.mis_conv_stats_reduce(part.data_ptr(), d.N, rows, Cout, ws2.data_ptr(), S1.data_ptr(), S2.data_ptr(), stream_ptr()), "mis_conv_stats_reduce")


def conv_stats_supported(x0, y0, *, Cin, Cout, grid=None):
    """whether a plain single-source fp32 3x3x3 conv_igemm(x0 -> y0) would run on a kernel with the statistics epilogue (conv_igemm(stats=...))"""
    lib = load()
    x0, y0 = _v(x0), _v(y0)
    d = ConvDesc()
    d.dtype, d.ksize, d.is3d = dtype_code(x0.dtype), 3, 1 if x0.t.dim() == 5 else 0
    d.N, d.D, d.H, d.W = grid if grid is not None else (x0.N, x0.D, x0.H, x0.W)
    d.Cin = d.Cin0 = Cin
    d.Cout = d.Cout0 = Cout
    d.x0, d.x0_ld, d.x0_D, d.x0_H, d.x0_W = x0.ptr, x0.ld, x0.D, x0.H, x0.W
    d.y0, d.y0_ld = y0.ptr, y0.ld
    return lib.mis_conv_stats_rows(C.byref(d)) > 0


def dispatch_override(name, value=1):
    """set one kernel-selection switch of the library (include/misamd.h: mis_dispatch_override; e.g. "MIS_CONV_NOPP"); value < 0: back to the environment's value;
    name None: reset all.  Unknown names raise."""
    check(load().mis_dispatch_override(None if name is None else name.encode(), int(value)), "mis_dispatch_override")


def dispatch_switch(name):
    """current value of a kernel-selection switch (include/misamd.h: mis_dispatch_switch)"""
    v = load().mis_dispatch_switch(name.encode())
    if v < 0:
        check(v, "mis_dispatch_switch")
    return v


class dispatch_switches:
    """context manager: `with ops.dispatch_switches(MIS_CONV_NOPP=1): ...` - the switches are restored on exit"""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            dispatch_override(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            dispatch_override(k, -1)
        return False


def tile_queue_init():
    """allocate and zero the current device's tile-queue counter pool NOW (engine construction) rather than inside the first launch (include/misamd.h)"""
    check(load().mis_tile_queue_init(), "mis_tile_queue_init")


def tile_queue_reset():
    """zero the tile-queue counters of the current stream / of the capture running on it (a memset node): the engines call it at the start of every train step"""
    check(load().mis_tile_queue_reset(stream_ptr()), "mis_tile_queue_reset")


def tile_queue_check():
    """device-synchronising: raises MisError when a persistent kernel drew a ticket past its launch's last one since the previous check (that launch left output tiles
    unwritten - csrc/conv_pp_common.hpp tq_tile); tests, bench.py, smoke() and GraphedTrainStep call it"""
    lib = load()
    n = lib.mis_tile_queue_errors()
    if n != 0:
        raise MisError(lib.mis_last_error().decode("utf-8", "replace"))


def build_has_experiments():
    """True when libmisamd.so was built with `make EXPERIMENTS=1` (csrc/experiments: MIS_CONV_PPS / MIS_CONV_PPC2)"""
    return bool(load().mis_build_has_experiments())


def conv_last_dispatch():
    """name of the kernel configuration the last conv_igemm call of this thread ran (e.g. 'k3.2d.bn256.dma')"""
    return load().mis_conv_last_dispatch().decode()


def wgrad_last_dispatch():
    """(configuration name, split-K factor) of this thread's last wgrad call"""
    lib = load()
    return lib.mis_wgrad_last_dispatch().decode(), lib.mis_wgrad_last_nsplit()


_ws_cache = {}
_ws_captured = set()          # keys whose buffer a stream capture has baked into a graph
_ws_retired = []              # ... such buffers after they were outgrown: a graph may still replay into them - they are never handed back to the allocator


def _ws_key(tag, device):
    return (tag, str(device))


def workspace(nbytes, device, tag="default"):
    """Grow-only scratch buffers (caller-owned workspaces of the C ABI), one per (tag, device).  A buffer that a stream capture has used stays alive when a larger request
    replaces it (a second engine with a larger shape used to release the buffer a captured GraphedTrainStep of the first still wrote its slabs into)."""
    key = _ws_key(tag, device)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        if buf is not None and key in _ws_captured:
            _ws_retired.append(buf)
            _ws_captured.discard(key)
        buf = torch.empty((max(nbytes, 1) + 3) // 4, dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    if key not in _ws_captured and buf.is_cuda and torch.cuda.is_current_stream_capturing():
        _ws_captured.add(key)
    return buf


class _SideReduce:
    """Second stream for the slab reductions of wgrad (HBM-bound) so that they run under the caller's next MFMA kernel.
    Two workspaces alternate; a workspace is reused only after the reduction that read it has finished (event wait)."""

    def __init__(self):
        self.per_device = {}

    def state(self, device):
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        st = self.per_device.get(str(device))
        if st is None:
            st = {"stream": torch.cuda.Stream(device=device), "events": [None, None], "n": 0}
            self.per_device[str(device)] = st
        return st


_side = _SideReduce()


def wgrad_join(device=None):
    """Make the current stream wait for every side-stream reduction issued so far (before dw / dbias are read)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:                      # "cuda" and "cuda:0" are the same device: the table is keyed by the indexed form
        dev = torch.device("cuda", torch.cuda.current_device())
    st = _side.per_device.get(str(dev))
    if st is not None:
        torch.cuda.current_stream(dev).wait_stream(st["stream"])


def wgrad_reduce_batch(items):
    """reduce the split-K slabs of the weight gradients deferred into `items` (wgrad(..., defer=items)): two launches per 16 layers; empties the list"""
    lib = load()
    for i in range(0, len(items), 16):
        chunk = items[i:i + 16]
        arr = (WgradReduceItem * len(chunk))(*chunk)
        check(lib.mis_wgrad_reduce_batch(arr, len(chunk), stream_ptr()), "mis_wgrad_reduce_batch")
    del items[:]


def wgrad(x0, dy, dw, *, ksize, Cin, Cout, grid=None, x1=None, dw_layout=0, alpha=1.0, in_scale=None, in_shift=None, dbias=None, side=False,
          dw_per_sample=None, dbias_per_sample=None, defer=None, ws_tag=None, real=None):
    """side=True: the reduction kernels go to a second stream (see _SideReduce); the caller must call wgrad_join() before using dw.
    defer=list: only the MFMA kernel runs; the slab reduction is described by an item appended to the list and done for a group of layers by wgrad_reduce_batch(list)
    (the slabs live in a workspace of this layer's own, `ws_tag`, until then)."""
    lib = load()
    x0 = _v(x0)
    dy = _v(dy)
    d = WgradDesc()
    d.dtype = dtype_code(x0.dtype)
    d.ksize = ksize
    if grid is None:
        grid = (dy.N, dy.D, dy.H, dy.W)
    d.N, d.D, d.H, d.W = grid
    d.is3d = 1 if dy.t.dim() == 5 else 0
    d.Cin, d.Cout = Cin, Cout
    d.x0, d.x0_ld, d.x0_D, d.x0_H, d.x0_W = x0.ptr, x0.ld, x0.D, x0.H, x0.W
    d.Cin0 = x0.C if x1 is not None else Cin
    if x1 is not None:
        x1 = _v(x1)
        d.x1, d.x1_ld, d.x1_D, d.x1_H, d.x1_W = x1.ptr, x1.ld, x1.D, x1.H, x1.W
    d.in_scale = None if in_scale is None else in_scale.data_ptr()
    d.in_shift = None if in_shift is None else in_shift.data_ptr()
    d.dy, d.dy_ld = dy.ptr, dy.ld
    d.dw, d.dw_layout, d.alpha = dw.data_ptr(), dw_layout, alpha
    d.dbias = None if dbias is None else dbias.data_ptr()
    d.dw_per_sample = None if dw_per_sample is None else dw_per_sample.data_ptr()
    d.dbias_per_sample = None if dbias_per_sample is None else dbias_per_sample.data_ptr()
    need = lib.mis_wgrad_workspace_bytes(C.byref(d))
    if need == 0:
        check(-1, "mis_wgrad_workspace_bytes")
    st = None
    if side:
        st = _side.state(dy.t.device)
        slot = st["n"] & 1
        st["n"] += 1
        cur = _ws_cache.get(_ws_key(f"wgrad{slot}", dy.t.device))
        if cur is None or cur.numel() * 4 < need:     # growing = freeing the old buffer: no side-stream reduction may still be reading it
            st["stream"].synchronize()
        ws = workspace(need, dy.t.device, f"wgrad{slot}")
        if st["events"][slot] is not None:            # the reduction that last read this workspace must be done
            torch.cuda.current_stream(dy.t.device).wait_event(st["events"][slot])
        d.reduce_stream = st["stream"].cuda_stream
    elif defer is not None:
        if ws_tag is None:
            raise MisError("wgrad(defer=...): ws_tag (a workspace of the layer's own) is required")
        ws = workspace(need, dy.t.device, "wgrad:" + ws_tag)
        item = WgradReduceItem()
        d.defer = C.addressof(item)
    else:
        ws = workspace(need, dy.t.device, "wgrad")
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    taps = 1 if ksize == 1 else (27 if d.is3d else 9)
    key = ("wgrad", "bf16" if d.dtype == MIS_BF16 else "f32", f"k{ksize}", "3d" if d.is3d else "2d", "",
           f"{d.N}x{d.D}x{d.H}x{d.W} {Cin}->{Cout}")
    rc_, rco_ = (Cin, Cout) if real is None else real
    with _Timed(key, 2.0 * d.N * d.D * d.H * d.W * taps * rc_ * rco_, exec_flops=2.0 * d.N * d.D * d.H * d.W * taps * Cin * Cout):
        check(lib.mis_wgrad(C.byref(d), stream_ptr()), "mis_wgrad")
    if st is not None:
        ev = torch.cuda.Event()
        ev.record(st["stream"])
        st["events"][slot] = ev
    if defer is not None:
        defer.append(item)


def first_conv_fwd(x_nchw, w, bias, y, relu_bits=None):
    lib = load()
    y = _v(y)
    N, Cin, H, W = x_nchw.shape
    check(lib.mis_conv3x3_first_fwd_rb(dtype_code(y.dtype), x_nchw.data_ptr(), N, Cin, H, W, w.data_ptr(),
                                       None if bias is None else bias.data_ptr(), y.ptr, y.ld, y.C,
                                       None if relu_bits is None else relu_bits.data_ptr(), stream_ptr()),
          "mis_conv3x3_first_fwd")


def relu_bits_bytes(N, H, W, C):
    """size in bytes of the ReLU-bits tensor of a bf16 (N, H, W, C) activation (csrc/relu_bits.hpp; C % 64 == 0)"""
    return int(load().mis_relu_bits_bytes(int(N), int(H), int(W), int(C)))


def relu_bits(y, bits):
    """bits = (y > 0), one bit per element of the bf16 NHWC view y, in the layout the convolutions' `mask_bits` reads (the stand-alone producer)"""
    y = _v(y)
    check(load().mis_relu_bits(y.ptr, y.ld, y.N, y.H, y.W, y.C, bits.data_ptr(), stream_ptr()), "mis_relu_bits")


def first_conv_wgrad(x_nchw, dy, dw, db):
    lib = load()
    dy = _v(dy)
    N, Cin, H, W = x_nchw.shape
    ws = workspace(lib.mis_conv3x3_first_wgrad_workspace_bytes(N, Cin, H, W, dy.C), x_nchw.device, "first")
    check(lib.mis_conv3x3_first_wgrad(dtype_code(dy.dtype), x_nchw.data_ptr(), N, Cin, H, W, dy.ptr, dy.ld, dy.C,
                                      ws.data_ptr(), dw.data_ptr(), None if db is None else db.data_ptr(), stream_ptr()),
          "mis_conv3x3_first_wgrad")


def colsum(x, out, fold=1, alpha=1.0):
    lib = load()
    x = _v(x)
    ws = workspace(lib.mis_colsum_workspace_bytes(x.npix, x.C), x.t.device, "colsum")
    check(lib.mis_colsum(dtype_code(x.dtype), x.ptr, x.ld, x.npix, x.C, fold, alpha, ws.data_ptr(), out.data_ptr(),
                         stream_ptr()), "mis_colsum")


def chanstats(x, s, sq):
    lib = load()
    x = _v(x)
    npix = x.D * x.H * x.W
    ws = workspace(lib.mis_chanstats_workspace_bytes(x.N, npix, x.C), x.t.device, "chanstats")
    check(lib.mis_chanstats(dtype_code(x.dtype), x.ptr, x.ld, x.N, npix, x.C, ws.data_ptr(), s.data_ptr(), sq.data_ptr(),
                            stream_ptr()), "mis_chanstats")


def maxpool2_fwd(x, y, pbits=None):
    """pbits (2-D, even H and W): uint8 (N, H/2, W/2, C) "pool bits" for maxpool2_bwd(pbits=...) - arg-max position and input sign per pooled element"""
    lib = load()
    x, y = _v(x), _v(y)
    if pbits is not None:
        check(lib.mis_maxpool2_fwd_pb(dtype_code(x.dtype), x.ptr, x.ld, y.ptr, y.ld, x.N, x.H, x.W, x.C, pbits.data_ptr(), stream_ptr()), "mis_maxpool2_fwd_pb")
        return
    check(lib.mis_maxpool2_fwd(dtype_code(x.dtype), x.ptr, x.ld, y.ptr, y.ld, x.N, x.D, x.H, x.W, x.C, stream_ptr()),
          "mis_maxpool2_fwd")


def maxpool2_bwd(x, dy, dx, add=None, relu_mask=True, pbits=None):
    """pbits (from maxpool2_fwd): the backward pass with the ReLU mask, without reading x (x may then be None)"""
    lib = load()
    dy, dx = _v(dy), _v(dx)
    if add is not None:
        add = _v(add)
    if pbits is not None:
        assert relu_mask, "pool bits carry the ReLU mask"
        check(lib.mis_maxpool2_bwd_pb(dtype_code(dx.dtype), pbits.data_ptr(), dy.ptr, dy.ld, None if add is None else add.ptr, 0 if add is None else add.ld,
                                      dx.ptr, dx.ld, dx.N, dx.H, dx.W, dx.C, stream_ptr()), "mis_maxpool2_bwd_pb")
        return
    x = _v(x)
    check(lib.mis_maxpool2_bwd(dtype_code(x.dtype), x.ptr, x.ld, dy.ptr, dy.ld, None if add is None else add.ptr,
                               0 if add is None else add.ld, dx.ptr, dx.ld, x.N, x.D, x.H, x.W, x.C, 1 if relu_mask else 0,
                               stream_ptr()), "mis_maxpool2_bwd")


def pack_conv_weight(w, w_fwd, w_dgrad=None):
    """w: fp32 [Cout, Cin, *k] (reference layout) -> packed operands (dtype of w_fwd)."""
    lib = load()
    Cout, Cin = w.shape[0], w.shape[1]
    taps = w[0, 0].numel()
    check(lib.mis_pack_conv_weight(dtype_code(w_fwd.dtype), w.data_ptr(), Cout, Cin, taps, w_fwd.data_ptr(),
                                   None if w_dgrad is None else w_dgrad.data_ptr(), stream_ptr()), "mis_pack_conv_weight")


class PackTable:
    """device-resident table for `pack_batch`: entries (w fp32, w_fwd, w_dgrad | None, kind) with kind 0 = pack_conv_weight layout (w [Cout, Cin, *k]),
    kind 1 = pack_convt_weight (w [Cin, Cq, 2, 2]).  The tensors' addresses are baked in: rebuild it if a buffer is re-allocated."""

    def __init__(self, entries, device):
        import numpy as np
        from ._lib import pack_item2_dtype
        dt = pack_item2_dtype()           # MisPackItem2 (include/misamd.h): the compact-grid form; the mirror is checked against the library's own layout at load time
        tab = np.zeros(len(entries), dtype=dt)
        self.keep = []
        dts = set()
        blk = 0
        for i, (w, wf, wd, kind) in enumerate(entries):
            assert w.dtype == torch.float32 and w.is_contiguous() and wf.is_contiguous() and (wd is None or wd.is_contiguous())
            rows, cols = w.shape[0], w.shape[1]
            taps = w[0, 0].numel() if kind == 0 else 4
            nbx, nby = (cols + 31) // 32, (rows + 31) // 32
            tab[i] = (w.data_ptr(), wf.data_ptr(), 0 if wd is None else wd.data_ptr(), rows, cols, taps, kind, blk, nbx)
            blk += nbx * nby
            dts.add(wf.dtype)
            self.keep.append((w, wf, wd))
        self.total_blocks = blk
        assert len(dts) == 1
        self.dtype = dts.pop()
        self.n = len(entries)
        self.max_rows, self.max_cols = int(tab["rows"].max()), int(tab["cols"].max())
        self.ptrs = tuple(int(v) for v in tab["w"]) + tuple(int(v) for v in tab["w_fwd"])
        self.dev = torch.from_numpy(tab.view(np.uint8).reshape(-1).copy()).to(device)


def pack_batch(table):
    """every weight repack of a network in one launch (mis_pack_batch)"""
    lib = load()
    check(lib.mis_pack_batch2(dtype_code(table.dtype), table.dev.data_ptr(), table.n, table.total_blocks, stream_ptr()), "mis_pack_batch2")


def pack_convt_weight(w, w_fwd, w_dgrad):
    lib = load()
    Cin, Cq = w.shape[0], w.shape[1]
    check(lib.mis_pack_convt_weight(dtype_code(w_fwd.dtype), w.data_ptr(), Cin, Cq, w_fwd.data_ptr(), w_dgrad.data_ptr(),
                                    stream_ptr()), "mis_pack_convt_weight")


LOSS_NONE, LOSS_CE, LOSS_BCE, LOSS_BCEDICE, LOSS_EXTERNAL = -1, 0, 1, 2, 3


def head_loss(y, w, b, *, loss, labels=None, logits=None, argmax=None, loss_out=None, dy=None, dw=None, db=None,
              grad_scale=1.0, alpha=1.0, beta=1.0, phase=0):
    lib = load()
    y = _v(y)
    d = HeadDesc()
    d.dtype = dtype_code(y.dtype)
    d.loss = loss
    d.npix_per_image = y.D * y.H * y.W
    d.N, d.Cfeat, d.C = y.N, y.C, w.shape[0]
    d.y, d.y_ld = y.ptr, y.ld
    d.w, d.b = w.data_ptr(), b.data_ptr()
    d.labels = None if labels is None else labels.data_ptr()
    d.logits = None if logits is None else logits.data_ptr()
    d.argmax = None if argmax is None else argmax.data_ptr()
    d.loss_out = None if loss_out is None else loss_out.data_ptr()
    if dy is not None:
        dy = _v(dy)
        d.dy, d.dy_ld = dy.ptr, dy.ld
        d.dw, d.db = dw.data_ptr(), db.data_ptr()
    d.grad_scale, d.alpha, d.beta, d.phase = grad_scale, alpha, beta, phase
    ws = workspace(lib.mis_head_workspace_bytes(C.byref(d)), y.t.device, "head")
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    check(lib.mis_head_loss(C.byref(d), stream_ptr()), "mis_head_loss")


def conv3x3_head_fused(x, w_packed, bias, dy, wh, bh, *, Cin, loss, labels, logits, argmax, loss_out, dw, db, grad_scale=1.0, dry_run=False):
    """up_conv.3.second (3x3, Cin -> 64, bias + ReLU) with the 1x1 head, the loss and their backward in its epilogue (include/misamd.h: mis_conv3x3_head_fused):
    `dy` (N, H, W, 64) receives dL/dfeatures where the features would have been written.  Returns False (nothing launched) when the configuration is not eligible;
    dry_run=True only asks."""
    lib = load()
    x, dy = _v(x), _v(dy)
    c = ConvDesc()
    c.dtype, c.ksize = dtype_code(x.dtype), 3
    c.N, c.D, c.H, c.W, c.is3d = x.N, 1, x.H, x.W, 0
    c.Cin, c.Cout, c.Cin0, c.Cout0 = Cin, 64, Cin, 64
    c.x0, c.x0_ld, c.x0_D, c.x0_H, c.x0_W = x.ptr, x.ld, 1, x.H, x.W
    c.w, c.bias, c.relu = w_packed.data_ptr(), bias.data_ptr(), 1
    c.y0, c.y0_ld, c.y0_mode = dy.ptr, dy.ld, OUT_PLAIN
    h = HeadDesc()
    h.dtype, h.loss = c.dtype, loss
    h.npix_per_image = x.H * x.W
    h.N, h.Cfeat, h.C = x.N, 64, wh.shape[0]
    h.y, h.y_ld = dy.ptr, dy.ld                          # (unused by the fused kernel: the features are never stored)
    h.w, h.b = wh.data_ptr(), bh.data_ptr()
    h.labels = labels.data_ptr()
    h.logits = None if logits is None else logits.data_ptr()
    h.argmax = None if argmax is None else argmax.data_ptr()
    h.loss_out = loss_out.data_ptr()
    h.dy, h.dy_ld = dy.ptr, dy.ld
    h.dw, h.db = dw.data_ptr(), db.data_ptr()
    h.grad_scale, h.alpha, h.beta, h.phase = grad_scale, 1.0, 1.0, 0
    ws = workspace(lib.mis_head_workspace_bytes(C.byref(h)), x.t.device, "head")
    h.workspace, h.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    if not lib.mis_conv3x3_head_fused_eligible(C.byref(c), C.byref(h)):
        return False
    if dry_run:
        return True
    key = ("conv_igemm", "bf16", "k3", "2d", "bn64", f"{c.N}x1x{c.H}x{c.W} {Cin}->64+head")
    with _Timed(key, 2.0 * c.N * c.H * c.W * 9 * Cin * 64, tag="k3.2d.ppd8.head"):
        check(lib.mis_conv3x3_head_fused(C.byref(c), C.byref(h), stream_ptr()), "mis_conv3x3_head_fused")
    return True


def sumsq(g, partials):
    lib = load()
    check(lib.mis_sumsq(g.data_ptr(), g.numel(), partials.data_ptr(), stream_ptr()), "mis_sumsq")


def sumsq_npartials(n):
    return load().mis_sumsq_npartials(n)


def adamw_step(p, g, m, v, *, partials, max_norm, lr, beta1, beta2, eps, weight_decay, step, gradnorm_out=None):
    lib = load()
    check(lib.mis_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                             None if partials is None else partials.data_ptr(), 0 if partials is None else partials.numel(),
                             max_norm, lr, beta1, beta2, eps, weight_decay, step,
                             None if gradnorm_out is None else gradnorm_out.data_ptr(), stream_ptr()), "mis_adamw_step")


def adamw_step_dev(p, g, m, v, *, partials, max_norm, lr_dev, beta1, beta2, eps, weight_decay, step_dev, advance, hyper, gradnorm_out=None):
    """AdamW with the step counter and the learning rate in device memory (captured train steps, graph.GraphedTrainStep)"""
    lib = load()
    check(lib.mis_adamw_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                 None if partials is None else partials.data_ptr(), 0 if partials is None else partials.numel(), max_norm,
                                 lr_dev.data_ptr(), beta1, beta2, eps, weight_decay, step_dev.data_ptr(), 1 if advance else 0, hyper.data_ptr(),
                                 None if gradnorm_out is None else gradnorm_out.data_ptr(), stream_ptr()), "mis_adamw_step_dev")


def nchw_to_nhwc(x, y):
    """x: fp32 (N, C, *spatial) -> y channels-last view (dtype of y)."""
    lib = load()
    y = _v(y)
    N, Cc = x.shape[0], x.shape[1]
    S = x[0, 0].numel()
    check(lib.mis_nchw_to_nhwc(dtype_code(y.dtype), x.data_ptr(), y.ptr, y.ld, N, Cc, S, stream_ptr()), "mis_nchw_to_nhwc")


def nhwc_to_nchw(x, y):
    lib = load()
    x = _v(x)
    S = x.D * x.H * x.W
    check(lib.mis_nhwc_to_nchw(dtype_code(x.dtype), x.ptr, x.ld, y.data_ptr(), x.N, x.C, S, stream_ptr()), "mis_nhwc_to_nchw")


def probe_mfma(which, a, b, c):
    lib = load()
    check(lib.mis_probe_mfma(which, a.data_ptr(), None if b is None else b.data_ptr(), c.data_ptr(), stream_ptr()),
          "mis_probe_mfma")


# ---- GroupNorm / 3-D helpers ---------------------------------------------------------------------------
def gn_fwd_finalize(sum0, sq0, C0, mult0, sum1, sq1, C1, mult1, N, G, count, gamma, beta, Cpad, scale, shift, mean, rstd, eps=1e-5):
    lib = load()
    check(lib.mis_gn_fwd_finalize(sum0.data_ptr(), sq0.data_ptr(), C0, mult0, None if sum1 is None else sum1.data_ptr(),
                                  None if sq1 is None else sq1.data_ptr(), C1, mult1, N, G, float(count), gamma.data_ptr(), beta.data_ptr(),
                                  eps, Cpad, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), stream_ptr()),
          "mis_gn_fwd_finalize")


def gn_apply(x, Cs, up, grid, scale, shift, Ctot, c_off, y):
    """materialise the GroupNorm output of one source: y[..., c_off:c_off+Cs] = scale * x(src voxel) + shift (mis_gn_apply); y = the destination tensor (N, D, H, W, >= Ctot)"""
    lib = load()
    x = _v(x)
    N, D, H, W = grid
    if not y.is_contiguous() or tuple(y.shape[:4]) != (N, D, H, W) or y.shape[-1] < Ctot:
        raise MisError(f"gn_apply: destination {tuple(y.shape)} does not match grid {grid} / {Ctot} channels")
    check(lib.mis_gn_apply(dtype_code(x.dtype), x.ptr, x.ld, Cs, 1 if up else 0, N, D, H, W, scale.data_ptr(), shift.data_ptr(), Ctot, c_off, y.data_ptr(),
                           y.shape[-1], stream_ptr()), "mis_gn_apply")


def gn_bwd_stats(dy, x, Cs, up, grid, S1, S2, Ctot, c_off):
    """dy: full-grid View over ALL Ctot channels; x: the source tensor View (Cs channels)."""
    lib = load()
    dy, x = _v(dy), _v(x)
    N, D, H, W = grid
    ws = workspace(lib.mis_gn_bwd_stats_workspace_bytes(N, Cs), x.t.device, "gnbwd")
    check(lib.mis_gn_bwd_stats(dtype_code(x.dtype), dy.ptr, dy.ld, x.ptr, x.ld, Cs, 1 if up else 0, N, D, H, W, ws.data_ptr(),
                               S1.data_ptr(), S2.data_ptr(), Ctot, c_off, stream_ptr()), "mis_gn_bwd_stats")


def gn_bwd_stats_from_dw(gy, w, dw_per_sample, gy_colsum, scale, shift, mean, groups, Cs, S1, S2):
    """S1 / S2 of the GroupNorm backward from the per-sample weight gradients and border sums of gy (mis_gn_bwd_stats_from_dw): no pass over dyn and x.
    gy: View (N, D, H, W, Cout); w, dw_per_sample[n]: (Cout, Cw, 3, 3, 3) fp32; scale / shift: (N, >= Cs); mean: (N, groups)"""
    lib = load()
    gy = _v(gy)
    Cout, Cw = w.shape[0], w.shape[1]
    ws = workspace(lib.mis_gn_bwd_stats_from_dw_workspace_bytes(gy.N, Cout), gy.t.device, "gn_from_dw")
    check(lib.mis_gn_bwd_stats_from_dw(dtype_code(gy.dtype), gy.ptr, gy.ld, gy.N, gy.D, gy.H, gy.W, Cout, w.data_ptr(), dw_per_sample.data_ptr(), Cw,
                                       gy_colsum.data_ptr(), scale.data_ptr(), shift.data_ptr(), scale.shape[-1], mean.data_ptr(), groups, Cs, ws.data_ptr(),
                                       S1.data_ptr(), S2.data_ptr(), stream_ptr()), "mis_gn_bwd_stats_from_dw")


class GnCondTable:
    """host-side description of every GroupNorm (gamma, beta) pair inside ONE flat fp32 parameter buffer, for gn_cond"""

    def __init__(self, flat, pairs):
        n = len(pairs)
        self.n = n
        self.flat = flat
        base = flat.data_ptr()
        self.goff = (C.c_ulonglong * n)(*[(g.data_ptr() - base) // 4 for g, _ in pairs])
        self.boff = (C.c_ulonglong * n)(*[(b.data_ptr() - base) // 4 for _, b in pairs])
        self.cnt = (C.c_int * n)(*[g.numel() for g, _ in pairs])


def gn_cond(table, ratio, flags):
    """flags[l] (int32, device) = 1 when any channel of GroupNorm layer l has |gamma| < ratio * |beta| (include/misamd.h: mis_gn_cond); no synchronisation"""
    check(load().mis_gn_cond(table.flat.data_ptr(), table.goff, table.boff, table.cnt, table.n, float(ratio), flags.data_ptr(), stream_ptr()), "mis_gn_cond")


def gn_bwd_finalize(S1, S2, mean, rstd, gamma, N, Cc, G, count, p, q, r, dgamma, dbeta):
    lib = load()
    check(lib.mis_gn_bwd_finalize(S1.data_ptr(), S2.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), N, Cc, G, float(count),
                                  p.data_ptr(), q.data_ptr(), r.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), stream_ptr()),
          "mis_gn_bwd_finalize")


def gn_bwd_apply(dy, x, Cs, up, grid, p, q, r, Ctot, c_off, dx, relu_mask=False, add=None):
    lib = load()
    dy, x, dx = _v(dy), _v(x), _v(dx)
    if add is not None:
        add = _v(add)
    N, D, H, W = grid
    check(lib.mis_gn_bwd_apply(dtype_code(x.dtype), dy.ptr, dy.ld, x.ptr, x.ld, Cs, 1 if up else 0, N, D, H, W, p.data_ptr(), q.data_ptr(),
                               r.data_ptr(), Ctot, c_off, 1 if relu_mask else 0, None if add is None else add.ptr,
                               0 if add is None else add.ld, dx.ptr, dx.ld, stream_ptr()), "mis_gn_bwd_apply")


def first3d_fwd(x, scale, shift, sstride, w, Cout, y, Cpad):
    lib = load()
    y = _v(y)
    N, D, H, W = x.shape[0], x.shape[-3], x.shape[-2], x.shape[-1]
    check(lib.mis_first3d_fwd(dtype_code(y.dtype), x.data_ptr(), scale.data_ptr(), shift.data_ptr(), sstride, N, D, H, W, w.data_ptr(), Cout,
                              y.ptr, y.ld, Cpad, stream_ptr()), "mis_first3d_fwd")


def first3d_bwd(x, mean, rstd, gamma, beta, dy, Cpad, w, Cout, dw, dgamma, dbeta, dxn=None):
    """first 3-D layer backward: dW, and the 1-channel GroupNorm's dgamma / dbeta (device scalars); dxn optional."""
    lib = load()
    dy = _v(dy)
    N, D, H, W = x.shape[0], x.shape[-3], x.shape[-2], x.shape[-1]
    ws = workspace(lib.mis_first3d_bwd_workspace_bytes(), x.device, "first3d")
    check(lib.mis_first3d_bwd(dtype_code(dy.dtype), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), N, D, H, W,
                              dy.ptr, dy.ld, Cpad, w.data_ptr(), Cout, ws.data_ptr(), dw.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                              None if dxn is None else dxn.data_ptr(), stream_ptr()), "mis_first3d_bwd")


def relu_mask(dy, y, dx):
    lib = load()
    dy, y, dx = _v(dy), _v(y), _v(dx)
    check(lib.mis_relu_mask(dtype_code(y.dtype), dy.ptr, dy.ld, y.ptr, y.ld, dx.ptr, dx.ld, y.npix, y.C, stream_ptr()), "mis_relu_mask")


def bn_fwd_finalize(s, sq, N, Cc, count_total, gamma, beta, running_mean, running_var, training, scale, shift, mean, rstd, eps=1e-5, momentum=0.1):
    lib = load()
    check(lib.mis_bn_fwd_finalize(None if s is None else s.data_ptr(), None if sq is None else sq.data_ptr(), N, Cc, float(count_total),
                                  gamma.data_ptr(), beta.data_ptr(), eps, momentum, None if running_mean is None else running_mean.data_ptr(),
                                  None if running_var is None else running_var.data_ptr(), 1 if training else 0, scale.data_ptr(),
                                  shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), stream_ptr()), "mis_bn_fwd_finalize")


def bn_bwd_finalize(S1, S2, mean, rstd, gamma, N, Cc, count_total, training, p, q, r, dgamma, dbeta):
    lib = load()
    check(lib.mis_bn_bwd_finalize(S1.data_ptr(), S2.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), N, Cc, float(count_total),
                                  1 if training else 0, p.data_ptr(), q.data_ptr(), r.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                  stream_ptr()), "mis_bn_bwd_finalize")


def affine_act(x, y, scale, shift, relu=True):
    lib = load()
    x, y = _v(x), _v(y)
    check(lib.mis_affine_act(dtype_code(x.dtype), x.ptr, x.ld, y.ptr, y.ld, x.N, x.D * x.H * x.W, x.C, scale.data_ptr(), shift.data_ptr(),
                             1 if relu else 0, stream_ptr()), "mis_affine_act")


def convt3_col2im(cols, u):
    """cols: (N, d, h, w, 27*C) contiguous; u: View/tensor (N, 2d, 2h, 2w, C)."""
    lib = load()
    u = _v(u)
    N, d, h, w = cols.shape[:4]
    check(lib.mis_convt3_col2im(dtype_code(cols.dtype), cols.data_ptr(), u.ptr, u.ld, N, d, h, w, u.C, stream_ptr()), "mis_convt3_col2im")


def convt3_im2col(gu, gcols):
    lib = load()
    gu = _v(gu)
    N, d, h, w = gcols.shape[:4]
    check(lib.mis_convt3_im2col(dtype_code(gcols.dtype), gu.ptr, gu.ld, gcols.data_ptr(), N, d, h, w, gu.C, stream_ptr()), "mis_convt3_im2col")


def maxpoolk_fwd(x, y, k):
    lib = load()
    x, y = _v(x), _v(y)
    check(lib.mis_maxpoolk_fwd(dtype_code(x.dtype), x.ptr, x.ld, y.ptr, y.ld, x.N, x.H, x.W, x.C, k, stream_ptr()), "mis_maxpoolk_fwd")


def maxpoolk_bwd(x, dy, dx, k):
    lib = load()
    x, dy, dx = _v(x), _v(dy), _v(dx)
    check(lib.mis_maxpoolk_bwd(dtype_code(x.dtype), x.ptr, x.ld, dy.ptr, dy.ld, dx.ptr, dx.ld, x.N, x.H, x.W, x.C, k, stream_ptr()), "mis_maxpoolk_bwd")


def bilinear_up_fwd(x, y, scale):
    lib = load()
    x, y = _v(x), _v(y)
    check(lib.mis_bilinear_up_fwd(dtype_code(x.dtype), x.ptr, x.ld, y.ptr, y.ld, x.N, x.H, x.W, x.C, scale, stream_ptr()), "mis_bilinear_up_fwd")


def bilinear_up_bwd(dy, dx, scale):
    lib = load()
    dy, dx = _v(dy), _v(dx)
    ws = workspace(lib.mis_bilinear_up_bwd_workspace_bytes(dx.N, dx.H, dx.W, dx.C, scale), dx.t.device, "bilinear")
    check(lib.mis_bilinear_up_bwd(dtype_code(dx.dtype), dy.ptr, dy.ld, dx.ptr, dx.ld, dx.N, dx.H, dx.W, dx.C, scale, ws.data_ptr(), stream_ptr()),
          "mis_bilinear_up_bwd")


def bn_bwd_stats(dy, z, scale, shift, S1, S2):
    """S1[n][c] = sum m*dy, S2 = sum m*dy*z with the ReLU mask m = (z*scale + shift > 0) recomputed from z"""
    lib = load()
    dy, z = _v(dy), _v(z)
    npix = z.D * z.H * z.W
    ws = workspace(lib.mis_bn_bwd_stats_workspace_bytes(z.N, z.C), z.t.device, "bn_bwd")
    check(lib.mis_bn_bwd_stats(dtype_code(z.dtype), dy.ptr, dy.ld, z.ptr, z.ld, z.N, npix, z.C, scale.data_ptr(), shift.data_ptr(), ws.data_ptr(),
                               S1.data_ptr(), S2.data_ptr(), stream_ptr()), "mis_bn_bwd_stats")


def bn_bwd_apply(dy, z, scale, shift, p, q, r, dz):
    lib = load()
    dy, z, dz = _v(dy), _v(z), _v(dz)
    npix = z.D * z.H * z.W
    check(lib.mis_bn_bwd_apply(dtype_code(z.dtype), dy.ptr, dy.ld, z.ptr, z.ld, z.N, npix, z.C, scale.data_ptr(), shift.data_ptr(), p.data_ptr(),
                               q.data_ptr(), r.data_ptr(), dz.ptr, dz.ld, stream_ptr()), "mis_bn_bwd_apply")


def se_fwd(e, y, W1, b1, W2, b2, w, b0, st):
    """scSE forward (csrc/se3d.hip) on a channels-last activation e -> y; `st` carries the per-block buffers sum, sq, mean, z1, a [N][C], bgate [N][S]"""
    lib = load()
    e, y = _v(e), _v(y)
    S = e.D * e.H * e.W
    chanstats(e, st.sum, st.sq)
    check(lib.mis_se_fc_fwd(st.sum.data_ptr(), float(S), W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(), e.N, e.C, st.mean.data_ptr(),
                            st.z1.data_ptr(), st.a.data_ptr(), stream_ptr()), "mis_se_fc_fwd")
    check(lib.mis_se_apply_fwd(dtype_code(e.dtype), e.ptr, e.ld, e.N, S, e.C, st.a.data_ptr(), w.data_ptr(), b0.data_ptr(), st.bgate.data_ptr(), y.ptr, y.ld,
                               stream_ptr()), "mis_se_apply_fwd")


def se_bwd(g, e, W1, W2, w, st, dW1, db1, dW2, db2, dw, db0):
    """scSE backward + the ReLU mask of e, in place on g (dL/d(out) -> dL/d(pre-activation of e)); parameter gradients are written (not accumulated)"""
    lib = load()
    g, e = _v(g), _v(e)
    S = e.D * e.H * e.W
    ws = workspace(lib.mis_se_bwd_workspace_bytes(e.N, e.C), e.t.device, "se_bwd")
    dt = dtype_code(e.dtype)
    check(lib.mis_se_bwd_reduce(dt, g.ptr, g.ld, e.ptr, e.ld, e.N, S, e.C, st.a.data_ptr(), st.bgate.data_ptr(), ws.data_ptr(), st.dq.data_ptr(),
                                st.da.data_ptr(), dw.data_ptr(), db0.data_ptr(), stream_ptr()), "mis_se_bwd_reduce")
    check(lib.mis_se_fc_bwd(st.da.data_ptr(), st.a.data_ptr(), st.z1.data_ptr(), st.mean.data_ptr(), W1.data_ptr(), W2.data_ptr(), e.N, e.C, float(S),
                            ws.data_ptr(), dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(), st.cross.data_ptr(), stream_ptr()), "mis_se_fc_bwd")
    check(lib.mis_se_bwd_apply(dt, g.ptr, g.ld, e.ptr, e.ld, e.N, S, e.C, st.a.data_ptr(), st.bgate.data_ptr(), st.dq.data_ptr(), w.data_ptr(),
                               st.cross.data_ptr(), g.ptr, g.ld, stream_ptr()), "mis_se_bwd_apply")


SE_SCSE, SE_CSE, SE_SSE = 0, 1, 2


def se_layer_fwd(e, y, mode, W1=None, b1=None, W2=None, b2=None, w=None, b0=None):
    """one of se.py's three layers on its own (csrc/se3d.hip, mis_se_layer_*): e -> y channels-last, mode SE_SCSE / SE_CSE / SE_SSE.  W1, W2: [C][C] fp32 (the
    caller zero-pads other reduction ratios), w [C], b0 [1].  Returns the tuple of saved fp32 state for se_layer_bwd: (mean, z1, a, bgate) - None where unused."""
    lib = load()
    ev, yv = _v(e), _v(y)
    S = ev.D * ev.H * ev.W
    dev = ev.t.device
    mean = z1 = a = bgate = None
    if mode != SE_SSE:
        csum, csq = torch.empty(ev.N, ev.C, device=dev), torch.empty(ev.N, ev.C, device=dev)
        chanstats(e, csum, csq)
        mean, z1, a = (torch.empty(ev.N, ev.C, device=dev) for _ in range(3))
        check(lib.mis_se_fc_fwd(csum.data_ptr(), float(S), W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(), ev.N, ev.C, mean.data_ptr(), z1.data_ptr(),
                                a.data_ptr(), stream_ptr()), "mis_se_fc_fwd")
    if mode != SE_CSE:
        bgate = torch.empty(ev.N, S, device=dev)
    p = lambda t: 0 if t is None else t.data_ptr()       # noqa: E731
    check(lib.mis_se_layer_fwd(dtype_code(ev.dtype), ev.ptr, ev.ld, ev.N, S, ev.C, p(a), p(w), p(b0), p(bgate), yv.ptr, yv.ld, mode, stream_ptr()), "mis_se_layer_fwd")
    return mean, z1, a, bgate


def se_layer_bwd(g, e, de, mode, state, W1=None, W2=None, w=None):
    """backward of se_layer_fwd: g = dL/dy -> de = dL/de (may alias g); returns (dW1, db1, dW2, db2, dw, db0) fp32, None where the mode has no such parameter"""
    lib = load()
    gv, ev, dv = _v(g), _v(e), _v(de)
    S = ev.D * ev.H * ev.W
    dev = ev.t.device
    mean, z1, a, bgate = state
    N, C = ev.N, ev.C
    ws = workspace(lib.mis_se_bwd_workspace_bytes(N, C), dev, "se_bwd")
    dt = dtype_code(ev.dtype)
    dq = torch.empty(N, S, device=dev) if mode != SE_CSE else None
    da, dw, db0 = torch.empty(N, C, device=dev), torch.empty(C, device=dev), torch.empty(1, device=dev)
    p = lambda t: 0 if t is None else t.data_ptr()       # noqa: E731
    check(lib.mis_se_layer_bwd_reduce(dt, gv.ptr, gv.ld, ev.ptr, ev.ld, N, S, C, p(a), p(bgate), ws.data_ptr(), p(dq), da.data_ptr(), dw.data_ptr(), db0.data_ptr(),
                                      mode, stream_ptr()), "mis_se_layer_bwd_reduce")
    dW1 = db1 = dW2 = db2 = cross = None
    if mode != SE_SSE:
        dW1, dW2 = torch.empty(C, C, device=dev), torch.empty(C, C, device=dev)
        db1, db2, cross = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(N, C, device=dev)
        check(lib.mis_se_fc_bwd(da.data_ptr(), a.data_ptr(), z1.data_ptr(), mean.data_ptr(), W1.data_ptr(), W2.data_ptr(), N, C, float(S), ws.data_ptr(),
                                dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(), cross.data_ptr(), stream_ptr()), "mis_se_fc_bwd")
    check(lib.mis_se_layer_bwd_apply(dt, gv.ptr, gv.ld, ev.ptr, ev.ld, N, S, C, p(a), p(bgate), p(dq), p(w), p(cross), dv.ptr, dv.ld, mode, 0, stream_ptr()),
          "mis_se_layer_bwd_apply")
    if mode == SE_CSE:
        dw = db0 = None
    return dW1, db1, dW2, db2, dw, db0


def upconv_gather_fwd(z, y, scale, C, bias=None):
    """z (N, h, w, 9*C) dense -> y (N, h*s, w*s, C): the 3x3 taps of conv3x3(bilinear_up_s(x)) gathered from the low-resolution tap products"""
    lib = load()
    y = _v(y)
    N, h, w, c9 = z.shape
    if c9 != 9 * C or not z.is_contiguous() or y.H != h * scale or y.W != w * scale or y.C != C:
        raise MisError(f"upconv_gather_fwd: z {tuple(z.shape)} / y ({y.N},{y.H},{y.W},{y.C}) do not match scale {scale}, C {C}")
    ws = None
    if _os.environ.get("MISAMD_UPCONV_SINGLE_PASS") != "1":
        ws = workspace(lib.mis_upconv_gather_fwd_workspace_bytes(dtype_code(z.dtype), N, h, w, scale, C), z.device, "upconv_v").data_ptr()
    check(lib.mis_upconv_gather_fwd(dtype_code(z.dtype), z.data_ptr(), y.ptr, y.ld, None if bias is None else bias.data_ptr(), N, h, w, scale, C, ws,
                                    stream_ptr()), "mis_upconv_gather_fwd")


def upconv_gather_bwd(dy, dz, scale, C):
    lib = load()
    dy = _v(dy)
    N, h, w, c9 = dz.shape
    if c9 != 9 * C or not dz.is_contiguous() or dy.H != h * scale or dy.W != w * scale or dy.C != C:
        raise MisError(f"upconv_gather_bwd: dz {tuple(dz.shape)} / dy ({dy.N},{dy.H},{dy.W},{dy.C}) do not match scale {scale}, C {C}")
    check(lib.mis_upconv_gather_bwd(dtype_code(dz.dtype), dy.ptr, dy.ld, dz.data_ptr(), N, h, w, scale, C, stream_ptr()), "mis_upconv_gather_bwd")


def add_act(a, b, y, relu=False):
    lib = load()
    a, b, y = _v(a), _v(b), _v(y)
    check(lib.mis_add_act(dtype_code(y.dtype), a.ptr, a.ld, b.ptr, b.ld, y.ptr, y.ld, y.npix, y.C, 1 if relu else 0, stream_ptr()), "mis_add_act")


def expand1_fwd(x, w, b, y):
    """x: fp32 (N, 1, D, H, W) volume; y: NDHWC view; y[v][c] = w[c] * x[v] + b[c]"""
    lib = load()
    y = _v(y)
    check(lib.mis_expand1_fwd(dtype_code(y.dtype), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.ptr, y.ld, y.npix, y.C, stream_ptr()), "mis_expand1_fwd")


def expand1_bwd(x, dy, dw, db):
    lib = load()
    dy = _v(dy)
    ws = workspace(lib.mis_expand1_bwd_workspace_bytes(dy.C), x.device, "expand1")
    check(lib.mis_expand1_bwd(dtype_code(dy.dtype), x.data_ptr(), dy.ptr, dy.ld, dy.npix, dy.C, ws.data_ptr(), dw.data_ptr(), db.data_ptr(), stream_ptr()),
          "mis_expand1_bwd")


# ---- stand-alone / general-shape 3-D blocks (csrc/blocks3d.hip) --------------------------------------------
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_ELU = 0, 1, 2, 3


def norm_act_fwd(x, y, scale, shift, act, slope=0.0):
    """y = act(scale[n, c] * x + shift[n, c]); scale = shift = None: activation only"""
    lib = load()
    x, y = _v(x), _v(y)
    check(lib.mis_norm_act_fwd(dtype_code(x.dtype), x.ptr, x.ld, y.ptr, y.ld, x.N, x.D * x.H * x.W, x.C, None if scale is None else scale.data_ptr(),
                               None if shift is None else shift.data_ptr(), act, slope, stream_ptr()), "mis_norm_act_fwd")


def norm_act_bwd(dy, x, dz, scale, shift, act, slope=0.0):
    lib = load()
    dy, x, dz = _v(dy), _v(x), _v(dz)
    check(lib.mis_norm_act_bwd(dtype_code(x.dtype), dy.ptr, dy.ld, x.ptr, x.ld, dz.ptr, dz.ld, x.N, x.D * x.H * x.W, x.C,
                               None if scale is None else scale.data_ptr(), None if shift is None else shift.data_ptr(), act, slope, stream_ptr()),
          "mis_norm_act_bwd")


def mask_scale(x, mask, y, alpha):
    """y = alpha * x * (mask > 0) (dropout: mask = the Bernoulli draws in the activation dtype)"""
    lib = load()
    x, mask, y = _v(x), _v(mask), _v(y)
    check(lib.mis_mask_scale(dtype_code(x.dtype), x.ptr, x.ld, mask.ptr, mask.ld, y.ptr, y.ld, x.npix, x.C, float(alpha), stream_ptr()), "mis_mask_scale")


def gn_fwd_finalize_ld(s, sq, N, Cc, ld, G, count, gamma, beta, scale, shift, mean, rstd, eps=1e-5):
    lib = load()
    check(lib.mis_gn_fwd_finalize_ld(s.data_ptr(), sq.data_ptr(), N, Cc, ld, G, float(count), gamma.data_ptr(), beta.data_ptr(), eps, scale.data_ptr(),
                                     shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), stream_ptr()), "mis_gn_fwd_finalize_ld")


def gn_bwd_finalize_ld(S1, S2, mean, rstd, gamma, N, Cc, ld, G, count, p, q, r, dgamma, dbeta):
    lib = load()
    check(lib.mis_gn_bwd_finalize_ld(S1.data_ptr(), S2.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), N, Cc, ld, G, float(count),
                                     p.data_ptr(), q.data_ptr(), r.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), stream_ptr()), "mis_gn_bwd_finalize_ld")


def pool3d_fwd(x, y, k, avg=False):
    """x (N, D, H, W, C) -> y (N, D//kd, H//kh, W//kw, C): MaxPool3d / AvgPool3d(kernel = stride = k), floor mode"""
    lib = load()
    x, y = _v(x), _v(y)
    kd, kh, kw = k
    if (y.N, y.D, y.H, y.W) != (x.N, x.D // kd, x.H // kh, x.W // kw) or y.C != x.C:
        raise MisError(f"pool3d_fwd: output grid {(y.N, y.D, y.H, y.W, y.C)} does not match input {(x.N, x.D, x.H, x.W, x.C)} / window {k}")
    check(lib.mis_pool3d_fwd(dtype_code(x.dtype), 1 if avg else 0, kd, kh, kw, x.ptr, x.ld, y.ptr, y.ld, x.N, x.D, x.H, x.W, x.C, stream_ptr()),
          "mis_pool3d_fwd")


def pool3d_bwd(x, dy, dx, k, avg=False):
    lib = load()
    x, dy, dx = _v(x), _v(dy), _v(dx)
    kd, kh, kw = k
    if (dy.N, dy.D, dy.H, dy.W) != (x.N, x.D // kd, x.H // kh, x.W // kw) or (dx.N, dx.D, dx.H, dx.W) != (x.N, x.D, x.H, x.W) or dy.C != x.C or dx.C != x.C:
        raise MisError("pool3d_bwd: grids do not match")
    check(lib.mis_pool3d_bwd(dtype_code(x.dtype), 1 if avg else 0, kd, kh, kw, x.ptr, x.ld, dy.ptr, dy.ld, dx.ptr, dx.ld, x.N, x.D, x.H, x.W, x.C,
                             stream_ptr()), "mis_pool3d_bwd")


def nearest_maps(src, dst, device):
    """torch's `F.interpolate(mode='nearest', size=dst)` index rule (aten UpSample.h nearest_idx: floor(dst_index * (float)src / dst), clamped), per axis:
    (forward maps dst -> src, inverse [lo, hi) ranges src -> dst), int32 device tensors"""
    import numpy as np
    fwd, inv = [], []
    for s, d in zip(src, dst):
        scale = np.float32(s) / np.float32(d)
        m = np.minimum(np.floor(np.arange(d, dtype=np.float32) * scale).astype(np.int64), s - 1)
        if d == s:
            m = np.arange(d)
        r = np.zeros((s, 2), dtype=np.int32)
        for j in range(s):
            hit = np.nonzero(m == j)[0]
            if hit.size:
                r[j] = (hit[0], hit[-1] + 1)
                assert hit[-1] - hit[0] + 1 == hit.size       # monotone map: the readers of a source index are contiguous
        fwd.append(torch.from_numpy(m.astype(np.int32)).to(device))
        inv.append(torch.from_numpy(r.reshape(-1)).to(device))
    return fwd, inv


def gather3d_fwd(x, y, maps):
    """y[n, d, h, w, :] = x[n, mD[d], mH[h], mW[w], :] over the Views' channel slices (equal channel counts)"""
    lib = load()
    x, y = _v(x), _v(y)
    if x.C != y.C or x.N != y.N or [m.numel() for m in maps] != [y.D, y.H, y.W]:
        raise MisError("gather3d_fwd: slices / maps do not match")
    check(lib.mis_gather3d_fwd(dtype_code(x.dtype), x.ptr, x.ld, y.ptr, y.ld, x.N, x.D, x.H, x.W, y.D, y.H, y.W, x.C, maps[0].data_ptr(), maps[1].data_ptr(),
                               maps[2].data_ptr(), stream_ptr()), "mis_gather3d_fwd")


def gather3d_bwd(dy, dx, ranges):
    lib = load()
    dy, dx = _v(dy), _v(dx)
    if dx.C != dy.C or dx.N != dy.N or [r.numel() for r in ranges] != [2 * dx.D, 2 * dx.H, 2 * dx.W]:
        raise MisError("gather3d_bwd: slices / ranges do not match")
    check(lib.mis_gather3d_bwd(dtype_code(dy.dtype), dy.ptr, dy.ld, dx.ptr, dx.ld, dx.N, dx.D, dx.H, dx.W, dy.D, dy.H, dy.W, dx.C, ranges[0].data_ptr(),
                               ranges[1].data_ptr(), ranges[2].data_ptr(), stream_ptr()), "mis_gather3d_bwd")
