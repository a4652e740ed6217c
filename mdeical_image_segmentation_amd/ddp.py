"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference has no explicit collective (nn.DataParallel in model/unet3d/trainer.py:23-24, or whatever HF
Trainer does under torchrun).  Here every rank runs the fused engine on its shard of the minibatch; gradients are
summed with bucketed all-reduces issued on a second HIP stream as soon as a decoder/encoder stage's wgrad kernels
have been enqueued, so the exchange (124 MB fp32 for the 2-D net, ~0.2-1.4 ms over xGMI) hides behind the remaining
backward kernels.  The 1/world_size factor is folded into the loss gradient (grad_scale), so SUM gives the mean.
Buckets are the contiguous ranges of the flat fp32 gradient buffer that belong to one module.
"""
import ctypes as C

import torch
import torch.distributed as dist


def native_comm_init(group=None):
    """Create this process's RCCL communicator behind the C ABI (mis_comm_init).  The 128-byte unique id is made by rank 0 and carried to the
    other ranks over the existing torch.distributed group (any backend: it is host data).  Call on the rank's own HIP device."""
    from . import _lib
    lib = _lib.load()
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
    uid = (C.c_char * 128)()
    if rank == 0:
        _lib.check(lib.mis_comm_unique_id(uid), "mis_comm_unique_id")
    if world > 1:
        box = [bytes(uid)]
        dist.broadcast_object_list(box, src=0, group=group)
        uid = (C.c_char * 128).from_buffer_copy(box[0])
    _lib.check(lib.mis_comm_init(uid, rank, world), "mis_comm_init")
    return world


def native_comm_finalize():
    from . import _lib
    _lib.check(_lib.load().mis_comm_finalize(), "mis_comm_finalize")


def module_ranges(flat, prefixes):
    """Contiguous [lo, hi) ranges of the flat buffer covered by the weight tensors of the given module prefixes,
    merged when adjacent.  Biases (a few KB in total) are reduced once at the end by GradReducer.finish()."""
    spans = []
    for name, (off, n, _shape) in flat.offsets.items():
        if name.endswith("bias"):
            continue
        if any(name.startswith(p + ".") for p in prefixes):
            spans.append((off, off + (n + 63) // 64 * 64))
    spans.sort()
    merged = []
    for lo, hi in spans:
        if merged and merged[-1][1] == lo:
            merged[-1] = (merged[-1][0], hi)
        else:
            merged.append((lo, hi))
    return merged


class GradReducer:
    """Events are PERSISTENT members (one per bucket slot, created on first use and reused every step): nothing in the ordering of the two
    streams depends on the lifetime of a temporary event object.  timing=True additionally brackets every bucket (on the communication
    stream) and the final join (on the compute stream) with timing events, read back by timing_ms() after a synchronisation:
    all-reduce time = sum of bucket durations, exposed time = how long the compute stream sat in finish() waiting for the last bucket."""

    def __init__(self, flat, group=None, timing=False, backend="torch"):
        """backend "torch": torch.distributed all_reduce (ProcessGroupNCCL = RCCL on ROCm; gloo on CPU); "native": mis_allreduce_bucket on the
        communicator made by native_comm_init() - same collective, no torch in the data path."""
        self.flat = flat
        self.group = group
        self.backend = backend
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # without a process group there is nothing to exchange; WITH one the collectives run even for a single rank (free on one rank, and it lets a one-GPU
        # box execute exactly the code an 8-GPU launch executes: bench.py MISAMD_BENCH_REHEARSAL=nccl1)
        self._skip = backend != "native" and not dist.is_initialized()
        if backend == "native":
            from . import _lib
            self._lib = _lib.load()
            self.world = self._lib.mis_comm_world()
            if self.world < 1:
                raise _lib.MisError("GradReducer(backend='native') needs native_comm_init() first")
        self.on_gpu = flat.g.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=flat.g.device) if self.on_gpu else None
        self.works = []
        self.timing = timing and self.on_gpu
        self._ready = []          # compute -> comm ordering events, one per bucket slot
        self._done = torch.cuda.Event() if self.on_gpu else None
        self._slot = 0
        self.buckets_per_step = 0
        self._t_buckets, self._t_join = [], []
        self._step = 0

    def _reduce(self, t):
        if self._skip:
            return
        if self.on_gpu:
            if self._slot == len(self._ready):
                self._ready.append(torch.cuda.Event())
            ev = self._ready[self._slot]
            self._slot += 1
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                if self.timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.stream)
                if self.backend == "native":
                    from . import _lib
                    _lib.check(self._lib.mis_allreduce_bucket(t.data_ptr(), t.numel(), self.stream.cuda_stream), "mis_allreduce_bucket")
                else:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                if self.timing:
                    e1.record(self.stream)
                    self._t_buckets.append((e0, e1, self._step, self._slot - 1, t.numel()))
        else:
            self.works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def stage_done(self, prefixes):
        for lo, hi in module_ranges(self.flat, prefixes):
            self._reduce(self.flat.g[lo:hi])

    def finish(self):
        """Reduce the bias region and make the compute stream wait for every bucket."""
        if self._skip:
            return
        self._reduce(self.flat.g[self.flat.n_decay:])
        if self.on_gpu:
            self.buckets_per_step = self._slot
            self._slot = 0
            self._done.record(self.stream)
            cur = torch.cuda.current_stream()
            if self.timing:
                j0, j1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                j0.record(cur)
            cur.wait_event(self._done)
            if self.timing:
                j1.record(cur)
                self._t_join.append((j0, j1))
            self._step += 1
        else:
            self.buckets_per_step = len(self.works)
            for w in self.works:
                w.wait()
            self.works = []

    def reset_timing(self):
        self._t_buckets, self._t_join = [], []
        self._step = 0

    def timing_ms(self):
        """(sum of bucket all-reduce durations, sum of compute-stream waits in finish()) in ms since reset_timing(); synchronises"""
        if not self.timing:
            return 0.0, 0.0
        torch.cuda.synchronize()
        ar = sum(b[0].elapsed_time(b[1]) for b in self._t_buckets)
        ex = sum(a.elapsed_time(b) for a, b in self._t_join)
        return ar, ex

    def timing_breakdown(self):
        """per bucket slot (in issue order: head first, biases last), averaged over the steps since reset_timing(): bytes, all-reduce ms, and EXPOSED ms = the part of the
        bucket's interval on the communication stream that lies after the compute stream arrived at finish()'s join (what the backward could not hide); synchronises"""
        if not self.timing or not self._t_join:
            return []
        torch.cuda.synchronize()
        nsteps = len(self._t_join)
        acc = {}
        for e0, e1, step, slot, numel in self._t_buckets:
            j0 = self._t_join[step][0]
            t0, t1 = j0.elapsed_time(e0), j0.elapsed_time(e1)          # bucket start / end relative to the compute stream's arrival at the join (negative: before)
            a = acc.setdefault(slot, [numel * 4, 0.0, 0.0])
            a[1] += e0.elapsed_time(e1)
            a[2] += max(t1, 0.0) - max(t0, 0.0)
        return [{"bucket": slot, "bytes": v[0], "allreduce_ms": round(v[1] / nsteps, 4), "exposed_ms": round(v[2] / nsteps, 4)} for slot, v in sorted(acc.items())]
