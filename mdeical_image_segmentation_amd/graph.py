"""hipGraph capture of a fused train step.

The fused engines issue ~150 kernel launches per step from Python.  At the benchmark shapes (bs 32 x 512², 2 x 128³) the GPU queue never drains, so
launch cost is hidden; at small batches / images it is the bound.  `GraphedTrainStep` captures ONE train step - forward, loss, backward, gradient
clip, AdamW, operand repack - into a hipGraph (torch.cuda.CUDAGraph drives hipStreamBeginCapture on the stream every launcher of the C ABI already
enqueues on) and replays it with a single launch.  What makes the step capturable: all activations / gradients / workspaces are pre-allocated by the
engine, nothing synchronises, and the step-dependent optimizer scalars live in device memory (`mis_adamw_step_dev`: the step counter is advanced BY
the graph, the learning rate is a device scalar the caller may overwrite between replays)."""
import torch

from ._lib import MisError


class GraphedTrainStep:
    def __init__(self, eng, inputs, targets, warmup=2):
        if not inputs.is_cuda:
            raise MisError("GraphedTrainStep: CUDA tensors only")
        self.eng = eng
        self.inputs, self.targets = inputs.clone(), targets.clone()
        f = eng.flat
        keep = (f.p.clone(), f.m.clone(), f.v.clone(), eng.step_count)         # warm-up steps must not train
        side, eng.side_reduce = getattr(eng, "side_reduce", False), False      # one stream inside the capture
        self._side = side
        cur = torch.cuda.current_stream(inputs.device)
        s = torch.cuda.Stream(device=inputs.device)
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(max(1, warmup)):                                    # allocates every buffer / workspace, sets the kernel attributes
                self._step()
        cur.wait_stream(s)
        torch.cuda.synchronize(inputs.device)
        self._restore(keep)
        # the GroupNorm-backward route of every layer (statistics from the weight gradient, or the direct pass: engine3d.refresh_gn_flags) is baked into the captured
        # launches: settle the flags now, remember the routes, and refuse a replay once the parameters have drifted across the threshold
        if hasattr(eng, "refresh_gn_flags"):
            eng.refresh_gn_flags(sync=True)
        self._routes = eng.gn_routes() if hasattr(eng, "gn_routes") and getattr(eng, "gn_from_dw", False) else None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._step()
        self._restore(keep)                                                    # capture does not execute, but keep the state explicit

    def _restore(self, keep):
        eng, f = self.eng, self.eng.flat
        f.p.copy_(keep[0])
        f.m.copy_(keep[1])
        f.v.copy_(keep[2])
        eng.step_count = keep[3]
        eng.opt_step.fill_(keep[3])
        eng.repack()

    def _step(self):
        eng = self.eng
        eng.forward(self.inputs, self.targets, train=True)
        eng.backward()
        eng.optimizer_step_dev()

    def __call__(self, inputs=None, targets=None, lr=None):
        """replays the captured step (optionally on new data / a new learning rate); returns the device loss scalar"""
        if inputs is not None:
            self.inputs.copy_(inputs)
        if targets is not None:
            self.targets.copy_(targets)
        if lr is not None:
            self.eng.lr_dev.fill_(float(lr))
        if self._routes is not None:
            # (outside the capture: one small kernel + an asynchronous read-back per replay; the check uses the newest flags that have arrived)
            self.eng.refresh_gn_flags()
            self.eng._poll_gn_flags()
            if self.eng.gn_routes() != self._routes:
                raise MisError("GraphedTrainStep: a GroupNorm layer's |gamma| / |beta| ratio crossed the conditioning threshold (engine3d.GN_COND_RATIO): the captured step "
                               "holds the other backward route for it - capture a new GraphedTrainStep")
        self.graph.replay()
        self.eng.step_count += 1
        return self.eng.loss_buf[:1]

    def release(self):
        self.eng.side_reduce = self._side
