"""A whole training step as ONE HIP graph.

The CycleGAN step at 128^3 is ~2 500 kernel launches; the host (autograd + ctypes, two threads) needs ~60 ms to enqueue what the GPU executes in ~67 ms,
and through the discriminator phase — hundreds of few-microsecond kernels — the GPU waits for the host (8 ms of idle time per step, tools/gap_analysis.py).
raygun trains on fixed-size crops (train_config: input_shape / output_shape), so the launch sequence of a step is the same every iteration: it is recorded
once (torch.cuda.graph = hipStreamBeginCapture ... hipGraphInstantiate) and replayed with one hipGraphLaunch per step.

What capture needs from the step (and what GraphedStep checks or arranges):
  * every tensor the step reads is written IN PLACE between replays (`GraphedStep.inputs`: copy_ the next batch into them);
  * no host read inside the step (no .item() / float(loss): LinkCycleLoss keeps its loss_dict entries as 0-dim tensors);
  * optimisers whose step-dependent scalars live on the device: FusedAdam(capturable=True) (rg_adam_step_multi_hyp);
  * kernels whose launch arguments do not encode "which launch this is": igemm5's stream-K flags are lowered by their reader, not compared with an epoch;
  * everything the library allocates for itself (stream-K flag rows, kernel attributes) exists before the capture: `warmup` eager steps come first.
Data-parallel training stays eager (RCCL collectives are capturable, but the hooks' host-side bookkeeping is not replayed)."""
import os

import torch

from . import _debug


class GraphedStep:
    """fn() -> tensor or tuple of tensors: one complete step (zero_grad, forward, loss, backward(s), optimizer step(s)) on static inputs.
    Calls 1..warmup run eagerly on a side stream, call warmup+1 is captured (and is a real step: capture executes nothing, so the graph is replayed at
    once), later calls replay.  `optimizers`: the FusedAdam instances stepped inside fn (their host-side step counts follow the replays)."""

    def __init__(self, fn, optimizers=(), warmup=3):
        self.fn, self.optimizers, self.warmup = fn, list(optimizers), int(warmup)
        self.graph, self.out, self.calls = None, None, 0
        self.stream = None
        # the diagnostic modes wait for / copy back after every library call (guard bands, named synchronisation): nothing of that can sit in a capture
        self.eager_only = bool(_debug.ON) or bool(os.environ.get("RAYGUN_AMD_DEBUG_SYNC"))
        for o in self.optimizers:
            if not getattr(o, "capturable", False):
                raise RuntimeError("raygun_amd.GraphedStep: optimisers stepped inside a captured step must be FusedAdam(capturable=True)")

    def __call__(self):
        self.calls += 1
        if self.graph is not None:
            for o in self.optimizers:
                o.sync_hyper()
            self.graph.replay()
            for o in self.optimizers:
                o.replayed()
            return self.out
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        if self.calls <= self.warmup or self.eager_only:
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                out = self.fn()
            cur.wait_stream(self.stream)
            return out
        for o in self.optimizers:
            o.sync_hyper()
        g = torch.cuda.CUDAGraph()
        self.stream.wait_stream(cur)
        with torch.cuda.graph(g, stream=self.stream):
            self.out = self.fn()
        cur.wait_stream(self.stream)
        self.graph = g
        g.replay()          # the captured call is a step like any other (its host-side bookkeeping — step counts, versions — ran during the capture)
        return self.out
