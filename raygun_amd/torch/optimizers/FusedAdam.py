"""torch.optim.Adam semantics on the fused HIP kernel (rg_adam_step_multi): one launch per 24 parameter tensors, state keys
(step / exp_avg / exp_avg_sq) identical to torch's so optimizer state_dicts interchange.
Reference call site: optimizers/utils.py:4-9 -> torch.optim.Adam(params, **optim_kwargs) (systems/CycleGAN.py:167-184).

overlap_backward(): the update of a parameter is enqueued — on a second HIP stream — as soon as loss.backward() has accumulated its gradient, so the
HBM-bound Adam pass (config U: 259 M parameters x 30 bytes = 1.4 ms) runs UNDER the MFMA-bound kernels of the rest of the backward pass instead of
after them; step() then only waits for that stream.  Same kernel, same arithmetic, same state: results are bit-identical to the plain order.
Measured on MI355X (config U, 196^3): 22.38 ms per step against 22.19 in the plain order — the convolution kernels run 2 waves x 256 registers per
SIMD, the whole register file, so nothing of a second stream becomes resident beside them and the update only moves between kernel boundaries.
Kept as an option (RAYGUN_AMD_ADAM_OVERLAP=1 in the train driver) for networks whose backward pass is not made of such kernels."""
import torch

from ... import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, capturable=False):
        if amsgrad:
            raise NotImplementedError("raygun_amd.FusedAdam: amsgrad is not built")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.grad_scale = 1.0   # gradients are multiplied by this inside the kernel (amp.LossScaler sets 1 / loss scale around step())
        self._ov = None         # overlap_backward() state
        # capturable (torch.optim.Adam's word for it): the step-dependent scalars — lr, 1 - beta1^t, sqrt(1 - beta2^t) — live in DEVICE memory and are advanced
        # by device ops inside step(), so a step() captured in a HIP graph (raygun_amd/graphs.py) stays right on every replay.  All parameters of a group
        # then share ONE step count (torch's capturable Adam keeps one per parameter; here it is one per group, the multi-tensor launch's granularity).
        self.capturable = bool(capturable)
        self._cap = {}          # id(group) -> {"t": float64[1] step count, "hyp": float32[3], "lr": the lr last written}

    # ------------------------------------------------------------------------------------------------ device-side step scalars (capturable)
    def _cap_state(self, group, device):
        c = self._cap.get(id(group))
        if c is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("raygun_amd.FusedAdam(capturable=True): run at least one eager step before capturing (the device-side step state is "
                                   "allocated on first use)")
            t0 = max([float(self.state[p]["step"]) for p in group["params"] if p in self.state and len(self.state[p])] or [0.0])
            c = {"t": torch.full((1,), t0, dtype=torch.float64, device=device), "hyp": torch.zeros(3, dtype=torch.float32, device=device), "lr": None,
                 "b": torch.tensor(group["betas"], dtype=torch.float64, device=device)}
            self._cap[id(group)] = c
        return c

    def _cap_advance(self, group, device):
        """t += 1 and the two bias corrections, on the device (captured with the step when a graph is being recorded)."""
        c = self._cap_state(group, device)
        if c["lr"] != float(group["lr"]):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("raygun_amd.FusedAdam: the learning rate changed while a graph was being captured; call sync_hyper() before the capture")
            c["hyp"][0:1].fill_(float(group["lr"])); c["lr"] = float(group["lr"])
        c["t"].add_(1.0)
        bc = 1.0 - torch.pow(c["b"], c["t"])           # float64: [1 - b1^t, 1 - b2^t]
        c["hyp"][1:2].copy_(bc[0:1]); c["hyp"][2:3].copy_(torch.sqrt(bc[1:2]))
        return c["hyp"]

    def sync_hyper(self):
        """Host -> device: learning rates a scheduler changed since the last step (call between graph replays; graphs.GraphedStep does)."""
        for group in self.param_groups:
            c = self._cap.get(id(group))
            if c is not None and c["lr"] != float(group["lr"]):
                c["hyp"][0:1].fill_(float(group["lr"])); c["lr"] = float(group["lr"])

    def replayed(self):
        """A captured step() has run once more on the device: advance the host-side step counts (state_dict fidelity) and the parameters' versions —
        operands cached per version OTHER than the forward pack (backward-data packs, Winograd transforms) were computed inside the graph from the
        weights before its update, so an eager call after the replay must not take them for current."""
        for group in self.param_groups:
            if id(group) in self._cap:
                for p in group["params"]:
                    st = self.state.get(p)
                    if st is not None and len(st):
                        st["step"] += 1
                        torch.autograd.graph.increment_version(p)

    # ------------------------------------------------------------------------------------------------ one parameter -> one work item
    def _item(self, p):
        """(p, g, exp_avg, exp_avg_sq) with the step count advanced — the checks and layout fixes of one parameter's update."""
        if not p.is_cuda:
            raise RuntimeError("raygun_amd.FusedAdam: parameters must live on the GPU (no CPU fallback)")
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        st["step"] += 1
        g = p.grad
        # the kernel walks the STORAGE: any dense layout works (contiguous or channels-last conv weights) as long as
        # parameter, gradient and both moments share it
        if p.dtype != torch.float32 or not ops.is_dense(p):
            raise RuntimeError("raygun_amd.FusedAdam: fp32 dense master parameters expected")
        if g.dtype != torch.float32 or g.stride() != p.stride():
            g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
        for k in ("exp_avg", "exp_avg_sq"):
            if st[k].stride() != p.stride():  # e.g. a state dict loaded from a contiguous checkpoint
                st[k] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(st[k])
        return (p, g, st["exp_avg"], st["exp_avg_sq"]), int(st["step"])

    def _launch(self, group, step, items):
        b1, b2 = group["betas"]
        hyp = self._cap_advance(group, items[0][0].device) if self.capturable else None
        fwds = ops.adam_step_multi_(items, float(group["lr"]), b1, b2, group["eps"], group["weight_decay"], step, float(self.grad_scale), hyp)
        for (p, _, _, _), fwd in zip(items, fwds):
            # the kernel wrote through the raw pointer: tell autograd (packed-weight caches key on _version)
            torch.autograd.graph.increment_version(p)
            if fwd is not None:  # the forward operand was refreshed in the same pass: it is current for the new version
                p._rg_pack_cache = {"ver": (p._version, p.data_ptr()), fwd[2]: fwd[0], ("fwd_pack" if len(fwd[2]) == 4 else "fwd_pack_f16"): fwd}

    # ------------------------------------------------------------------------------------------------ the update under the backward pass
    def overlap_backward(self, enable=True, flush_numel=1 << 22):
        """Enqueue every parameter's update from its post-accumulate-grad hook, on a side stream (small tensors are collected until `flush_numel`
        elements are waiting: one multi-tensor launch instead of one per bias).  The caller's contract — the plain training step of the reference
        (BaseTrain.py:42-60) keeps it: ONE backward() per step(), nothing reads or rewrites the gradients between the two (no clipping, no loss-scale
        check: grad_scale must stay 1), no gradient averaging across ranks after backward.  step() is then the join."""
        if not enable:
            if self._ov is not None:
                for h in self._ov["handles"]:
                    h.remove()
                self._ov = None
            return self
        if self._ov is not None:
            return self
        ov = dict(stream=None, pending=[], numel=0, done=set(), handles=[], flush_numel=int(flush_numel), group_of={})
        for group in self.param_groups:
            for p in group["params"]:
                ov["group_of"][p] = group
                if p.requires_grad:
                    ov["handles"].append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._ov = ov
        return self

    @torch.no_grad()
    def _on_grad(self, p):
        ov = self._ov
        if ov is None or p.grad is None or p in ov["done"]:
            return
        if self.grad_scale != 1.0:
            raise RuntimeError("raygun_amd.FusedAdam: overlap_backward() cannot run under a loss scaler (the inf check must precede every update)")
        item, step = self._item(p)
        ov["pending"].append((ov["group_of"][p], step, item))
        ov["numel"] += p.numel()
        ov["done"].add(p)
        if ov["numel"] >= ov["flush_numel"]:
            self._flush()

    @torch.no_grad()
    def _flush(self):
        ov = self._ov
        if not ov["pending"]:
            return
        if ov["stream"] is None:
            ov["stream"] = torch.cuda.Stream()
        side = ov["stream"]
        side.wait_stream(torch.cuda.current_stream())     # the gradients (and the last reads of the old weights) are ahead of us on the main stream
        batches = {}
        for group, step, item in ov["pending"]:
            batches.setdefault((id(group), step), (group, step, []))[2].append(item)
        with torch.cuda.stream(side):
            for group, step, items in batches.values():
                self._launch(group, step, items)
        ov["pending"], ov["numel"] = [], 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ov = self._ov
        for group in self.param_groups:
            if self.capturable and id(group) not in self._cap:      # (before this step's counts advance: the device count starts where the host's stands)
                dev = next((p.device for p in group["params"] if p.is_cuda), None)
                if dev is not None:
                    self._cap_state(group, dev)
            batches = {}      # step count -> [(p, g, exp_avg, exp_avg_sq)]: one multi-tensor launch per 24 tensors that share the hyper-parameters
            for p in group["params"]:
                if p.grad is None or (ov is not None and p in ov["done"]):
                    continue
                item, step = self._item(p)
                batches.setdefault(0 if self.capturable else step, []).append(item)   # capturable: one step count per group (on the device)
            for step, items in batches.items():
                self._launch(group, step, items)
        if ov is not None:
            self._flush()
            if ov["stream"] is not None:
                torch.cuda.current_stream().wait_stream(ov["stream"])
            ov["done"].clear()
        return loss
