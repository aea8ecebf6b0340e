"""torch.optim.Adam semantics on the fused HIP kernel (rg_adam_step_multi): one launch per 24 parameter tensors, state keys
(step / exp_avg / exp_avg_sq) identical to torch's so optimizer state_dicts interchange.
Reference call site: optimizers/utils.py:4-9 -> torch.optim.Adam(params, **optim_kwargs) (systems/CycleGAN.py:167-184)."""
import torch

from ... import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("raygun_amd.FusedAdam: amsgrad is not built")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.grad_scale = 1.0   # gradients are multiplied by this inside the kernel (amp.LossScaler sets 1 / loss scale around step())

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            batches = {}      # step count -> [(p, g, exp_avg, exp_avg_sq)]: one multi-tensor launch per 24 tensors that share the hyper-parameters
            for p in group["params"]:
                if p.grad is None:
                    continue
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
                batches.setdefault(int(st["step"]), []).append((p, g, st["exp_avg"], st["exp_avg_sq"]))
            for step, items in batches.items():
                fwds = ops.adam_step_multi_(items, float(group["lr"]), b1, b2, group["eps"], group["weight_decay"], step, float(self.grad_scale))
                for (p, _, _, _), fwd in zip(items, fwds):
                    # the kernel wrote through the raw pointer: tell autograd (packed-weight caches key on _version)
                    torch.autograd.graph.increment_version(p)
                    if fwd is not None:  # the forward operand was refreshed in the same pass: it is current for the new version
                        p._rg_pack_cache = {"ver": (p._version, p.data_ptr()), fwd[2]: fwd[0], ("fwd_pack" if len(fwd[2]) == 4 else "fwd_pack_f16"): fwd}
        return loss
