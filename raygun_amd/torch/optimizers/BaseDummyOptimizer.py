"""Holder of the real optimisers whose step() only advances LR schedulers — drop-in for
raygun.torch.optimizers.BaseDummyOptimizer (reference: optimizers/BaseDummyOptimizer.py:5-47)."""
import torch


class BaseDummyOptimizer(torch.nn.Module):
    def __init__(self, scheduler=None, scheduler_kwargs={}, **optimizers):
        super().__init__()
        self.schedulers = {}
        for name, optimizer in optimizers.items():
            setattr(self, name, optimizer)
            if isinstance(scheduler, str):
                if scheduler == "LambdaLR":
                    # constant LR for n_epochs, then linear decay to zero over n_epochs_decay (:24-31)
                    def lambda_rule(epoch):
                        return 1.0 - max(0, epoch + scheduler_kwargs["epoch_count"] - scheduler_kwargs["n_epochs"]) / (
                            scheduler_kwargs["n_epochs_decay"] + 1.0)

                    self.schedulers[name] = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
                else:
                    self.schedulers[name] = getattr(torch.optim.lr_scheduler, scheduler)(optimizer, **scheduler_kwargs)
            elif scheduler is not None:
                self.schedulers[name] = scheduler(optimizer, **scheduler_kwargs)

    def step(self):
        for name, scheduler in self.schedulers.items():
            scheduler.step()
