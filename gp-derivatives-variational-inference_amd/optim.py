"""Fused Adam on the HIP kernel ``dsvgp_adam_step`` with torch.optim.Adam's update rule
(the reference steps two ``torch.optim.Adam`` instances per iteration, directional_vi.py:193-199,251-254).
Being a ``torch.optim.Optimizer`` it works with the reference's MultiStepLR / LambdaLR schedulers."""
import torch

from . import _ops


class NGD(torch.optim.Optimizer):
    """gpytorch.optim.NGD (1.4.0): ``theta <- theta - lr * num_data * grad`` on the natural parameters of a
    NaturalVariationalDistribution, whose ``.grad`` holds the expectation-parameter gradients
    (reference directional_vi.py:186-187)."""

    def __init__(self, params, num_data, lr=0.1):
        self.num_data = num_data
        super().__init__(params, defaults=dict(lr=lr))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                p.add_(p.grad, alpha=(-group["lr"] * self.num_data))
        return None


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            batch = {}                                   # tensors that share the step count go out in one launch
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                batch.setdefault((st["step"], p.device), []).append((p.data, g, st["exp_avg"], st["exp_avg_sq"]))
            for (step, dev), items in batch.items():
                ctx = _ops.Context.get(dev)
                for i in range(0, len(items), _ops.ADAM_MAX_TENSORS):
                    ps, gs, ms, vs = zip(*items[i:i + _ops.ADAM_MAX_TENSORS])
                    if len(ps) == 1:
                        _ops.adam_step_(ctx, ps[0], gs[0], ms[0], vs[0], group["lr"], b1, b2, group["eps"], step)
                    else:
                        _ops.adam_step_multi_(ctx, ps, gs, ms, vs, group["lr"], b1, b2, group["eps"], step)
        return loss
