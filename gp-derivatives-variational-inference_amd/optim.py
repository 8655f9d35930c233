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
        params = list(params)
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
        self._hp_host = self._hp_dev = None

    # ---- HIP-graph replay of the step (directional_vi.TrainLoop): the update is captured ONCE with (lr, step) read from
    # device memory; per replay the host only refreshes a pinned [groups, 2] table that a captured copy brings over, and
    # keeps ``state[p]["step"]`` in sync so that eager steps, schedulers and checkpoints continue seamlessly ----
    def capture_tables(self, device):
        n = len(self.param_groups)
        if self._hp_host is None or self._hp_host.shape[0] != n:
            self._hp_host = torch.zeros(n, 2, dtype=torch.float32).pin_memory()
            self._hp_dev = torch.zeros(n, 2, dtype=torch.float32, device=device)
        return self._hp_host, self._hp_dev

    def fill_host_table(self):
        """(lr, step number of the NEXT update) of every group into the pinned table"""
        for gi, group in enumerate(self.param_groups):
            steps = [self.state[p]["step"] for p in group["params"] if p in self.state and self.state[p]]
            self._hp_host[gi, 0] = float(group["lr"])
            self._hp_host[gi, 1] = float((steps[0] if steps else 0) + 1)

    @torch.no_grad()
    def step_captured(self, guard=None):
        """the update of ``step`` as launches that read (lr, step) from ``_hp_dev`` (refreshed by the caller's captured
        copy of ``_hp_host``); skipped on the device while ``guard[0] != 0``.  State must exist (one eager step before)."""
        self._hp_dev.copy_(self._hp_host, non_blocking=True)
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            items = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    raise RuntimeError("FusedAdam.step_captured needs initialised state (run one eager step first)")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                items.append((p.data, g, st["exp_avg"], st["exp_avg_sq"]))
            if not items:
                continue
            ctx = _ops.Context.get(items[0][0].device)
            for i in range(0, len(items), _ops.ADAM_MAX_TENSORS):
                ps, gs, ms, vs = zip(*items[i:i + _ops.ADAM_MAX_TENSORS])
                _ops.adam_step_multi_dev_(ctx, ps, gs, ms, vs, self._hp_dev[gi], b1, b2, group["eps"], guard)

    def advance_host_state(self):
        """what ``step`` does to the Python-side state, after a replayed update"""
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st:
                    st["step"] += 1
        self._opt_called = True          # (the LR schedulers check that the optimizer stepped before them)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        batch = {}      # tensors that share device, step count and hyper-parameters go out in ONE launch, whatever group they are in
        self._collect(batch)
        _launch_batches(batch)
        return loss

    def _collect(self, batch):
        """this optimizer's share of an update: advances the per-parameter step counts and appends (param, grad, exp_avg, exp_avg_sq)
        to ``batch[(step, device, lr, beta1, beta2, eps, dtype)]``"""
        for group in self.param_groups:                  # (train_gp's hyper-parameter optimizer has two groups with equal settings)
            b1, b2 = group["betas"]
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
                key = (st["step"], p.device, float(group["lr"]), float(b1), float(b2), float(group["eps"]), p.dtype)
                batch.setdefault(key, []).append((p.data, g, st["exp_avg"], st["exp_avg_sq"], bool(getattr(p, "_dsvgp_tril", False))))


def _launch_batches(batch, guard=None):
    for (step, dev, lr, b1, b2, eps, dt), items in batch.items():
        ctx = _ops.Context.get(dev)
        for i in range(0, len(items), _ops.ADAM_MAX_TENSORS):
            ps, gs, ms, vs, tr = zip(*items[i:i + _ops.ADAM_MAX_TENSORS])
            if len(ps) == 1 and dt == torch.float32 and not tr[0] and guard is None:
                _ops.adam_step_(ctx, ps[0], gs[0], ms[0], vs[0], lr, b1, b2, eps, step)
            else:
                # (float32 or float64 tensors; a lower-triangular parameter -- chol_variational_covar -- is walked below its diagonal only;
                #  ``guard``: the update is skipped on the device while guard[0] != 0)
                _ops.adam_step_multi_(ctx, ps, gs, ms, vs, lr, b1, b2, eps, step, tril=tr, guard=guard)


def can_step_together(optimizers):
    """whether step_together would take these optimizers (all FusedAdam, no step hooks registered)"""
    return bool(optimizers) and all(isinstance(o, FusedAdam) for o in optimizers) and not any(
        getattr(o, "_optimizer_step_pre_hooks", None) or getattr(o, "_optimizer_step_post_hooks", None) for o in optimizers)


@torch.no_grad()
def step_together(optimizers, guard=None):
    """``opt.step()`` of several FusedAdam instances as ONE set of launches: the reference steps two ``torch.optim.Adam`` per iteration
    (variational parameters, then hyper-parameters: directional_vi.py:251-254) whose settings and step counts coincide, so their
    tensors share a multi-tensor launch (one ~4 us launch fewer per step: visible at BASELINE config 2).  The two updates touch
    disjoint parameters and each reads its own learning rate, so the order against the schedulers' steps does not matter.
    Returns False (and does nothing) unless every optimizer is a FusedAdam without registered step hooks.
    ``guard`` (int32 device tensor; round 6): the launches are skipped ON THE DEVICE while guard[0] != 0 -- the status word of the step that
    made the gradients, for a loop that does not wait for that status before it queues the update (TrainLoop's deferred check)."""
    if not optimizers or not all(isinstance(o, FusedAdam) for o in optimizers):
        return False
    # Optimizer.step() is bypassed here: an optimizer that carries step hooks (``register_step_pre_hook`` / ``_post_hook``) keeps the
    # separate ``step()`` calls, which run them
    if any(getattr(o, "_optimizer_step_pre_hooks", None) or getattr(o, "_optimizer_step_post_hooks", None) for o in optimizers):
        return False
    batch = {}
    for o in optimizers:
        o._collect(batch)
        o._opt_called = True             # (what the LR schedulers' step-order check looks at ...
        o._step_count = getattr(o, "_step_count", 0) + 1      # ... and the wrapped step counter older schedulers read)
    if guard is not None and any(k[6] != torch.float32 for k in batch):
        raise ValueError("a guarded update takes float32 parameters")
    _launch_batches(batch, guard)
    return True


def make_adam(param_groups, lr=1e-3):
    """torch.optim.Adam(param_groups, lr) of the reference loop (directional_vi.py:189-198): the fused multi-tensor HIP update, for
    the fp32 model and (round 4: ``dsvgp_adam_step_multi_f64``) for the float64 model mode alike."""
    return FusedAdam(list(param_groups), lr=lr)
