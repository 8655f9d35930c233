"""RBF kernel with directional-derivative blocks -- HIP-backed mirror of the reference plugin
``directionalvi/RBFKernelDirectionalGrad.py`` (same class name, ``forward(x1, x2, diag=False, v1=, v2=)``
signature, ``set_num_directions`` / ``num_outputs_per_input``, same errors).

The output is the dense interleaved ``[n1(p+1), n2(p+1)]`` block matrix of reference :41-108,
assembled by ``dsvgp_kernel_fwd`` (csrc/assemble.hip) directly in that layout; ``diag=True`` follows
:110-119.  No ScaleKernel factor is applied here (outputscale = 1), exactly like the reference class.
"""
import torch

from . import _ops


class _KernelFn(torch.autograd.Function):
    """K(x1, x2; v1, v2) [n1(p+1), n2(p+1)] on the HIP assembly kernels, differentiable w.r.t. both point sets, both
    direction sets and the lengthscale.  ``dsvgp_kernel_bwd`` returns the gradients of side 1; side 2 is the same call on
    G^T with the roles swapped (K(x2, x1; v2, v1) = K(x1, x2; v1, v2)^T)."""

    @staticmethod
    def _packs(x1, x2, v1, v2, ell):
        ctx = _ops.Context.get(x1.device)
        n1, d = x1.shape
        n2 = x2.shape[0]
        p = v1.shape[0] // n1 if n1 else 0
        if x1.dtype == torch.float64:                   # fp64 model mode (exp_script.py:56): csrc/assemble64.hip
            hyp = torch.stack([ell.detach().to(x1), *torch.tensor([1.0, 0.0, 0.0]).to(x1)]).contiguous()
            x1c = x1.detach().contiguous()
            center = x1c.mean(0).contiguous()
            p1 = _ops.pack_points_f64(ctx, x1c, v1.detach().double().contiguous(), p, hyp, center)
            p2 = _ops.pack_points_f64(ctx, x2.detach().double().contiguous(), v2.detach().double().contiguous(), p, hyp, center)
        else:
            hyp = torch.zeros(4, dtype=torch.float32, device=x1.device)
            hyp[0] = ell.detach()
            hyp[1] = 1.0
            x1c = x1.detach().float().contiguous()
            center = _ops.column_mean(ctx, x1c)         # covar_dist's `adjustment = x1.mean(-2)` [gpytorch 1.4.0]
            p1 = _ops.pack_points(ctx, x1c, v1.detach().float().contiguous(), p, hyp, center)
            p2 = _ops.pack_points(ctx, x2.detach().float().contiguous(), v2.detach().float().contiguous(), p, hyp, center)
        return ctx, hyp, p1, p2, n1, n2, d, p

    @staticmethod
    def forward(actx, x1, x2, v1, v2, ell):
        ctx, hyp, p1, p2, n1, n2, d, p = _KernelFn._packs(x1, x2, v1, v2, ell)
        actx.save_for_backward(x1, x2, v1, v2, ell)
        if x1.dtype == torch.float64:
            return _ops.kernel_fwd_f64(ctx, p1, n1, p2, n2, d, p, hyp)
        return _ops.kernel_fwd(ctx, p1, n1, p2, n2, d, p, hyp)

    @staticmethod
    def backward(actx, G):
        x1, x2, v1, v2, ell = actx.saved_tensors
        ctx, hyp, p1, p2, n1, n2, d, p = _KernelFn._packs(x1, x2, v1, v2, ell)
        is64 = x1.dtype == torch.float64
        dt = torch.float64 if is64 else torch.float32
        bwd = _ops.kernel_bwd_f64 if is64 else _ops.kernel_bwd
        need = actx.needs_input_grad
        G = G.to(dt).contiguous()
        z = lambda t: torch.zeros(t.shape, dtype=dt, device=t.device)
        d_x1, d_v1, d_hyp = z(x1), z(v1), torch.zeros(4, dtype=dt, device=x1.device)
        bwd(ctx, G, p1, n1, p2, n2, d, p, hyp, False, d_x1, d_v1, d_hyp)
        d_x2 = d_v2 = None
        if need[1] or need[3]:
            d_x2, d_v2, scratch = z(x2), z(v2), torch.zeros(4, dtype=dt, device=x1.device)
            bwd(ctx, G.t().contiguous(), p2, n2, p1, n1, d, p, hyp, False, d_x2, d_v2, scratch)
        cast = lambda g, t: None if g is None else g.to(t.dtype)
        return cast(d_x1, x1), cast(d_x2, x2), cast(d_v1, v1), cast(d_v2, v2), d_hyp[0].to(ell.dtype).reshape(ell.shape)


class RBFKernelDirectionalGrad(torch.nn.Module):
    def __init__(self):
        super().__init__()
        # gpytorch RBFKernel: raw_lengthscale [1,1], Positive (softplus) constraint, init 0
        self.register_parameter("raw_lengthscale", torch.nn.Parameter(torch.zeros(1, 1)))
        self.n_dir1 = 0

    @property
    def lengthscale(self):
        return torch.nn.functional.softplus(self.raw_lengthscale)

    @lengthscale.setter
    def lengthscale(self, value):
        v = torch.as_tensor(value, dtype=self.raw_lengthscale.dtype, device=self.raw_lengthscale.device)
        with torch.no_grad():   # inverse softplus
            self.raw_lengthscale.copy_((v + torch.log(-torch.expm1(-v))).reshape(1, 1))

    def _hyp(self, device):
        hyp = torch.empty(4, dtype=torch.float32, device=device)
        hyp[0] = self.lengthscale.reshape(()).to(device)
        hyp[1] = 1.0
        hyp[2] = 0.0
        hyp[3] = 0.0
        return hyp

    def forward(self, x1, x2, diag=False, **params):
        n1, d = x1.shape[-2:]
        n2 = x2.shape[-2]
        v1, v2 = params["v1"], params["v2"]
        n_dir1 = int(v1.shape[-2] / n1)
        n_dir2 = int(v2.shape[-2] / n2)
        assert n_dir1 == n_dir2, "v1 and v2 must contain same number of directions"
        self.set_num_directions(n_dir1)
        if not diag:
            # an autograd participant like the reference's forward (:41-108): gradients reach x1, x2, v1, v2 and the
            # lengthscale through dsvgp_kernel_bwd (csrc/assemble.hip / assemble64.hip)
            return _KernelFn.apply(x1, x2, v1, v2, self.lengthscale.reshape(()).to(x1.device))
        if not (n1 == n2 and torch.eq(x1, x2).all() and n_dir1 == n_dir2 and torch.eq(v1, v2).all()):
            raise RuntimeError("diag=True only works when x1 == x2 and v1 == v2")
        if x1.dtype == torch.float64:                   # fp64 model mode (exp_script.py:56)
            ell = self.lengthscale.reshape(()).to(x1)
            row = torch.cat([torch.ones(1).to(x1), (1.0 / ell ** 2).expand(n_dir2)])
            return row.repeat(n2)
        with torch.no_grad():
            return _ops.kernel_diag(_ops.Context.get(x1.device), n2, n_dir2, self._hyp(x1.device))

    def set_num_directions(self, num_directions):
        self.n_dir1 = num_directions

    def num_outputs_per_input(self, x1, x2):
        return self.n_dir1 + 1
