"""RBF kernel with directional-derivative blocks -- HIP-backed mirror of the reference plugin
``directionalvi/RBFKernelDirectionalGrad.py`` (same class name, ``forward(x1, x2, diag=False, v1=, v2=)``
signature, ``set_num_directions`` / ``num_outputs_per_input``, same errors).

The output is the dense interleaved ``[n1(p+1), n2(p+1)]`` block matrix of reference :41-108,
assembled by ``dsvgp_kernel_fwd`` (csrc/assemble.hip) directly in that layout; ``diag=True`` follows
:110-119.  No ScaleKernel factor is applied here (outputscale = 1), exactly like the reference class.
"""
import torch

from . import _ops


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

    @torch.no_grad()
    def forward(self, x1, x2, diag=False, **params):
        n1, d = x1.shape[-2:]
        n2 = x2.shape[-2]
        v1, v2 = params["v1"], params["v2"]
        n_dir1 = int(v1.shape[-2] / n1)
        n_dir2 = int(v2.shape[-2] / n2)
        assert n_dir1 == n_dir2, "v1 and v2 must contain same number of directions"
        self.set_num_directions(n_dir1)
        ctx = _ops.Context.get(x1.device)
        if x1.dtype == torch.float64:                   # fp64 model mode (exp_script.py:56): csrc/assemble64.hip
            hyp = torch.stack([self.lengthscale.reshape(()).to(x1), *torch.tensor([1.0, 0.0, 0.0]).to(x1)]).contiguous()
            if not diag:
                x1c = x1.contiguous()
                center = x1c.mean(0).contiguous()
                p1 = _ops.pack_points_f64(ctx, x1c, v1.double().contiguous(), n_dir1, hyp, center)
                p2 = _ops.pack_points_f64(ctx, x2.double().contiguous(), v2.double().contiguous(), n_dir2, hyp, center)
                return _ops.kernel_fwd_f64(ctx, p1, n1, p2, n2, d, n_dir1, hyp)
            if not (n1 == n2 and torch.eq(x1, x2).all() and torch.eq(v1, v2).all()):
                raise RuntimeError("diag=True only works when x1 == x2 and v1 == v2")
            row = torch.cat([torch.ones(1).to(x1), (1.0 / hyp[0] ** 2).expand(n_dir2)])
            return row.repeat(n2)
        hyp = self._hyp(x1.device)
        if not diag:
            x1c = x1.float().contiguous()
            center = _ops.column_mean(ctx, x1c)         # covar_dist's `adjustment = x1.mean(-2)` [gpytorch 1.4.0]
            p1 = _ops.pack_points(ctx, x1c, v1.float().contiguous(), n_dir1, hyp, center)
            p2 = _ops.pack_points(ctx, x2.float().contiguous(), v2.float().contiguous(), n_dir2, hyp, center)
            return _ops.kernel_fwd(ctx, p1, n1, p2, n2, d, n_dir1, hyp)
        if not (n1 == n2 and torch.eq(x1, x2).all() and n_dir1 == n_dir2 and torch.eq(v1, v2).all()):
            raise RuntimeError("diag=True only works when x1 == x2 and v1 == v2")
        return _ops.kernel_diag(ctx, n2, n_dir2, hyp)

    def set_num_directions(self, num_directions):
        self.n_dir1 = num_directions

    def num_outputs_per_input(self, x1, x2):
        return self.n_dir1 + 1
