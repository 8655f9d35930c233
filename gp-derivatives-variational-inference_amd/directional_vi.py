"""DSVGP training / evaluation harness -- drop-in mirror of the reference
``directionalvi/directional_vi.py`` (``GPModel`` :25-65, ``select_cols_of_y`` :68-90, ``train_gp`` :93-268,
``eval_gp`` :271-305): same names, arguments, defaults, return types and interleaved output ordering.

What is different underneath (MI355X-first):
  * the dataset is made resident in HBM once; each step's minibatch is gathered on the GPU
    (``dsvgp_gather_batch``) from a per-epoch ``randperm`` instead of ``DataLoader``'s per-index collate
    plus a host->device copy per step (directional_vi.py:133,229-232);
  * ``likelihood(model(x))`` + ``mll`` + ``backward`` run as one fused HIP forward/backward (``_step.py``);
  * both Adam optimizers are the fused HIP Adam (``optim.FusedAdam``), stepped with the same
    per-iteration LR schedulers as the reference (:251-254);
  * under ``torch.distributed`` (one process per GPU) every global minibatch is sharded by rows and
    gradients are reduced with one RCCL all-reduce (``parallel.DataParallel``).
``use_ngd=True`` swaps q(u) for a NaturalVariationalDistribution stepped by ``optim.NGD`` (reference :35-37,186-187) on the
same engine; ``use_ciq=True`` additionally whitens with K_ZZ^{-1/2} by contour-integral quadrature + msMINRES
(``CiqDirectionalGradVariationalStrategy``, ``csrc/ciq.hip``).
"""
import os
import random
import sys

import numpy as np
import torch
import torch.distributed as dist

from . import _ops
from .CiqDirectionalGradVariationalStrategy import CiqDirectionalGradVariationalStrategy
from .DirectionalGradVariationalStrategy import DirectionalGradVariationalStrategy
from .RBFKernelDirectionalGrad import RBFKernelDirectionalGrad
from .gp_shim import (ApproximateGP, CholeskyVariationalDistribution, ConstantMean, GaussianLikelihood,
                      NaturalVariationalDistribution, PredictiveLogLikelihood, ScaleKernel, VariationalELBO)
from . import optim as _optim
from .optim import NGD, FusedAdam, make_adam
from .parallel import DataParallel


class GPModel(ApproximateGP):
    def __init__(self, inducing_points, inducing_directions, dim, learn_inducing_locations=True, **kwargs):
        torch.nn.Module.__init__(self)
        self.num_inducing = len(inducing_points)
        self.num_directions = int(len(inducing_directions) / self.num_inducing)  # num directions per point
        num_directional_derivs = self.num_directions * self.num_inducing
        if kwargs.get("variational_distribution") == "NGD":                       # :35-37
            variational_distribution = NaturalVariationalDistribution(self.num_inducing + num_directional_derivs)
        else:
            variational_distribution = CholeskyVariationalDistribution(self.num_inducing + num_directional_derivs)
        self._ciq = kwargs.get("variational_strategy") == "CIQ"
        if self._ciq:                                                             # :46-48
            variational_strategy = CiqDirectionalGradVariationalStrategy(
                self, inducing_points, inducing_directions, variational_distribution,
                learn_inducing_locations=learn_inducing_locations)
        else:
            strategy_class = getattr(self, "_strategy_class", None) or DirectionalGradVariationalStrategy
            variational_strategy = strategy_class(
                self, inducing_points, inducing_directions, variational_distribution,
                learn_inducing_locations=learn_inducing_locations)
        self.variational_strategy = variational_strategy
        self._engine = None
        self.data_parallel = None
        self.mean_module = ConstantMean()
        self.covar_module = ScaleKernel(RBFKernelDirectionalGrad())
        if self._ciq:
            # stable initialization of lengthscale for CIQ (:58-60): lengthscale = 1 / num_inducing through the
            # inverse of the softplus constraint
            ell = torch.tensor(1.0 / self.num_inducing)
            with torch.no_grad():
                self.covar_module.base_kernel.raw_lengthscale.fill_(float(torch.log(torch.expm1(ell))))

    @property
    def engine(self):
        eng = ApproximateGP.engine.fget(self)
        eng.whitening = "ciq" if self._ciq else "cholesky"
        eng.data_outputs = "all"
        eng.shared_directions = False
        return eng

    # --- parameter plumbing for the HIP engine (order = _step.PARAM_NAMES) ---
    def _param_list(self, likelihood=None):
        vs = self.variational_strategy
        vd = vs._variational_distribution
        raw_noise = (likelihood.noise_covar.raw_noise if likelihood is not None
                     else torch.zeros(1, device=vs.inducing_points.device))
        q = ([vd.natural_vec, vd.natural_mat] if isinstance(vd, NaturalVariationalDistribution)
             else [vd.variational_mean, vd.chol_variational_covar])
        return [vs.inducing_points, vs.inducing_directions] + q + [
            self.mean_module.constant, self.covar_module.raw_outputscale,
            self.covar_module.base_kernel.raw_lengthscale, raw_noise]

    def _param_names(self):
        from ._step import NGD_PARAM_NAMES, PARAM_NAMES
        ngd = isinstance(self.variational_strategy._variational_distribution, NaturalVariationalDistribution)
        return NGD_PARAM_NAMES if ngd else PARAM_NAMES

    def _param_dict(self, likelihood=None):
        return {k: _ops.detach_keep(v) for k, v in zip(self._param_names(), self._param_list(likelihood))}


def select_cols_of_y(y_batch, minibatch_dim, dim):
    """
    randomly select columns of y to train on, but always select function values as part of the batch
    (reference directional_vi.py:68-90).  Returns (y_batch[:, idx], canonical derivative directions).
    """
    idx_y = random.sample(range(1, dim + 1), minibatch_dim)  # ensures unique entries
    idx_y += [0]  # append 0 to the list for function values
    idx_y.sort()
    y_batch = y_batch[:, idx_y]
    E_canonical = torch.eye(dim).to(y_batch.device)
    derivative_directions = E_canonical[np.array(idx_y[1:]) - 1]
    return y_batch, derivative_directions


def _dataset_tensors(dataset, device, dtype=None):
    """Materialise a torch Dataset of (x[d], y[d+1]) pairs as two HBM-resident matrices: float32, or float64 when the
    process runs with ``torch.set_default_dtype(torch.float64)`` like the reference's experiment scripts (exp_script.py:56)."""
    if dtype is None:
        dtype = torch.float64 if torch.get_default_dtype() == torch.float64 else torch.float32
    tensors = getattr(dataset, "tensors", None)
    if tensors is not None and len(tensors) == 2:
        X, Y = tensors
    else:
        xs, ys = zip(*[dataset[i] for i in range(len(dataset))])
        X, Y = torch.stack(xs), torch.stack(ys)
    return (X.to(device=device, dtype=dtype).contiguous(),
            Y.to(device=device, dtype=dtype).contiguous())


class TrainLoop:
    """State of one training run: resident dataset, model, the two optimizers / schedulers and the
    per-iteration body of the reference loop (directional_vi.py:229-254).  ``train_gp`` drives it over
    shuffled epochs; ``bench.py`` drives the same ``step`` for its timed region."""

    def __init__(self, X, Y, model, likelihood, mll, optimizers, schedulers, minibatch_dim, dp, col_rng, perm_gen,
                 full_gradient=False, dfree=False):
        self.X, self.Y, self.model, self.likelihood, self.mll = X, Y, model, likelihood, mll
        self.variational_optimizer, self.hyperparameter_optimizer = optimizers
        self.variational_scheduler, self.hyperparameter_scheduler = schedulers
        self.minibatch_dim, self.dp, self.col_rng, self.perm_gen = minibatch_dim, dp, col_rng, perm_gen
        self.autograd_protocol = False          # True: loss = -mll(output, y); loss.backward() through torch.autograd
        self.full_gradient = full_gradient      # grad_svgp: all d+1 target columns, no derivative_directions kwarg
        self.dfree = dfree                      # dfree_directional_vi: function values only, first p canonical directions
        self.device = X.device
        self.dim = X.shape[1]
        self.ctx = _ops.Context.get(self.device)
        self.E_canonical = torch.eye(self.dim, device=self.device, dtype=X.dtype)
        self.graph = None                       # HIP-graph replay of the step: True = on (also DSVGP_GRAPH=1), None / False = eager
        self._graphs, self._graph_seen, self._graph_pending = {}, {}, None
        self._deferred_pending = None       # (idx, idx_y, learning rates) of an eager step whose factorisation status has not been read yet
        self.defer_status = None            # None: DSVGP_DEFER_STATUS (default off); True / False

    def epoch_permutation(self):
        return torch.randperm(self.X.shape[0], device=self.device, generator=self.perm_gen)   # DataLoader(shuffle=True)

    def step(self, idx, need_variance=False):
        """One optimisation step on the GLOBAL minibatch ``idx`` (row indices into the dataset).
        ``need_variance``: True = the caller will read ``output.variance`` of THIS forward, so the per-output path is used
        instead of the ELBO fast path; "values" = only the function-value rows are read (the reference's every-50-steps nll
        print, :255-260: ``output.variance.sqrt()[::num_directions + 1]``) -- the step stays on the fast path and
        ``output.value_variance`` is formed from the A it leaves behind, before the optimizers change the parameters."""
        eng = self.model.engine
        self._values_only = False
        if need_variance == "values":
            ok = (hasattr(eng, "value_variances") and self.mll.mll_type == "ELBO" and eng.whitening == "cholesky"
                  and not eng.shared_directions and not self.dfree and not getattr(self, "plain", False)
                  and hasattr(self.model.variational_strategy._variational_distribution, "chol_variational_covar")
                  and not self.autograd_protocol)
            self._values_only, need_variance = ok, not ok
        eng.elbo_fast = not need_variance
        dim, p, dp = self.dim, self.minibatch_dim, self.dp
        if dp is not None:
            dp.global_batch = idx.shape[0]
            dp.replicated_step = idx.shape[0] < dp.world      # tail smaller than the world: no empty shards (see parallel.py)
            if not dp.replicated_step:
                lo, hi = dp.shard_bounds(idx.shape[0])
                idx = idx[lo:hi]
        # select random columns of y to train on (function values always included), :68-90
        if self.dfree or getattr(self, "plain", False):      # scalar targets: dfree_directional_vi / traditional_vi
            idx_y = [0]
        else:
            idx_y = list(range(dim + 1)) if self.full_gradient else sorted(self.col_rng.sample(range(1, dim + 1), p) + [0])
        if self._graph_eligible(need_variance):
            self._deferred_check_previous()
            return self._graph_step(idx, idx_y)
        if self._graph_pending is not None:
            self._graph_check_previous()
        self._deferred_check_previous()
        return self._eager_step(idx, idx_y, defer=self._defer_eligible(need_variance))

    # ---- deferred status of the one-call step (round 6) -------------------------------------------------------------------------------
    # The eager loop waited ~100 us per step at M' = 600 for the factorisation's status before it queued the optimizers' update.  Now the
    # update is queued at once, guarded ON THE DEVICE by the status word (optim.step_together(guard=)), and the status of step t is read before
    # step t + 1 is queued -- by then it has long arrived.  A failed factorisation (parameters untouched) is redone eagerly through
    # psd_safe_cholesky's jitter ladder, exactly as the graph replay does it (_graph_check_previous).  OPT-IN (``loop.defer_status = True`` /
    # DSVGP_DEFER_STATUS=1): measured at C2 / C3 / C4 on one box, alternating, the step time does not move (0.505-0.528 against 0.505-0.511 /
    # 6.28-6.30 / 11.96-12.01 against 11.95-11.98 ms: the device, not the host, bounds those steps) -- it takes the host's jitter out of the
    # small-problem step, nothing more.
    def _defer_eligible(self, need_variance):
        mode = getattr(self, "defer_status", None)
        if mode is None:
            mode = os.environ.get("DSVGP_DEFER_STATUS", "0") == "1"
        if not mode or need_variance or self.dp is not None or self.autograd_protocol:
            return False
        eng = self.model.engine
        return (eng.whitening == "cholesky" and not eng.shared_directions and self.mll.mll_type == "ELBO" and eng.collective is None
                and getattr(eng, "c_step", True) and not getattr(eng, "deterministic", False) and not getattr(eng, "host_trace", None)
                and _optim.can_step_together([self.variational_optimizer, self.hyperparameter_optimizer])
                and self.X.dtype == torch.float32)

    def _deferred_check_previous(self):
        pend, self._deferred_pending = getattr(self, "_deferred_pending", None), None
        if pend is None:
            return
        if self.model.engine.deferred_check() == 0:
            return
        # K_ZZ + 1e-3 I was not positive definite in fp64: the guarded update did nothing; redo the step with the learning rates and step
        # counts it was issued with (jitter ladder of psd_safe_cholesky, NotPSDError after three tries)
        idx, idx_y, lrs_then = pend
        opts = (self.variational_optimizer, self.hyperparameter_optimizer)
        lrs_now = [[g["lr"] for g in o.param_groups] for o in opts]
        for o, ls in zip(opts, lrs_then):
            for g, lr in zip(o.param_groups, ls):
                g["lr"] = lr
                for prm in g["params"]:
                    if o.state.get(prm):
                        o.state[prm]["step"] -= 1
        try:
            cols = torch.tensor(idx_y, dtype=torch.int32, device=self.device)
            self._device_step(idx.contiguous(), cols, len(idx_y) - 1)
            if not _optim.step_together(list(opts)):
                for o in opts:
                    o.step()
        finally:
            for o, ls in zip(opts, lrs_now):
                for g, lr in zip(o.param_groups, ls):
                    g["lr"] = lr

    def _eager_step(self, idx, idx_y, defer=False):
        dp = self.dp
        eng = self.model.engine
        eng.defer_status = bool(defer)
        # host -> device without a stream sync: pinned staging ring + non_blocking copy (a pageable torch.tensor(...,
        # device=) blocks the host until the stream has drained, i.e. until the previous step has finished)
        slot = self._cols_slot = (getattr(self, "_cols_slot", -1) + 1) % 8
        if not hasattr(self, "_cols_pinned") or self._cols_pinned.shape[1] != len(idx_y):
            self._cols_pinned = torch.empty(8, len(idx_y), dtype=torch.int32).pin_memory()
        self._cols_pinned[slot].copy_(torch.tensor(idx_y, dtype=torch.int32))
        cols = self._cols_pinned[slot].to(self.device, non_blocking=True)
        loss, output, y_batch = self._device_step(idx.contiguous(), cols, len(idx_y) - 1)
        output._value_stride = len(idx_y)
        if self._values_only and getattr(self.model.engine, "_last_fast", None) is not None:
            output._value_varn = self.model.engine.value_variances(self.model._param_dict(self.likelihood))
        eng.defer_status = False
        guard = eng.deferred_guard() if defer else None          # (None: the step took a path that read its status itself)
        if guard is not None:
            lrs = [[g["lr"] for g in o.param_groups] for o in (self.variational_optimizer, self.hyperparameter_optimizer)]
            ok = _optim.step_together([self.variational_optimizer, self.hyperparameter_optimizer], guard=guard)
            assert ok                                            # (_defer_eligible: both are FusedAdam; hooks would have made it False)
            self._deferred_pending = (idx, list(idx_y), lrs)
            self.variational_scheduler.step()
            self.hyperparameter_scheduler.step()
        elif _optim.step_together([self.variational_optimizer, self.hyperparameter_optimizer]):      # both Adam: one multi-tensor launch
            self.variational_scheduler.step()
            self.hyperparameter_scheduler.step()
        else:
            self.variational_optimizer.step()
            self.variational_scheduler.step()
            self.hyperparameter_optimizer.step()
            self.hyperparameter_scheduler.step()
        if dp is not None and dp.check_every > 0:
            self._check_replicas(dp)
        return loss, output, y_batch

    def _check_replicas(self, dp, force=False):
        """DSVGP_DP_CHECK (parallel.DataParallel.check_replicas): every N-th step the replicas compare the parameters nobody reduces
        after the update -- all of them: an Adam step on identical gradients keeps identical replicas identical -- and on a mismatch
        take rank 0's parameters AND rank 0's Adam moments (a replica that drifted once would otherwise drift again)."""
        params = [q for opt in (self.variational_optimizer, self.hyperparameter_optimizer)
                  for g in opt.param_groups for q in g["params"]]
        same = dp.check_replicas([q.data for q in params], force=force)
        if not same:
            import torch.distributed as dist
            for opt in (self.variational_optimizer, self.hyperparameter_optimizer):
                for g in opt.param_groups:
                    for q in g["params"]:
                        st = opt.state.get(q)
                        for k in ("exp_avg", "exp_avg_sq"):
                            if st and k in st:
                                dist.broadcast(st[k], dp.src0, group=dp.group)
        return same

    def _arange_p(self, p):
        t = getattr(self, "_arange_cache", None)
        if t is None or t.numel() != p:
            t = self._arange_cache = torch.arange(p, dtype=torch.int32, device=self.device)
        return t

    def _device_step(self, idx, cols, py):
        """minibatch gather + fused ELBO forward / backward (gradients land in ``.grad``); everything here is stream work"""
        dim, p, dev = self.dim, self.minibatch_dim, self.device
        nb = idx.shape[0]
        self.ctx.bind()                         # the library launches on torch's CURRENT stream (the capture stream under a graph)
        Db = None
        if self.X.dtype == torch.float64:       # fp64 model mode: plain index_select (O(B d) copies)
            x_batch = self.X.index_select(0, idx)
            y_batch = self.Y.index_select(0, idx).index_select(1, cols.long()).reshape(-1)
        else:
            x_batch = torch.empty(nb, dim, dtype=torch.float32, device=dev)
            y_batch = torch.empty(nb * (py + 1), dtype=torch.float32, device=dev)
            fused_dirs = (not self.dfree and not self.full_gradient and py == p and p > 0
                          and self.E_canonical.shape == (dim, dim) and self.E_canonical.dtype == torch.float32)
            Db = torch.empty(nb * p, dim, dtype=torch.float32, device=dev) if fused_dirs else None
            _ops.gather_batch(self.ctx, self.X, self.Y, idx, cols, py, x_batch, y_batch,  # interleaved y, :241
                              self.E_canonical if fused_dirs else None, Db)              # + the directions, :238
        kwargs = {}
        # (the directions are rows of E_canonical, the same p of them for every point: said to the engine as an index list next to
        #  the matrix itself -- _ops.state_directions -- so that K_ZX runs on the canonical-direction assembly kernels)
        if self.dfree:                          # dfree_directional_vi.py:224-227
            kwargs["derivative_directions"] = _ops.state_directions(self.E_canonical[:p].repeat(nb, 1), self._arange_p(p), 0)
        elif not self.full_gradient:
            if Db is not None:
                kwargs["derivative_directions"] = _ops.state_directions(Db, cols[1:], 1)
            else:
                derivative_directions = self.E_canonical.index_select(0, cols[1:].long() - 1)
                kwargs["derivative_directions"] = _ops.state_directions(derivative_directions.repeat(nb, 1), cols[1:], 1)   # :238

        self.variational_optimizer.zero_grad()
        self.hyperparameter_optimizer.zero_grad()
        output = self.likelihood(self.model(x_batch, **kwargs))
        if self.autograd_protocol:
            loss = -self.mll(output, y_batch)           # the reference's three lines, through torch.autograd
            loss.backward()
        else:
            loss = self.mll.backward_step(output, y_batch)      # same numbers, gradients assigned directly
        return loss, output, y_batch

    # ---- HIP-graph replay of the steady-state step (launch-bound regimes: M' of a few hundred, per-rank shards) ----------
    # The whole device side of a step -- gather, assembly, Cholesky chain, solves, ELBO terms, backward, both Adam updates,
    # ~110 launches at M' = 600 -- is captured once per batch shape and replayed with ONE launch.  What varies per step
    # lives in device memory: the row indices and derivative columns (static buffers refreshed before the replay), the
    # learning rates and step counts of the two optimizers (pinned table + captured copy), the noise (the 2 vbar factor is
    # applied by the kernels, _step._elbo_fast).  The potrf status cannot be read inside a graph: the captured Adam kernels
    # are guarded by the status word, and the host looks at the status of step t before it launches step t + 1; a failed
    # factorisation (parameters untouched) is then redone eagerly through psd_safe_cholesky's jitter ladder.
    def _graph_eligible(self, need_variance):
        mode = self.graph
        if mode is None:
            env = os.environ.get("DSVGP_GRAPH")
            mode = None if env is None else env == "1"
        if mode is False or need_variance or self.dp is not None or self.autograd_protocol:
            return False
        eng = self.model.engine
        ok = (eng.whitening == "cholesky" and not eng.shared_directions and self.mll.mll_type == "ELBO"
              and isinstance(self.variational_optimizer, FusedAdam) and isinstance(self.hyperparameter_optimizer, FusedAdam)
              and eng.collective is None)
        if not ok:
            return False
        # opt-in only (``loop.graph = True`` / DSVGP_GRAPH=1 / ``bench.py --graph on``).  Measured on MI355X (round 2): the C2 step
        # (M' = 600, ~75 launches) is bound by its chain of DEPENDENT kernels (ten 24 us Cholesky launches, the fp64 products,
        # ~35 kernels at the few-us floor), which a graph replays at the same kernel-boundary cost: 0.74 ms replayed against
        # 0.77-0.80 ms eager at the end of round 2 (0.82-0.84 against 0.80 before the small products split K).
        return mode is True

    def _graph_step(self, idx, idx_y):
        # graph replay captures the piecewise path (its launches go to the capture stream one by one); the one-call step
        # (dsvgp_elbo_step_f32: its own second stream, a host read of the status word) is the alternative, not a part of it --
        # and the eager warm-up steps below must have created the piecewise path's buffers before the capture
        self.model.engine.c_step = False
        key = (idx.shape[0], len(idx_y))
        gs = self._graphs.get(key)
        if gs is None:
            seen = self._graph_seen[key] = self._graph_seen.get(key, 0) + 1
            if seen <= 2 or len(self._graphs) >= 2:        # eager warm-up (allocations, optimizer state); ragged tails stay eager
                self._graph_check_previous()
                return self._eager_step(idx, idx_y)
            self._graph_check_previous()
            gs = self._graphs[key] = self._capture(idx, idx_y)
        self._graph_check_previous()
        gs.idx.copy_(idx)
        gs.cols_host.copy_(torch.tensor(idx_y, dtype=torch.int32))
        for opt in (self.variational_optimizer, self.hyperparameter_optimizer):
            opt.fill_host_table()
        gs.graph.replay()
        gs.done.record()
        self._graph_pending = (gs, idx, list(idx_y))
        for opt, sch in ((self.variational_optimizer, self.variational_scheduler),
                         (self.hyperparameter_optimizer, self.hyperparameter_scheduler)):
            opt.advance_host_state()
            sch.step()
        return gs.loss, gs.output, gs.y_batch

    def _capture(self, idx, idx_y):
        import types
        dev, eng = self.device, self.model.engine
        gs = types.SimpleNamespace()
        gs.idx = idx.clone().contiguous()
        gs.cols_host = torch.tensor(idx_y, dtype=torch.int32).pin_memory()
        for opt in (self.variational_optimizer, self.hyperparameter_optimizer):
            opt.capture_tables(dev)
            opt.fill_host_table()
        torch.cuda.synchronize(dev)
        gs.graph = torch.cuda.CUDAGraph()
        guard = eng._buf["info"]                            # the potrf status word of the K_ZZ factorisation
        eng.capture_mode = True
        try:
            with torch.cuda.graph(gs.graph):
                cols = gs.cols_host.to(dev, non_blocking=True)
                gs.loss, gs.output, gs.y_batch = self._device_step(gs.idx, cols, len(idx_y) - 1)
                self.variational_optimizer.step_captured(guard)
                self.hyperparameter_optimizer.step_captured(guard)
        finally:
            eng.capture_mode = False
            self.ctx.bind()
        gs.done = torch.cuda.Event()
        gs.host_info = eng._host_info
        return gs

    def _graph_check_previous(self):
        """status of the previously replayed step (its Adam kernels did nothing if the factorisation failed)"""
        pend, self._graph_pending = self._graph_pending, None
        if pend is None:
            return
        gs, idx, idx_y = pend
        gs.done.synchronize()
        if int(gs.host_info[0]) == 0:
            return
        # K_ZZ + 1e-3 I was not positive definite in fp64: the parameters are untouched; redo the step eagerly (jitter
        # ladder of psd_safe_cholesky, NotPSDError after three tries) with the learning rates it was issued with
        opts = (self.variational_optimizer, self.hyperparameter_optimizer)
        lrs = [[g["lr"] for g in o.param_groups] for o in opts]
        for o in opts:
            for gi, g in enumerate(o.param_groups):
                g["lr"] = float(o._hp_host[gi, 0])
                for prm in g["params"]:
                    if o.state.get(prm):
                        o.state[prm]["step"] -= 1
        try:
            cols = torch.tensor(idx_y, dtype=torch.int32, device=self.device)
            self._device_step(idx.contiguous(), cols, len(idx_y) - 1)
            for o in opts:
                o.step()
        finally:
            for o, ls in zip(opts, lrs):
                for g, lr in zip(o.param_groups, ls):
                    g["lr"] = lr

    def finish(self):
        """drain the replay pipeline (the status of the last replayed / deferred step is checked here)"""
        self._graph_check_previous()
        self._deferred_check_previous()


def setup_training(train_dataset, num_inducing=128, num_directions=1, minibatch_size=1, minibatch_dim=1,
                   num_epochs=1, learning_rate_hypers=0.01, inducing_data_initialization=True, lr_sched=None,
                   mll_type="ELBO", gamma=0.1, fixed_inducing_locations=None, seed=None, tensors=None,
                   use_ngd=False, learning_rate_ngd=0.1, use_ciq=False, num_contour_quadrature=15, model_class=None,
                   dfree=False, shared=False):
    """Everything ``train_gp`` does before its loop (directional_vi.py:130-219); returns a TrainLoop."""
    assert num_directions == minibatch_dim
    if not torch.cuda.is_available():
        raise RuntimeError("train_gp needs an MI355X (HIP) device: the DSVGP hot path has no CPU fallback")
    device = torch.device("cuda", torch.cuda.current_device())
    if tensors is not None:          # already-resident (X, Y)
        X, Y = tensors
    else:
        X, Y = _dataset_tensors(train_dataset, device)
    if Y.dim() == 1:                 # derivative-free data: scalar targets
        Y = Y.reshape(-1, 1).contiguous()
    model_class = model_class or GPModel
    n_samples, dim = X.shape
    num_data = (dim + 1) * n_samples                                      # :136

    if inducing_data_initialization is True:
        inducing_points = X[:num_inducing].clone()                        # first M data rows, :140-145
    else:
        inducing_points = torch.rand(num_inducing, dim).to(X)             # :149
    inducing_directions = torch.eye(dim)[:num_directions].repeat(num_inducing, 1).to(X)
    if shared and inducing_data_initialization is not True:          # shared_directional_vi.py:150-155: not tiled
        inducing_directions = torch.eye(dim)[:num_directions].to(X)

    learn_inducing_locations = True
    if fixed_inducing_locations is not None:
        inducing_points = fixed_inducing_locations.to(device)
        learn_inducing_locations = False

    if use_ciq:                                                           # :164-165
        model = model_class(inducing_points, inducing_directions, dim, variational_distribution="NGD",
                            variational_strategy="CIQ", learn_inducing_locations=learn_inducing_locations)
    elif use_ngd:                                                         # :166-167
        model = model_class(inducing_points, inducing_directions, dim, variational_distribution="NGD",
                            learn_inducing_locations=learn_inducing_locations)
    else:
        model = model_class(inducing_points, inducing_directions, dim,
                            learn_inducing_locations=learn_inducing_locations)
    likelihood = GaussianLikelihood()
    model = model.to(X)                      # device and dtype of the data (fp64 model mode: see _step64)
    likelihood = likelihood.to(X)
    model.train()
    likelihood.train()

    dp = None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dp = DataParallel()
        model.data_parallel = dp
    if seed is None and dp is not None:
        seed_t = torch.randint(0, 2 ** 31 - 1, (1,), device=device)
        dist.broadcast(seed_t, 0)
        seed = int(seed_t.item())
    col_rng = random.Random(seed) if seed is not None else random      # reference: global `random`
    perm_gen = torch.Generator(device=device)
    if seed is not None:
        perm_gen.manual_seed(seed)
    else:
        perm_gen.seed()
    if use_ciq:
        model.engine.ciq_num_quadrature = int(num_contour_quadrature)    # gpytorch.settings.num_contour_quadrature, :165
    model.variational_strategy._maybe_init()                              # q(u) <- N(0,I) + 1e-3 randn (first call)
    if dp is not None:   # identical initial state on every rank
        for t in model._param_list(likelihood):
            dist.broadcast(t.data, 0)

    if use_ngd or use_ciq:                                                # :186-187
        variational_optimizer = NGD(list(model.variational_parameters()), num_data=num_data, lr=learning_rate_ngd)
    else:
        variational_optimizer = make_adam([{"params": list(model.variational_parameters())}], lr=learning_rate_hypers)
    hyperparameter_optimizer = make_adam([
        {"params": list(model.hyperparameters())},
        {"params": list(likelihood.parameters())},
    ], lr=learning_rate_hypers)

    if lr_sched == "step_lr":
        num_batches = int(np.ceil(n_samples / minibatch_size))
        milestones = [int(num_epochs * num_batches / 3), int(2 * num_epochs * num_batches / 3)]
        hyperparameter_scheduler = torch.optim.lr_scheduler.MultiStepLR(hyperparameter_optimizer, milestones, gamma=gamma)
        variational_scheduler = torch.optim.lr_scheduler.MultiStepLR(variational_optimizer, milestones, gamma=gamma)
    else:
        if lr_sched is None:
            lr_sched = lambda epoch: 1.0
        hyperparameter_scheduler = torch.optim.lr_scheduler.LambdaLR(hyperparameter_optimizer, lr_lambda=lr_sched)
        variational_scheduler = torch.optim.lr_scheduler.LambdaLR(variational_optimizer, lr_lambda=lr_sched)

    if mll_type == "ELBO":
        mll = VariationalELBO(likelihood, model, num_data=num_data)
    elif mll_type == "PLL":
        mll = PredictiveLogLikelihood(likelihood, model, num_data=num_data)
    else:
        raise ValueError("mll_type must be 'ELBO' or 'PLL'")
    return TrainLoop(X, Y, model, likelihood, mll, (variational_optimizer, hyperparameter_optimizer),
                     (variational_scheduler, hyperparameter_scheduler), minibatch_dim, dp, col_rng, perm_gen, dfree=dfree)


def train_gp(train_dataset, num_inducing=128,
             num_directions=1, minibatch_size=1, minibatch_dim=1, num_epochs=1,
             learning_rate_hypers=0.01, learning_rate_ngd=0.1,
             inducing_data_initialization=True,
             use_ngd=False,
             use_ciq=False,
             lr_sched=None,
             mll_type="ELBO",
             num_contour_quadrature=15,
             watch_model=False, gamma=0.1,
             verbose=True,
             fixed_inducing_locations=None,
             **args):
    """Train a Derivative GP with the Directional Derivative Variational Inference method
    (argument meaning identical to the reference, directional_vi.py:106-129).

    Extra keyword arguments understood through ``**args`` (all optional, ignored by the reference):
      ``seed`` (int): seeds the minibatch permutation and the derivative-column sampling (must be equal
      on all ranks under torch.distributed; broadcast from rank 0 when omitted);
      ``max_steps`` (int): stop after this many optimisation steps.
    """
    assert num_directions == minibatch_dim
    loop = setup_training(train_dataset, num_inducing, num_directions, minibatch_size, minibatch_dim, num_epochs,
                          learning_rate_hypers, inducing_data_initialization, lr_sched, mll_type, gamma,
                          fixed_inducing_locations, seed=args.get("seed"), use_ngd=use_ngd,
                          learning_rate_ngd=learning_rate_ngd, use_ciq=use_ciq,
                          num_contour_quadrature=num_contour_quadrature, model_class=args.get("_model_class"),
                          shared=bool(args.get("_shared")))
    n_samples = loop.X.shape[0]
    max_steps = args.get("max_steps")
    total_step = 0
    loss = None
    for i in range(num_epochs):
        perm = loop.epoch_permutation()
        for start in range(0, n_samples, minibatch_size):
            report = (total_step % 50 == 0) and verbose
            loss, output, y_batch = loop.step(perm[start:start + minibatch_size], need_variance="values" if report else False)
            if report:
                means = output.mean[::num_directions + 1]
                stds = output.value_variance.sqrt()          # = output.variance.sqrt()[::num_directions + 1]
                nll = -torch.distributions.Normal(means, stds).log_prob(y_batch[::num_directions + 1]).mean()
                print(f"Epoch: {i}; total_step: {total_step}, loss: {loss.item()}, nll: {nll}")
                sys.stdout.flush()
            total_step += 1
            if max_steps is not None and total_step >= max_steps:
                break
        if max_steps is not None and total_step >= max_steps:
            break

    loop.finish()
    if verbose and loss is not None:
        print(f"Done! loss: {loss.item()}")
        print("\nDone Training!")
    return loop.model, loop.likelihood


def eval_gp(test_dataset, model, likelihood,
            mll_type="ELBO", num_directions=1, minibatch_size=1, minibatch_dim=1):
    """Predictive means / variances (with likelihood noise) of all (p+1) outputs per test point,
    returned as CPU vectors of length N_test*(p+1), interleaved (reference directional_vi.py:271-305)."""
    assert num_directions == minibatch_dim
    dim = len(test_dataset[0][0])
    device = model.variational_strategy.inducing_points.device
    X, _ = _dataset_tensors(test_dataset, device, model.variational_strategy.inducing_points.dtype)
    n_test = X.shape[0]

    model.eval()
    likelihood.eval()

    means = []
    variances = []
    with torch.no_grad():
        for start in range(0, n_test, minibatch_size):
            x_batch = X[start:start + minibatch_size]
            # redo derivative directions b/c batch size is not consistent
            derivative_directions = torch.eye(dim, device=device)[:num_directions]
            derivative_directions = derivative_directions.repeat(len(x_batch), 1)
            if derivative_directions.is_cuda:   # eye(d)[:p] tiled: one-hot and shared (canonical-direction assembly)
                _ops.state_directions(derivative_directions, torch.arange(num_directions, dtype=torch.int32, device=device), 0)
            preds = likelihood(model(x_batch, derivative_directions=derivative_directions))
            means.append(preds.mean.cpu())
            variances.append(preds.variance.cpu())
    means = torch.cat(means) if means else torch.zeros(0)
    variances = torch.cat(variances) if variances else torch.zeros(0)
    print("Done Testing!")
    return means, variances
