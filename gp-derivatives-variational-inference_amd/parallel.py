"""Data-parallel minibatch sharding for the DSVGP step: one process per GPU, RCCL over xGMI.

The reference is single-process (no collective anywhere); this layer is new.
The ELBO log-likelihood is a sum over minibatch rows, so rank g takes rows [g*B/G, (g+1)*B/G) of every
global minibatch, computes partial gradients normalised by the GLOBAL row count, and an all-reduce(sum)
of a flat fp32 buffer [grads..., loss] makes them identical on every rank -- issued in two pieces: the
variational-parameter gradients (36 MB at M'=3000) as soon as they are final, overlapped with the rest of
the backward, and the small remainder at the end of the step.  K_ZZ, its
Cholesky factor and the KL term are replicated; the KL term is added on rank 0 only so that the sum
counts it once.  The Cholesky backward is linear in its upstream gradient, so reducing the final
parameter gradients (not L-bar) is exact.
"""
import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, group=None):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.global_batch = None      # set by the training loop before every step
        # collective algorithm of the large (early) operand: "allreduce" (RCCL picks ring / tree) or "rs_ag"
        # (reduce-scatter + all-gather); DSVGP_DP_ALGO overrides for A/B runs on an 8-GPU node
        import os
        self.algo = os.environ.get("DSVGP_DP_ALGO", "allreduce")
        self.rs_ag_min_numel = 1 << 16
        self._shards = {}
        # rank 0 OF THE GROUP as a global rank: what dist.broadcast's ``src`` means (a sub-group need not contain global rank 0)
        self.src0 = dist.get_global_rank(group, 0) if group is not None else 0
        self.replicated_step = False  # set by the training loop for a tail minibatch with fewer rows than ranks
        # Replica consistency of L_S, m (and their Adam moments) rests on the fixed-order G L_S product and on bitwise-identical
        # all-reduce results; nothing re-broadcasts them.  DSVGP_DP_CHECK=N: every N-th step the replicas compare a checksum of
        # their parameters' bit patterns (one small all-reduce of max / min) and, if they differ, take rank 0's copy and count
        # the event (``divergences``); 0 / unset: off.
        try:
            self.check_every = int(os.environ.get("DSVGP_DP_CHECK", "0"))
        except ValueError:
            self.check_every = 0
        self._check_step = 0
        self.checks = 0
        self.divergences = 0

    def check_replicas(self, tensors, force=False):
        """Divergence check (DSVGP_DP_CHECK=N, ``check_every``): True if the replicas of ``tensors`` agree bit for bit on every rank.
        Two checksums per tensor over its BIT PATTERN (the words viewed as int32: their int64 sum, and their sum weighted by
        1 + position mod 65521 -- a flipped bit, a NaN payload or two swapped entries all change it; NaNs compare like any other
        bits, so a NaN that every replica holds is not a divergence) go through ONE all-reduce(max) of [c, -c]: the replicas agree
        iff max(c) == min(c).  On disagreement every tensor is overwritten with rank 0's copy (broadcast) and ``divergences`` is
        incremented; the caller decides what else to re-synchronise (optimizer moments).  Runs on the eager step; a graph-replayed
        step is never data-parallel (TrainLoop._graph_eligible), so there is no replayed variant to cover."""
        self._check_step += 1
        if not force and (self.check_every <= 0 or self._check_step % self.check_every):
            return True
        self.checks += 1
        cs = []
        for t in tensors:
            w = t.detach().contiguous().view(-1)
            if w.numel() == 0:
                continue
            bits = w.view(torch.int32).to(torch.int64)          # (float64: two words per element)
            pos = torch.arange(bits.numel(), device=bits.device, dtype=torch.int64).remainder_(65521).add_(1)
            cs += [bits.sum(), (bits * pos).sum()]
        if not cs:
            return True
        c = torch.stack(cs)
        both = torch.cat([c, -c])
        dist.all_reduce(both, op=dist.ReduceOp.MAX, group=self.group)
        n = c.numel()
        same = bool(torch.equal(both[:n], -both[n:]))
        if not same:
            self.divergences += 1
            for t in tensors:
                dist.broadcast(t.detach(), self.src0, group=self.group)
        return same

    def shard_bounds(self, n):
        """rows [lo, hi) of a global batch of n rows owned by this rank (ragged tails go to low ranks)."""
        base, rem = divmod(n, self.world)
        lo = self.rank * base + min(self.rank, rem)
        return lo, lo + base + (1 if self.rank < rem else 0)

    def broadcast_floats(self, values, device):
        """rank 0's host scalars on every rank (e.g. the eigenvalue bounds of the CIQ quadrature)"""
        t = torch.tensor(values, dtype=torch.float64, device=device)
        dist.broadcast(t, self.src0, group=self.group)
        return [float(v) for v in t.tolist()]

    def all_reduce_sum(self, t):
        """sum over the ranks, in place, ordered on the current stream (the closing reduction of the step)"""
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def all_reduce_async(self, t):
        """sum over the ranks, asynchronously on the collective's own stream; ``.wait()`` orders the current stream after it.
        ``algo = "rs_ag"``: one reduce-scatter + one all-gather (every rank owns 1/world of the buffer: with RCCL's direct
        algorithms all 7 xGMI peers of a rank carry traffic at once instead of one ring neighbour); buffers that do not
        split evenly keep the plain all-reduce."""
        if self.algo == "rs_ag" and self.world > 1 and t.is_contiguous() and t.numel() % self.world == 0 \
                and t.numel() >= self.rs_ag_min_numel:
            return self._rs_ag_async(t.view(-1))
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def all_gather_async(self, out, inp):
        """out[world * n] <- the ranks' inp[n] in rank order, asynchronously (``.wait()`` orders the current stream after it).
        RCCL: one all_gather_into_tensor; gloo (CPU rehearsal / several ranks on one card): the list form."""
        if dist.get_backend(self.group) == "gloo":
            n = inp.numel()
            views = [out.view(-1)[r * n:(r + 1) * n] for r in range(self.world)]
            return dist.all_gather(views, inp.view(-1), group=self.group, async_op=True)
        return dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=self.group, async_op=True)

    def _rs_ag_async(self, flat):
        n = flat.numel() // self.world
        shard = self._shards.get((flat.dtype, n))
        if shard is None or shard.device != flat.device:
            shard = self._shards[(flat.dtype, n)] = torch.empty(n, dtype=flat.dtype, device=flat.device)
        if dist.get_backend(self.group) == "gloo":
            # gloo has no reduce-scatter: the same data movement as world point-to-point reductions (CPU rehearsal only)
            for r in range(self.world):
                dist.reduce(flat[r * n:(r + 1) * n], dst=dist.get_global_rank(self.group, r) if self.group is not None else r,
                            op=dist.ReduceOp.SUM, group=self.group)
            shard.copy_(flat[self.rank * n:(self.rank + 1) * n])
            h = dist.all_gather_into_tensor(flat, shard, group=self.group, async_op=True)
            return h
        h1 = dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        h2 = dist.all_gather_into_tensor(flat, shard, group=self.group, async_op=True)   # same communicator: stream-ordered

        class _Pair:
            def wait(self_inner):
                h1.wait()
                h2.wait()

        return _Pair()

    def loss_and_grads(self, engine, params, x, y, D, num_data, mll_type):
        """x, y, D are this rank's shard.  Returns globally reduced (loss, grads, local mu, local varn)."""
        p1 = y.shape[0] // max(x.shape[0], 1) if x.shape[0] else 1
        rows = (self.global_batch if self.global_batch is not None else x.shape[0] * self.world) * p1
        # engines with a flat gradient buffer use this object as their collective while the step runs (``all_reduce_async``):
        # either [G ; b^T] is summed early and m-bar / L_S-bar come back global, or they are reduced as soon as they are
        # final, under the rest of the backward; what is left for the end of the step is the small ``flat_late`` segment
        if self.replicated_step:
            # fewer rows than ranks (ragged tail of an epoch): an empty shard is not a case the kernels take, so every rank
            # computes the whole (tiny) batch and rank 0's result is broadcast -- replicas stay bit-identical
            loss, grads, mu, varn = engine.loss_and_grads(params, x, y, D, num_data, mll_type)
            flat = getattr(engine, "flat", None)
            if flat is not None:
                dist.broadcast(flat, self.src0, group=self.group)
                loss = flat[-1]
            else:
                for k in grads:
                    dist.broadcast(grads[k], self.src0, group=self.group)
                loss = loss.clone()
                dist.broadcast(loss, self.src0, group=self.group)
            return loss, grads, mu, varn
        hooked = hasattr(engine, "collective")
        if hooked:
            engine.collective = self
        try:
            loss, grads, mu, varn = engine.loss_and_grads(params, x, y, D, num_data, mll_type, global_rows=rows,
                                                          include_kl=(self.rank == 0))
        finally:
            if hooked:
                engine.collective = None
        flat = getattr(engine, "flat", None)
        names = list(grads.keys())
        ptrs = [grads[k].data_ptr() for k in names if grads[k].numel()]    # (an empty gradient, p = 0, has no address)
        if flat is None or not ptrs or flat.data_ptr() != min(ptrs):                      # engines without a flat buffer
            if getattr(engine, "variational_grads_global", False):
                raise RuntimeError("engine returned globally reduced gradients outside its flat buffer")
            flat = torch.cat([grads[k].reshape(-1) for k in names] + [loss.reshape(1).to(grads[names[0]].dtype)])
            views = None
        else:
            views = grads                       # the engine's gradients ARE views of [grads..., loss]
        early = getattr(engine, "_early_handle", None) if views is not None else None
        # (bench.py: HIP events around the final reduction on the main stream = the communication the step does not hide)
        ev = engine._event_pair() if hasattr(engine, "_event_pair") else None
        if views is not None and getattr(engine, "variational_grads_global", False):
            self.all_reduce_sum(engine.flat_late)     # [Z-bar, V-bar, scalars, loss]
        elif early is not None:
            self.all_reduce_sum(engine.flat_late)
            early.wait()
            engine._early_handle = None
        else:
            self.all_reduce_sum(flat)
        if ev is not None:
            engine._event_done("final_reduce", ev)
        if views is not None:
            return flat[-1], views, mu, varn
        off = 0
        out = {}
        for k in names:
            n = grads[k].numel()
            out[k] = flat[off:off + n].view_as(grads[k])
            off += n
        return flat[off], out, mu, varn
