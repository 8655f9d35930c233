"""CPU, world_size 2 over gloo: the data-parallel wrapper (row sharding, global normalisation, KL on
one rank, single flat all-reduce) reproduces the single-process step.  The compute engine is swapped
for the CPU oracle here -- the HIP engine itself is covered by the -m gpu tests -- so this exercises
exactly the N>1 logic that bench.py / train_gp run over RCCL on the GPU box."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _mpfiles import FileDict

import dsvgp_oracle as O


class OracleEngine:
    """Same call signature as dsvgp_amd.ElboEngine.loss_and_grads, computed by the oracle."""

    def loss_and_grads(self, params, x, y, D, num_data, mll_type="ELBO", global_rows=None, include_kl=True):
        ps = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
        loss, mu, varn = O.elbo_forward(ps, x, y, D, num_data, mll_type, global_rows)
        if not include_kl:
            loss = loss - O.kl_whitened(ps["variational_mean"], torch.tril(ps["chol_variational_covar"])) / num_data
        loss.backward()
        grads = {k: (ps[k].grad if ps[k].grad is not None else torch.zeros_like(ps[k])) for k in O.PARAM_NAMES}
        return loss.detach(), grads, mu.detach(), varn.detach()


def _flat_engine(dsvgp_amd):
    """The product engine's own gradient layout and early-reduce protocol (ElboEngine._alloc_grads /
    _variational_grads_final), with the numbers supplied by the oracle: the m-bar / L_S-bar segment is handed to the
    data-parallel layer BEFORE the remaining gradients are written, as in the HIP step."""

    class FlatOracleEngine(dsvgp_amd.ElboEngine):
        def loss_and_grads(self, params, x, y, D, num_data, mll_type="ELBO", global_rows=None, include_kl=True):
            loss, g, mu, varn = OracleEngine().loss_and_grads(params, x, y, D, num_data, mll_type, global_rows, include_kl)
            p32 = {k: v.float() for k, v in params.items()}
            grads, loss_out, _ = self._alloc_grads(p32, O.PARAM_NAMES)
            for k in ("variational_mean", "chol_variational_covar"):
                grads[k].copy_(g[k])
            self._allow_early = True
            self._variational_grads_final()
            self.fired = self._early_handle is not None
            for k in O.PARAM_NAMES:
                if k not in ("variational_mean", "chol_variational_covar"):
                    grads[k].copy_(g[k])
            loss_out.copy_(loss.reshape(1))
            return loss_out[0], grads, mu, varn

        # CPU stand-ins for the two HIP pack kernels (dsvgp_tril_pack_f32 / _unpack_f32): same packed layout
        def _tril_pack(self, ctx, src, extra, dst):
            n = src.shape[0]
            i, j = torch.tril_indices(n, n)
            dst[:i.numel()] = src[i, j]
            dst[i.numel():i.numel() + extra.numel()] = extra

        def _tril_unpack(self, ctx, src, dst, extra):
            n = dst.shape[0]
            i, j = torch.tril_indices(n, n)
            dst[i, j] = src[:i.numel()]
            extra.copy_(src[i.numel():i.numel() + extra.numel()])

    return FlatOracleEngine(torch.device("cpu"))


def _problem():
    g = torch.Generator().manual_seed(0)
    N, d, M, p, B = 80, 4, 7, 2, 11          # odd global batch: ragged shards
    X = torch.rand(N, d, generator=g, dtype=torch.float64)
    Y = O.testfun(X)
    P = O.init_params(X[:M], torch.eye(d)[:p].repeat(M, 1), torch.float64, 0.3, g)
    P["chol_variational_covar"] = torch.eye(M * 3, dtype=torch.float64) + 0.1 * torch.randn(M * 3, M * 3, generator=g, dtype=torch.float64).tril()
    x = X[20:20 + B]
    y = Y[20:20 + B][:, [0, 2, 3]].reshape(-1)
    D = torch.eye(d, dtype=torch.float64)[[1, 2]].repeat(B, 1)
    return P, x, y, D, (d + 1) * N, p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dsvgp_amd
    P, x, y, D, nd, p = _problem()
    dp = dsvgp_amd.DataParallel()
    lo, hi = dp.shard_bounds(x.shape[0])
    dp.global_batch = x.shape[0]
    loss, grads, mu, varn = dp.loss_and_grads(OracleEngine(), P, x[lo:hi], y[lo * (p + 1):hi * (p + 1)],
                                              D[lo * p:hi * p], nd, "ELBO")
    # the same step through the product engine's flat buffer: early (overlapped) + late all-reduce
    eng = _flat_engine(dsvgp_amd)
    loss2, grads2, _, _ = dp.loss_and_grads(eng, P, x[lo:hi], y[lo * (p + 1):hi * (p + 1)], D[lo * p:hi * p], nd, "ELBO")
    assert eng.fired and eng.collective is None and eng._early_handle is None
    # the early operand travelled as [tril(L_S-bar) | m-bar]: about half of the dense [m-bar, L_S-bar] segment
    Mp = P["variational_mean"].shape[0]
    assert eng.pack_reduce and eng.early_wire_numel == (Mp * (Mp + 1) // 2 + Mp + 2047) // 2048 * 2048
    assert Mp * (Mp + 1) // 2 + Mp < 0.55 * eng.flat_early.numel()
    # ... and the dense operand / the reduce-scatter + all-gather algorithm give the same numbers
    for pack, algo in ((False, "allreduce"), (True, "rs_ag"), (False, "rs_ag")):
        eng3 = _flat_engine(dsvgp_amd)
        eng3.pack_reduce = pack
        dp.algo, dp.rs_ag_min_numel = algo, 1
        loss3, grads3, _, _ = dp.loss_and_grads(eng3, P, x[lo:hi], y[lo * (p + 1):hi * (p + 1)], D[lo * p:hi * p], nd, "ELBO")
        e3 = max((grads3[k].double() - grads[k]).abs().max().item() / (1e-30 + grads[k].abs().max().item()) for k in grads)
        assert e3 < 1e-6 and abs(loss3.item() - loss.item()) < 1e-5 * abs(loss.item()), (pack, algo, e3)
    dp.algo = "allreduce"
    # a tail minibatch with fewer rows than ranks: computed on every rank, rank 0's gradients broadcast (no empty shard)
    dp.replicated_step = True
    dp.global_batch = 1
    l1, g1, _, _ = dp.loss_and_grads(OracleEngine(), P, x[:1], y[:p + 1], D[:p], nd, "ELBO")
    dp.replicated_step = False
    l1r, g1r, _, _ = O.elbo_loss_and_grads(P, x[:1], y[:p + 1], D[:p], nd)
    assert abs(l1.item() - l1r.item()) < 1e-12 and all((g1[k] - g1r[k]).abs().max().item() < 1e-12 for k in g1r)
    assert grads2["variational_mean"].data_ptr() == eng.flat.data_ptr()       # variational segment leads the buffer
    assert eng.flat_early.numel() + eng.flat_late.numel() == eng.flat.numel()
    err = max((grads2[k].double() - grads[k]).abs().max().item() / (1e-30 + grads[k].abs().max().item()) for k in grads)
    assert err < 1e-6 and abs(loss2.item() - loss.item()) < 1e-5 * abs(loss.item()), (err, loss2.item(), loss.item())
    # the collectives of the sharded replicated stage (round 3): every rank contributes its column block of [Q' | a] (width a
    # multiple of 4, zero padded) and its row block of L-bar; all-gather + the engine's reassembly give the whole matrices
    n = 13
    g = torch.Generator().manual_seed(9)
    Q = torch.randn(n, n + 1, generator=g)
    w = ((n + 1 + world - 1) // world + 3) // 4 * 4
    c0 = min(rank * w, n + 1)
    c1 = min(c0 + w, n + 1)
    loc = torch.zeros(n, w)
    loc[:, :c1 - c0] = Q[:, c0:c1]
    allq = torch.empty(world, n, w)
    dp.all_gather_async(allq, loc).wait()
    full = torch.empty(n, world * w)
    full.view(n, world, w).copy_(allq.permute(1, 0, 2))
    assert torch.equal(full[:, :n + 1], Q) and full[:, n + 1:].abs().max().item() == 0.0
    wr = ((n + world - 1) // world + 1) // 2 * 2       # (even, as both orchestrations of the sharded stage size it)
    r0 = min(rank * wr, n)
    r1 = min(r0 + wr, n)
    rows = torch.zeros(wr, n)
    rows[:r1 - r0] = Q[r0:r1, :n]
    allr = torch.empty(world * wr, n)
    dp.all_gather_async(allr, rows).wait()
    assert torch.equal(allr[:n], Q[:, :n])
    # opt-in divergence check (DSVGP_DP_CHECK): identical replicas pass; a replica that drifted in ONE element is detected on every
    # rank and overwritten with rank 0's copy; off by default (every call returns True without a collective)
    a, b = torch.arange(12.0).reshape(3, 4), torch.linspace(0, 1, 5)
    assert dp.check_every == 0 and dp.check_replicas([a, b]) is True
    assert dp.check_replicas([a, b], force=True) is True and dp.divergences == 0
    if rank == 1:
        a[2, 3] += 1e-6
    assert dp.check_replicas([a, b], force=True) is False and dp.divergences == 1
    assert torch.equal(a, torch.arange(12.0).reshape(3, 4))
    dp.check_every, dp._check_step = 2, 0
    assert dp.check_replicas([a, b]) is True and dp.check_replicas([a, b]) is True and dp.divergences == 1
    # the checksum is over the bit pattern: two swapped entries (equal sum and sum of squares) are a divergence, a NaN every replica
    # holds is not (it must not trigger a re-broadcast on every check), float64 and 0-dim tensors are taken as they are
    c = torch.arange(8.0)
    if rank == 1:
        c[[2, 5]] = c[[5, 2]]
    assert dp.check_replicas([c], force=True) is False and dp.divergences == 2 and torch.equal(c, torch.arange(8.0))
    e, f = torch.tensor([1.0, float("nan"), 3.0]), torch.tensor(0.5, dtype=torch.float64)
    assert dp.check_replicas([e, f, torch.zeros(0)], force=True) is True and dp.divergences == 2 and dp.checks >= 4
    out[rank] = (loss.item(), {k: v.clone() for k, v in grads.items()}, (lo, hi), mu.shape[0])
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_equals_single_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = FileDict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    out = out.collect(range(2))
    P, x, y, D, nd, p = _problem()
    l_ref, g_ref, _, _ = O.elbo_loss_and_grads(P, x, y, D, nd)
    assert out[0][2] == (0, 6) and out[1][2] == (6, 11)            # ragged tail goes to the low rank
    assert out[0][3] == 18 and out[1][3] == 15
    for r in (0, 1):
        loss, grads = out[r][0], out[r][1]
        assert abs(loss - l_ref.item()) < 1e-12
        for k in g_ref:
            assert (grads[k] - g_ref[k]).abs().max().item() < 1e-12, (r, k)
