"""GPU, 3 ranks on ONE card over gloo (a rehearsal of the N > 1 path: the driver's multi-GPU runs put one rank per GPU
over RCCL): the HIP engine under ``DataParallel`` -- row shards, the early all-reduce of [G ; b^T] with the column-split
Cholesky backward ("global Gram" schedule), and the early all-reduce of the variational gradients (general schedule) --
reproduces the single-process step."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _mpfiles import FileDict

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_step import make_problem
    return make_problem(600, 5, 40, 2, 97, seed=5)      # 97 rows over 3 ranks: ragged shards; 40 inducing points: 14/13/13


def _variant_problems():
    """(name, engine flags, params, x, y, D, num_data, p_data): the parameterisations of the SURVEY 8f variants"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dsvgp_oracle as O
    from test_gpu_step import make_problem
    out = []
    P, x, _, D, nd = make_problem(400, 3, 16, 2, 50, seed=11)                      # derivative-free data
    out.append(("dfree", dict(data_outputs="values"), P, x, O.testfun(x)[:, 0].contiguous(), D, nd, 0))
    P, x, y, D, nd = make_problem(400, 4, 14, 2, 50, seed=21)                      # shared inducing directions
    g = torch.Generator().manual_seed(4)
    P["inducing_directions"] = torch.eye(4)[:2] + 0.2 * torch.randn(2, 4, generator=g)
    P["variational_mean"] = 0.3 * torch.randn(16, generator=g)
    P["chol_variational_covar"] = torch.eye(16) + 0.05 * torch.randn(16, 16, generator=g)
    out.append(("shared", dict(shared_directions=True), P, x, y, D, nd, 2))
    from test_ngd import make_ngd_problem
    P, x, y, D, nd = make_ngd_problem(400, 2, 20, 2, 50, seed=402)                 # natural parameters (NGD)
    out.append(("ngd", {}, P, x, y, D, nd, 2))
    P, x, y, D, nd = make_problem(300, 6, 30, 0, 50, seed=3)                       # p = 0: plain SVGP
    out.append(("plain", {}, P, x, y, D, nd, 0))
    P, x, y, D, nd = make_ngd_problem(400, 2, 20, 2, 60, seed=402)                 # CIQ whitening (iterative: looser)
    out.append(("ciq", dict(whitening="ciq"), P, x, y, D, nd, 2))
    return out


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dsvgp_amd
    dev = torch.device("cuda", 0)
    P, x, y, D, nd = _problem()
    p = 2
    dp = dsvgp_amd.DataParallel()
    lo, hi = dp.shard_bounds(x.shape[0])
    dp.global_batch = x.shape[0]
    Pg = {k: v.to(dev) for k, v in P.items()}
    xs, ys, Ds = x[lo:hi].to(dev), y[lo * (p + 1):hi * (p + 1)].to(dev), D[lo * p:hi * p].to(dev)
    res = {}
    for mode, mll in (("global", "ELBO"), ("globalshard", "ELBO"), ("globalshardpy", "ELBO"), ("globaldet", "ELBO"), ("early", "ELBO"),
                      ("early", "PLL")):
        eng = dsvgp_amd.ElboEngine(dev)
        eng.global_gram = mode.startswith("global")
        eng.shard_replicated = mode.startswith("globalshard")     # Q' columns / L-bar rows per rank + two all-gathers
        eng.c_step = mode != "globalshardpy"                # "globalshard": the five-piece C entry point (dsvgp_elbo_step_dp_f32)
        eng.shard_min_mp = 0
        eng.deterministic = mode == "globaldet"             # fixed-order sums: the replicas' L_S-bar must be BITWISE equal
        loss, grads, mu, varn = dp.loss_and_grads(eng, Pg, xs, ys, Ds, nd, mll)
        torch.cuda.synchronize()
        assert eng.variational_grads_global == mode.startswith("global"), (mode, mll)
        assert getattr(eng, "sharded_stage_used", False) == mode.startswith("globalshard"), mode
        assert eng.c_step_used == (mode == "globalshard"), (mode, eng.c_step_used)
        assert eng.collective is None and eng._early_handle is None
        res[mode + mll] = (loss.item(), {k: v.cpu().clone() for k, v in grads.items()})
    for name, flags, P, x, y, D, nd, pdata in _variant_problems():
        lo, hi = dp.shard_bounds(x.shape[0])
        dp.global_batch = x.shape[0]
        pz = D.shape[0] // x.shape[0]
        eng = dsvgp_amd.ElboEngine(dev)
        for k, v in flags.items():
            setattr(eng, k, v)
        Pg = {k: v.to(dev) for k, v in P.items()}
        loss, grads, _, _ = dp.loss_and_grads(eng, Pg, x[lo:hi].to(dev), y[lo * (pdata + 1):hi * (pdata + 1)].to(dev),
                                              D[lo * pz:hi * pz].to(dev), nd, "ELBO")
        torch.cuda.synchronize()
        res[name] = (loss.item(), {k: v.cpu().clone() for k, v in grads.items()})
    out[rank] = res
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def dp_results(gpu_device):
    """ONE 3-rank run (3 GPU processes + this one) for all the tests of this module"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = FileDict()
    mp.spawn(_worker, args=(3, port, out), nprocs=3, join=True)
    return out.collect(range(3))


@pytest.mark.timeout(600)
def test_three_ranks_on_one_gpu_equal_single_process(dsvgp, gpu_device, dp_results):
    out = dp_results
    P, x, y, D, nd = _problem()
    Pg = {k: v.to(gpu_device) for k, v in P.items()}
    for mll in ("ELBO", "PLL"):
        eng = dsvgp.ElboEngine(gpu_device)
        l1, g1, _, _ = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd, mll)
        for mode in (("global", "globalshard", "globalshardpy", "globaldet", "early") if mll == "ELBO" else ("early",)):
            for r in range(3):
                loss, grads = out[r][mode + mll]
                assert abs(loss - l1.item()) < 2e-5 * abs(l1.item()), (mode, mll, r, loss, l1.item())
                for k in g1:
                    ref = g1[k].double().cpu()
                    err = (grads[k].double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
                    assert err < 2e-4, (mode, mll, r, k, err)
            # identical on every rank (what the replicated Adam step relies on)
            for k in g1:
                assert torch.equal(out[0][mode + mll][1][k], out[1][mode + mll][1][k]) or \
                    (out[0][mode + mll][1][k] - out[1][mode + mll][1][k]).abs().max().item() < 1e-6 * max(
                        g1[k].abs().max().item(), 1e-30), (mode, mll, k)


@pytest.mark.timeout(600)
def test_deterministic_replicas_are_bitwise_equal(dsvgp, gpu_device, dp_results):
    """global-Gram schedule: every rank forms L_S-bar / m-bar itself from the same reduced [G ; b^T]; the one product behind them
    (G L_S) adds its K slices in a fixed order on EVERY multi-rank path (round 4; the whole step under ``ElboEngine.deterministic``),
    so the replicas agree bit for bit and nothing re-broadcasts the variational parameters"""
    out = dp_results
    for mode in ("global", "globalshard", "globalshardpy", "globaldet"):
        for k in ("variational_mean", "chol_variational_covar"):
            for r in (1, 2):
                assert torch.equal(out[0][mode + "ELBO"][1][k], out[r][mode + "ELBO"][1][k]), (mode, k, r)


@pytest.mark.timeout(600)
def test_variants_under_data_parallel(dsvgp, gpu_device, dp_results):
    """derivative-free data, shared directions, natural parameters and p = 0 through DataParallel on 3 ranks (one card, gloo)
    against the single-process step of the same engine configuration."""
    out = dp_results
    for name, flags, P, x, y, D, nd, pdata in _variant_problems():
        eng = dsvgp.ElboEngine(gpu_device)
        for k, v in flags.items():
            setattr(eng, k, v)
        Pg = {k: v.to(gpu_device) for k, v in P.items()}
        l1, g1, _, _ = eng.loss_and_grads(Pg, x.to(gpu_device), y.to(gpu_device), D.to(gpu_device), nd, "ELBO")
        for r in range(3):
            loss, grads = out[r][name]
            ltol, gtol = (2e-3, 3e-2) if name == "ciq" else (3e-5, 5e-4)
            assert abs(loss - l1.item()) < ltol * abs(l1.item()), (name, r, loss, l1.item())
            assert set(grads) == set(g1)
            for k in g1:
                ref = g1[k].double().cpu()
                if ref.numel() == 0 or ref.abs().max().item() == 0:
                    continue
                err = (grads[k].double() - ref).abs().max().item() / ref.abs().max().item()
                assert err < gtol, (name, r, k, err)


# ------------------------------------------------------------------ the drop-in harnesses under torch.distributed
def _harness_cases(dsvgp_amd):
    """(name, train_gp callable(dataset, **kw), dataset builder) for every train_gp mirror"""
    import dsvgp_oracle as O
    from torch.utils.data import TensorDataset
    g = torch.Generator().manual_seed(7)
    X = torch.rand(360, 3, generator=g)
    Y = O.testfun(X)
    full, vals = TensorDataset(X, Y), TensorDataset(X, Y[:, 0].contiguous())
    kw = dict(minibatch_size=90, num_epochs=1, verbose=False, seed=11, max_steps=3, tqdm=False)
    return [
        ("directional_vi", lambda: dsvgp_amd.directional_vi.train_gp(full, num_inducing=12, num_directions=2, minibatch_dim=2,
                                                                     inducing_data_initialization=False, **kw)),
        ("directional_vi_ngd", lambda: dsvgp_amd.directional_vi.train_gp(full, num_inducing=12, num_directions=2,
                                                                         minibatch_dim=2, use_ngd=True,
                                                                         inducing_data_initialization=False, **kw)),
        ("grad_svgp", lambda: dsvgp_amd.grad_svgp.train_gp(full, 3, num_inducing=12, **kw)),
        ("dfree", lambda: dsvgp_amd.dfree_directional_vi.train_gp(full, num_inducing=12, num_directions=2, minibatch_dim=2,
                                                                  inducing_data_initialization=False, **kw)),
        ("shared", lambda: dsvgp_amd.shared_directional_vi.train_gp(full, num_inducing=12, num_directions=2, minibatch_dim=2,
                                                                    inducing_data_initialization=False, **kw)),
        ("traditional", lambda: dsvgp_amd.traditional_vi.train_gp(vals, 3, num_inducing=12, **kw)),
    ]


def _state(model, likelihood):
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if v.dtype.is_floating_point}
    sd.update({"lik." + k: v.detach().cpu().clone() for k, v in likelihood.state_dict().items()})
    return sd


def _harness_worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dsvgp_amd
    torch.cuda.set_device(0)
    res = {}
    for name, run in _harness_cases(dsvgp_amd):
        torch.manual_seed(100)                       # the random initialisations (Z ~ U, 1e-3 randn mean) are rank-0's anyway
        model, likelihood = run()
        torch.cuda.synchronize()
        res[name] = _state(model, likelihood)
    out[rank] = res
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_every_train_gp_mirror_runs_data_parallel(dsvgp, gpu_device):
    """``train_gp`` of every mirrored module on 2 ranks (gloo, one card): parameters identical across ranks after 3 steps and
    equal to the single-process run from the same seeds (global minibatch unchanged: rows are sharded, not re-sampled)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = FileDict()
    mp.spawn(_harness_worker, args=(2, port, out), nprocs=2, join=True)
    out = out.collect(range(2))
    for name, run in _harness_cases(dsvgp):
        torch.manual_seed(100)
        model, likelihood = run()
        ref = _state(model, likelihood)
        a, b = out[0][name], out[1][name]
        assert set(a) == set(ref)
        for k in ref:
            scale = max(ref[k].abs().max().item(), 1e-3) if ref[k].numel() else 1.0
            if ref[k].numel() == 0:
                continue
            assert (a[k] - b[k]).abs().max().item() <= 1e-6 * scale, (name, k, "ranks differ")
            assert (a[k] - ref[k]).abs().max().item() < 2e-3 * scale + 2e-4, (name, k, (a[k] - ref[k]).abs().max().item())


# ------------------------------------------------------------------ the same code path on a real RCCL communicator
def _rccl_worker(rank, world, port, out):
    """A DataParallel over a ONE-rank RCCL group made to take the N > 1 schedules (the collectives become identities on
    RCCL's own stream: what is exercised is the asynchronous all-reduce / wait protocol of the nccl backend, which the gloo
    rehearsals cannot show)."""
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import dsvgp_amd
    P, x, y, D, nd = _problem()
    Pg = {k: v.to(dev) for k, v in P.items()}
    xg, yg, Dg = x.to(dev), y.to(dev), D.to(dev)
    l0, g0, _, _ = dsvgp_amd.ElboEngine(dev).loss_and_grads(Pg, xg, yg, Dg, nd)
    res = {}
    for gg in (True, False):
        dp = dsvgp_amd.DataParallel()
        dp.world, dp.rank = 2, 0                       # schedules of a 2-rank job; this rank holds every row
        dp.shard_bounds = lambda n: (0, n)
        dp.global_batch = x.shape[0]
        eng = dsvgp_amd.ElboEngine(dev)
        eng.global_gram = gg
        loss, grads, _, _ = dp.loss_and_grads(eng, Pg, xg, yg, Dg, nd, "ELBO")
        torch.cuda.synchronize()
        assert eng.variational_grads_global == gg
        # m-bar / L_S-bar / loss are complete on this rank; Z-bar etc. lack the K_ZZ-bar columns of the absent rank when gg
        res[gg] = (abs(loss.item() - l0.item()) / abs(l0.item()),
                   (grads["chol_variational_covar"] - g0["chol_variational_covar"]).abs().max().item()
                   / g0["chol_variational_covar"].abs().max().item(),
                   (grads["variational_mean"] - g0["variational_mean"]).abs().max().item()
                   / g0["variational_mean"].abs().max().item(),
                   (grads["inducing_points"] - g0["inducing_points"]).abs().max().item()
                   / g0["inducing_points"].abs().max().item())
    # the all-gather of the sharded replicated stage on the nccl backend (one rank: an identity on RCCL's stream, asynchronous)
    dp1 = dsvgp_amd.DataParallel()
    inp = torch.randn(37, 12, device=dev)
    got = torch.zeros(1, 37, 12, device=dev)
    dp1.all_gather_async(got, inp).wait()
    torch.cuda.synchronize()
    res["allgather"] = bool(torch.equal(got[0], inp))
    out[0] = res
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_multi_rank_schedules_on_an_rccl_communicator(dsvgp, gpu_device):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = FileDict()
    mp.spawn(_rccl_worker, args=(1, port, out), nprocs=1, join=True)
    out = out.collect(range(1))
    res = out[0]
    for gg in (True, False):
        dl, dLS, dm, dZ = res[gg]
        assert dl < 1e-5 and dLS < 1e-4 and dm < 1e-4, (gg, res[gg])
    assert res[False][3] < 1e-4            # general schedule: everything is local, so Z-bar is complete too
    assert res["allgather"]


# ------------------------------------------------------------------ the sharded replicated stage at the size where it switches on
def _c4shard_worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dsvgp_amd
    from test_gpu_reftext import _errors, _load
    dev = torch.device("cuda", 0)
    g, P, x, y, D, nd = _load("c4shard")
    p = D.shape[0] // x.shape[0]
    dp = dsvgp_amd.DataParallel()
    lo, hi = dp.shard_bounds(x.shape[0])
    dp.global_batch = x.shape[0]
    Pg = {k: v.to(dev) for k, v in P.items()}
    xs, ys, Ds = x[lo:hi].to(dev), y[lo * (p + 1):hi * (p + 1)].to(dev), D[lo * p:hi * p].to(dev)
    res = {}
    for name, c_step in (("one-call", True), ("piecewise", False)):
        eng = dsvgp_amd.ElboEngine(dev)
        eng.c_step = c_step                           # defaults otherwise: global-Gram schedule, sharded stage from M' = 1024 up
        loss, grads, mu, varn = dp.loss_and_grads(eng, Pg, xs, ys, Ds, nd, "ELBO")
        torch.cuda.synchronize()
        assert eng.sharded_stage_used and eng.variational_grads_global and eng.c_step_used == c_step, name
        res[name] = dict(_errors(g, loss, grads, mu, varn)) if rank == 0 else None
        res[name + "/LS"] = grads["chol_variational_covar"].double().sum().item()      # (cheap cross-rank identity check)
        res[name + "/LSabs"] = grads["chol_variational_covar"].double().abs().sum().item()
        del eng
        torch.cuda.empty_cache()
    out[rank] = res
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_sharded_stage_at_c4_shard_geometry_against_reference_text(dsvgp, gpu_device):
    """The data-parallel step with the replicated M'^3 stage sharded (columns of [Q' | a], fp64 rows of L-bar, column blocks of
    K_ZZ-bar; two all-gathers) at M' = 3000 -- the size from which it is on by default -- on 2 ranks (one card, gloo), through the
    five-piece C entry point and through the piecewise path: loss and every gradient against the REFERENCE-TEXT vector of the C4
    per-rank shard (tests/golden/reftext_c4shard_step.npz), at the tolerances of the one-GPU step."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = FileDict()
    mp.spawn(_c4shard_worker, args=(2, port, out), nprocs=2, join=True)
    out = out.collect(range(2))
    from test_gpu_reftext import TOL32
    tol_loss, tol_head, tol_grad = TOL32["c4shard"]
    for name in ("one-call", "piecewise"):
        errs = out[0][name]
        print("[parity] reftext c4shard, 2 ranks, sharded stage (%s): %s" % (name, ", ".join("%s %.2e" % kv for kv in errs.items())))
        for k, v in errs.items():
            tol = tol_loss if k == "loss" else tol_head if k in ("mu", "varn") else tol_grad
            assert v < tol, (name, k, v, tol)
        # L_S-bar is formed by every rank itself: bitwise equal replicas
        assert out[0][name + "/LS"] == out[1][name + "/LS"] and out[0][name + "/LSabs"] == out[1][name + "/LSabs"], name


# ------------------------------------------------------------------ world = 4 and 8 through the five-piece C entry, one process
@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,M,p,B", [(8, 48, 2, 51), (8, 20, 5, 67), (4, 30, 2, 33)])
def test_virtual_ranks_through_the_dp_c_entry(dsvgp, gpu_device, world, M, p, B):
    """The shard bounds of an 8-rank (and 4-rank) job -- ragged row shards, 5-6 inducing points per rank in the column-split
    Cholesky backward, the Q' column blocks and L-bar row blocks of the sharded stage, the M' >= 4 world gate -- through
    ``dsvgp_elbo_step_dp_f32`` (phases 0-4) on ONE card: the GPU box admits at most 6 processes on a card, so the ranks are
    threads of this process with an in-process transport (tests/_virtual_ranks.py); engines, plans, workspaces and the C entry
    are the product's.  Loss and gradients against the one-rank step; the replicas' gradients bitwise equal."""
    from _virtual_ranks import VirtualWorld, make_virtual_dp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_step import make_problem
    d = 6
    P, x, y, D, nd = make_problem(500, d, M, p, B, seed=17 + world)
    dev = gpu_device
    Pg = {k: v.to(dev) for k, v in P.items()}
    eng1 = dsvgp.ElboEngine(dev)
    l1, g1, _, _ = eng1.loss_and_grads(Pg, x.to(dev), y.to(dev), D.to(dev), nd, "ELBO")
    torch.cuda.synchronize()
    l1, g1 = l1.item(), {k: v.double().cpu() for k, v in g1.items()}
    vw = VirtualWorld(world)

    def rank_fn(r, vw_):
        dp = make_virtual_dp(dsvgp, vw_, r)
        lo, hi = dp.shard_bounds(B)
        dp.global_batch = B
        eng = dsvgp.ElboEngine(dev)
        eng.global_gram, eng.shard_replicated, eng.c_step, eng.shard_min_mp = True, True, True, 0
        Pr = {k: v.clone() for k, v in Pg.items()}
        loss, grads, mu, _ = dp.loss_and_grads(eng, Pr, x[lo:hi].to(dev), y[lo * (p + 1):hi * (p + 1)].to(dev),
                                               D[lo * p:hi * p].to(dev), nd, "ELBO")
        torch.cuda.synchronize()
        assert eng.c_step_used and eng.sharded_stage_used and eng.variational_grads_global, (r, eng.c_step_used)
        assert mu.shape[0] == (hi - lo) * (p + 1)
        return loss.item(), {k: v.detach().cpu().clone() for k, v in grads.items()}, (lo, hi)

    out = vw.run(rank_fn)
    sizes = [hi - lo for _, _, (lo, hi) in out]
    assert sum(sizes) == B and max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
    for r, (loss, grads, _) in enumerate(out):
        assert abs(loss - l1) < 2e-5 * abs(l1), (world, r, loss, l1)
        for k, ref in g1.items():
            if ref.numel() == 0 or ref.abs().max().item() == 0:
                continue
            err = (grads[k].double() - ref).abs().max().item() / ref.abs().max().item()
            assert err < 3e-4, (world, r, k, err)
        for k in grads:                      # identical replicas: what the un-synchronised Adam steps rely on
            assert torch.equal(grads[k], out[0][1][k]), (world, r, k)


@pytest.mark.timeout(600)
def test_bench_line_of_a_four_rank_rehearsal(gpu_device):
    """`bench.py --gpus 4` as the driver launches it (torch.distributed.run, one process per rank), rehearsed on ONE card over gloo
    (DSVGP_REHEARSE_GLOO=1; four rank processes + this one stay under the box's six-process limit): the JSON line carries the
    world size, the collective block (backend, packed early operand, exposed waits) and a finite whole-job figure."""
    import json
    import subprocess
    env = dict(os.environ, DSVGP_REHEARSE_GLOO="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--config", "c4", "--steps", "4", "--warmup", "2",
           "--no-cpu-baseline", "--no-extras"]             # (C4: M' = 3000 >= ElboEngine.shard_min_mp, the regime of the five-piece C entry)
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=540)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 4 and j["rccl_ranks"] == 4 and j["steps"] == 4
    assert j["value"] > 0 and j["ms_per_step"] > 0 and j["config"]["per_gpu_batch"] * 4 == j["config"]["global_batch"]
    coll = j["config"]["collective"]
    assert coll["backend"] == "gloo" and coll["packed_triangle"] and coll["early_operand_floats"] > 0
    assert j["config"]["one_call_step"]                     # the five-piece C entry, not the piecewise orchestration


@pytest.mark.timeout(600)
def test_bench_self_launch_of_a_four_rank_rehearsal(gpu_device):
    """`python bench.py --gpus 4` from a plain invocation: bench.py starts ONE child launcher (`torch.distributed.run --standalone`)
    before anything in that process touches the GPU, relays its output and exits with its status -- rehearsed on one card over gloo.
    The line must carry what the first run on real multi-GPU hardware is to report: the replica check (on by default under the bench,
    no divergence), the host time of a rank step phase by phase, and the collective block."""
    import json
    import subprocess
    env = dict(os.environ, DSVGP_REHEARSE_GLOO="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DSVGP_DP_CHECK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--config", "c4", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-extras"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=540)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 4 and j["rccl_ranks"] == 4 and j["steps"] == 3 and j["value"] > 0
    coll = j["config"]["collective"]
    assert coll["backend"] == "gloo" and j["config"]["one_call_step"]
    rc = coll["replica_check"]
    assert rc["every"] == 64 and rc["checks"] >= 1 and rc["divergences"] == 0, rc
    host = coll["host_us_per_rank_step"]
    assert host["steps"] == 3 and host["total"] > 0
    for k in ("phase0_front_gram", "phase1_q_columns", "phase2_variational_lbar_rows", "phase3_dense_kernel_bwd", "phase4_chol_backward_tail",
              "wait_allreduce_G", "wait_allgather_q", "wait_allgather_lbar"):
        assert host[k] >= 0, (k, host)
    print("[dp] host us per rank step (4 gloo ranks on one card):", {k: round(v, 1) for k, v in host.items() if k not in ("unit",)})
