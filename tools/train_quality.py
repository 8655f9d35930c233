#!/usr/bin/env python3
"""End-to-end training run of a BASELINE configuration through the drop-in harness, with the quality figures the reference's
own test prints (/root/reference/tests/test_dsvgp.py:70-103: ``train_gp`` -> ``eval_gp`` -> MSE and mean negative predictive
density of the function values on held-out rows).  GPU box tool; nothing here is imported by the package.

  python3 tools/train_quality.py --config c4 --epochs 1 --data welch --out gpurun_out/train_c4.json
  python3 tools/train_quality.py --config c2 --against-oracle 50        # first 50 steps beside oracle/train_ref.py's op sequence

``--data sin``   the bench's synthetic function f = sin(2 pi |x|^2) on the unit cube (tests/testfun.py:4-12 in the reference);
``--data welch`` (d = 20 only) the 20-dimensional screening function of Welch et al. (1992) that the reference's experiments use
                 (utils/synthetic_functions.py:139-225), evaluated on [-0.5, 0.5]^20 with inputs mapped to the unit cube, f standardised
                 and the gradient scaled by 1 / sigma and by (ub - lb) as utils/load_data.py:17-34 does.  Restated here from the
                 published formula; the gradient is its analytic derivative.
The training loop is ``TrainLoop.step`` over shuffled epochs exactly as ``train_gp`` drives it (ragged last minibatch included);
the report steps (every 50th, reference :255-260) read the loss and the nll of that minibatch's function values.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def welch(Xu):
    """Welch et al. (1992) on [-0.5, 0.5]^20 for unit-cube inputs Xu: returns [f, df/dx_unit] (n x 21), float64"""
    X = Xu.double() - 0.5
    x = [X[:, i] for i in range(20)]
    f = (5 * x[11] / (1 + x[0]) + 5 * (x[3] - x[19]) ** 2 + x[4] + 40 * x[18] ** 3 - 5 * x[18] + 0.05 * x[1] + 0.08 * x[2]
         - 0.03 * x[5] + 0.03 * x[6] - 0.09 * x[8] - 0.01 * x[9] - 0.07 * x[10] + 0.25 * x[12] ** 2 - 0.04 * x[13]
         + 0.06 * x[14] - 0.01 * x[16] - 0.03 * x[17])
    g = torch.zeros(X.shape[0], 20, dtype=torch.float64, device=X.device)
    g[:, 0] = -5 * x[11] / (1 + x[0]) ** 2
    for i, c in ((1, 0.05), (2, 0.08), (4, 1.0), (5, -0.03), (6, 0.03), (8, -0.09), (9, -0.01), (10, -0.07), (13, -0.04),
                 (14, 0.06), (16, -0.01), (17, -0.03)):
        g[:, i] = c
    g[:, 3] = 10 * (x[3] - x[19])
    g[:, 11] = 5 / (1 + x[0])
    g[:, 12] = 0.5 * x[12]
    g[:, 18] = 120 * x[18] ** 2 - 5
    g[:, 19] = -10 * (x[3] - x[19])
    return torch.cat([f[:, None], g], 1)        # (ub - lb) = 1: the unit-cube gradient equals the gradient


def make_data(kind, n, d, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    X = torch.rand(n, d, device=device, generator=g)
    if kind == "welch":
        assert d == 20, "the Welch function is 20-dimensional"
        Y = welch(X)
    else:
        sq = (X * X).sum(1).double()
        Y = torch.cat([torch.sin(2 * math.pi * sq)[:, None], 4 * math.pi * torch.cos(2 * math.pi * sq)[:, None] * X.double()], 1)
    return X.contiguous(), Y


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c4", choices=["c4", "c2"])
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--data", default=None, choices=["sin", "welch"])
    ap.add_argument("--n-train", type=int, default=0, help="0 = the configuration's N")
    ap.add_argument("--n-test", type=int, default=100_000)
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--against-oracle", type=int, default=0,
                    help="C2-sized runs: this many steps beside oracle/train_ref's CPU trainer on IDENTICAL minibatches and columns")
    ap.add_argument("--out", default=None)
    ap.add_argument("--eval-batch", type=int, default=4096)
    return ap.parse_args(argv)


def run(args):
    import bench
    cfg = dict(bench.CONFIGS[args.config])
    d, M, p, B = cfg["d"], cfg["M"], cfg["p"], cfg["B"]
    N = args.n_train or cfg["N"]
    kind = args.data or ("welch" if d == 20 else "sin")
    dev = torch.device("cuda", 0)
    import dsvgp_amd
    from dsvgp_amd import directional_vi as dvi

    Xall, Yall = make_data(kind, N + args.n_test, d, dev, seed=0)
    # standardise f with the TRAINING rows' statistics, scale the gradient by 1 / sigma (utils/rescale.py:18-36)
    mu, sig = Yall[:N, 0].mean(), Yall[:N, 0].std(unbiased=True)
    Yall = torch.cat([((Yall[:, :1] - mu) / sig), Yall[:, 1:] / sig], 1).float().contiguous()
    Xtr, Ytr, Xte, Yte = Xall[:N].contiguous(), Yall[:N].contiguous(), Xall[N:].contiguous(), Yall[N:].contiguous()

    res = dict(workload=cfg["name"], data=kind, n_train=N, n_test=args.n_test, epochs=args.epochs, lr=args.lr,
               minibatch=B, steps_per_epoch=(N + B - 1) // B, ragged_tail_rows=N % B)

    if args.against_oracle > 0:
        res["against_oracle"] = against_oracle(dvi, Xtr, Ytr, cfg, args.against_oracle, args.lr)

    loop = dsvgp_amd.setup_training(None, num_inducing=M, num_directions=p, minibatch_size=B, minibatch_dim=p,
                                    num_epochs=args.epochs, learning_rate_hypers=args.lr, inducing_data_initialization=True,
                                    seed=0, tensors=(Xtr, Ytr))
    q = p + 1
    traj = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step = 0
    for ep in range(args.epochs):
        perm = loop.epoch_permutation()
        for s0 in range(0, N, B):
            report = step % 50 == 0
            loss, out, yb = loop.step(perm[s0:s0 + B], need_variance="values" if report else False)
            if report:
                means, stds = out.mean[::q], out.value_variance.sqrt()
                nll = -torch.distributions.Normal(means, stds).log_prob(yb[::q]).mean()
                traj.append(dict(step=step, epoch=ep, loss=float(loss.item()), batch_nll=float(nll.item()),
                                 rows=int(min(B, N - s0))))
            step += 1
    loop.finish()
    last_loss = float(loss.item())
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t0
    traj.append(dict(step=step - 1, epoch=args.epochs - 1, loss=last_loss, rows=int(N - (N - 1) // B * B)))
    res.update(total_steps=step, train_wall_s=t_train, train_ms_per_step_incl_reports=1e3 * t_train / step, trajectory=traj)

    model, lik = loop.model, loop.likelihood
    t0 = time.perf_counter()
    te_ds = torch.utils.data.TensorDataset(Xte, Yte)
    means, variances = dvi.eval_gp(te_ds, model, lik, num_directions=p, minibatch_size=args.eval_batch, minibatch_dim=p)
    t_eval = time.perf_counter() - t0
    yf = Yte[:, 0].cpu()
    mse = float(((means[::q] - yf) ** 2).mean())
    nll = float(-torch.distributions.Normal(means[::q], variances.sqrt()[::q]).log_prob(yf).mean())
    # the first p canonical derivative outputs too (eval_gp's directions are eye(d)[:p], directional_vi.py:292-294)
    dmse = [float(((means[1 + a::q] - Yte[:, 1 + a].cpu()) ** 2).mean()) for a in range(p)]
    res.update(eval_wall_s=t_eval, test_mse=mse, test_nll=nll, test_mse_of_a_constant_predictor=float((yf ** 2).mean()),
               test_mse_derivative_outputs=dmse, derivative_output_variance=[float(Yte[:, 1 + a].var()) for a in range(p)],
               variances_finite=bool(torch.isfinite(variances).all()), variance_min=float(variances.min()),
               hyperparameters=dict(lengthscale=float(torch.nn.functional.softplus(model.covar_module.base_kernel.raw_lengthscale).item()),
                                    outputscale=float(torch.nn.functional.softplus(model.covar_module.raw_outputscale).item()),
                                    noise=float(torch.nn.functional.softplus(lik.noise_covar.raw_noise).item() + 1e-4)))
    return res


def main():
    args = parse()
    res = run(args)
    print(json.dumps(res), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(json.dumps(res, indent=1) + "\n")


def against_oracle(dvi, Xtr, Ytr, cfg, nsteps, lr):
    """`nsteps` optimisation steps of the HIP harness and of the oracle's CPU trainer (oracle/train_ref.py's op sequence: four kernel
    assemblies, fp64 Cholesky + two solves, autograd, two torch.optim.Adam) from the SAME initial state on the SAME minibatches and
    derivative columns; returns both loss trajectories and their largest relative difference."""
    import random
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dsvgp_oracle as O
    import dsvgp_amd
    d, M, p, B = cfg["d"], cfg["M"], cfg["p"], cfg["B"]
    N = Xtr.shape[0]
    loop = dsvgp_amd.setup_training(None, num_inducing=M, num_directions=p, minibatch_size=B, minibatch_dim=p, num_epochs=1,
                                    learning_rate_hypers=lr, inducing_data_initialization=True, seed=1, tensors=(Xtr, Ytr))
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in loop.model._param_dict(loop.likelihood).items()}
    var = [P["variational_mean"], P["chol_variational_covar"]]
    hyp = [v for k, v in P.items() if k not in ("variational_mean", "chol_variational_covar")]
    opt_v, opt_h = torch.optim.Adam(var, lr=lr), torch.optim.Adam(hyp, lr=lr)
    Xc, Yc = Xtr.cpu(), Ytr.cpu()
    rng = random.Random(1)            # the loop's own column sampler is random.Random(seed): same sequence here
    perm = loop.epoch_permutation()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    gpu, cpu = [], []
    for k in range(nsteps):
        idx = perm[(k * B) % (N - B + 1):(k * B) % (N - B + 1) + B]
        loss, _, _ = loop.step(idx)
        gpu.append(float(loss.item()))
        cols = sorted(rng.sample(range(1, d + 1), p) + [0])
        xb, yb = Xc[idx.cpu()], Yc[idx.cpu()][:, cols].reshape(-1)
        D = torch.eye(d)[np.array(cols[1:]) - 1].repeat(B, 1)
        opt_v.zero_grad(); opt_h.zero_grad()
        l_ref, _, _ = O.elbo_forward(P, xb, yb, D, (d + 1) * N)
        l_ref.backward()
        opt_v.step(); opt_h.step()
        cpu.append(float(l_ref.detach()))
    rel = [abs(a - b) / max(abs(b), 1e-30) for a, b in zip(gpu, cpu)]
    return dict(steps=nsteps, hip_loss=gpu, oracle_loss=cpu, max_rel_diff=max(rel), rel_diff_at_last_step=rel[-1],
                note="same initial parameters, minibatches and derivative columns; the oracle trainer is plain torch on the CPU "
                     "(fp32 model, fp64 Cholesky / solves, torch.optim.Adam)")


if __name__ == "__main__":
    main()
