#!/usr/bin/env python3
"""Benchmark of the DSVGP ELBO training step (BASELINE.json metric) on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

A "step" is one iteration of ``directional_vi.train_gp``'s inner loop (reference
directionalvi/directional_vi.py:229-254): minibatch gather, fused ELBO forward + backward, both Adam
steps and both LR-scheduler steps.  Workload = BASELINE config 4: d=20, N=1M, M=500, p=5, global
minibatch 4096 (sharded by rows over the ranks, one RCCL all-reduce per step: strong scaling).
Synthetic data (X ~ U[0,1]^d, y=[f, grad f], f=sin(2 pi |x|^2), reference tests/testfun.py) is resident in
HBM before the timed region.  Rank 0 prints ONE JSON line with the roofline of the dominant kernel
(the fp64 MFMA panel-solve GEMM, timed live with HIP events on the launch stream) and the CPU
baseline (oracle step, reference-faithful op sequence, on this host's cores).
"""
import argparse
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    "c4": dict(name="DSVGP d=20 N=1M M=500 p=5 B=4096", d=20, N=1_000_000, M=500, p=5, B=4096),
    "c2": dict(name="DSVGP d=5 N=10k M=200 p=2 B=512", d=5, N=10_000, M=200, p=2, B=512),
    # diagnostics: the per-rank shard of C4 at 8 GPUs (B = 4096 / 8) run alone -- what one rank computes per step
    "c4shard8": dict(name="DSVGP d=20 N=1M M=500 p=5 B=512 (one rank's share of C4 at 8 GPUs)", d=20, N=1_000_000, M=500,
                     p=5, B=512),
    "c4shard4": dict(name="DSVGP d=20 N=1M M=500 p=5 B=1024 (one rank's share of C4 at 4 GPUs)", d=20, N=1_000_000, M=500,
                     p=5, B=1024),
    "c4shard2": dict(name="DSVGP d=20 N=1M M=500 p=5 B=2048 (one rank's share of C4 at 2 GPUs)", d=20, N=1_000_000, M=500,
                     p=5, B=2048),
    # BASELINE config 5: CIQ whitening + NGD (no dataset / batch size given there; N and B chosen like C4's per-GPU shard)
    "c5": dict(name="CIQ-DSVGP d=50 N=100k M=1024 p=5 B=512", d=50, N=100_000, M=1024, p=5, B=512, ciq=True),
}
PEAK_F64_MFMA_TFLOPS = 78.6     # MI355X FP64 matrix peak (spec)


def synthetic_data(N, d, device):
    g = torch.Generator(device=device).manual_seed(0)
    X = torch.rand(N, d, device=device, generator=g)
    sq = (X * X).sum(1)
    Y = torch.cat([torch.sin(2 * math.pi * sq)[:, None], 4 * math.pi * torch.cos(2 * math.pi * sq)[:, None] * X], 1)
    return X.contiguous(), Y.contiguous()


def cpu_baseline(cfg, budget_s=30.0):
    """Reference-faithful CPU step (oracle/train_ref.py) on a bounded sample: full C-config steps, as many
    as fit the budget after one untimed warm-up step."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import train_ref
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    n_sample = min(cfg["N"], 50_000)           # DataLoader over a bounded slice of the dataset (same batch shape)
    st = train_ref.RefTrainer(n_sample, cfg["d"], cfg["M"], cfg["p"], cfg["B"], num_data_override=(cfg["d"] + 1) * cfg["N"])
    t0 = time.time()
    st.step()
    first = time.time() - t0
    steps, t_acc = 0, 0.0
    while steps < 1 or (t_acc + first * 0.9 < budget_s and steps < 10):
        t0 = time.time()
        st.step()
        t_acc += time.time() - t0
        steps += 1
    return dict(value=steps / t_acc, unit="steps/s", cores=cores, kind="port",
                sample="%d timed full-size steps (B=%d, M'=%d) after 1 warm-up, dataset slice of %d rows; %.1f s/step"
                       % (steps, cfg["B"], cfg["M"] * (cfg["p"] + 1), n_sample, t_acc / steps))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--trsm-nb", type=int, default=int(os.environ.get("DSVGP_TRSM_NB", "4096")))
    ap.add_argument("--no-overlap", action="store_true", help="diagnostics: no side stream under the Cholesky chain")
    ap.add_argument("--no-fused-inverse", action="store_true", help="diagnostics: potrf + trtri recursion instead")
    ap.add_argument("--emulate-world", type=int, default=1,
                    help="diagnostics (N = 1): rank 0's work of a W-rank job, collectives skipped")
    ap.add_argument("--no-global-gram", action="store_true",
                    help="diagnostics (N > 1): all-reduce the variational gradients instead of [G ; b^T], replicated Cholesky backward")
    ap.add_argument("--no-lib-gemm", action="store_true", help="diagnostics: the dense K_ZX-bar product on gemm.hip")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                             % (args.gpus, args.gpus))
    # rehearsal on a one-GPU box: DSVGP_REHEARSE_GLOO=1 runs all ranks on cuda:0 over gloo (exercises the sharded path,
    # not a measurement); the driver's multi-GPU runs use one GPU per rank over RCCL
    rehearse = os.environ.get("DSVGP_REHEARSE_GLOO") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import dsvgp_amd
    d, N, M, p, B = cfg["d"], cfg["N"], cfg["M"], cfg["p"], cfg["B"]
    X, Y = synthetic_data(N, d, device)
    loop = dsvgp_amd.setup_training(None, num_inducing=M, num_directions=p, minibatch_size=B, minibatch_dim=p,
                                    num_epochs=1, learning_rate_hypers=0.01, inducing_data_initialization=True,
                                    seed=0, tensors=(X, Y), use_ciq=bool(cfg.get("ciq")))
    eng = loop.model.engine
    eng.trsm_nb = args.trsm_nb
    if args.no_overlap:
        eng.overlap = False        # (default None: automatic by problem size)
    eng.fused_inverse = not args.no_fused_inverse
    eng.lib_dense_gemm = not args.no_lib_gemm
    eng.global_gram = not args.no_global_gram
    if args.emulate_world > 1 and world == 1:
        # diagnostics on a one-GPU box: run rank 0's share of the work of a W-rank job with the collectives skipped
        # (use with --config c4shardW; not a measurement of the job, only of one rank's kernels)
        class _NoComm:
            rank, world = 0, args.emulate_world

            def all_reduce_async(self, t):
                class _H:
                    def wait(self):
                        pass
                return _H()
        eng.collective = _NoComm()
    perm = loop.epoch_permutation()
    nbatches = N // B

    def batch(k):
        k = k % nbatches
        return perm[k * B:(k + 1) * B]

    for k in range(args.warmup):
        loop.step(batch(k))
    eng.record_events = True
    eng.events = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        loss, _, _ = loop.step(batch(args.warmup + k))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    eng.record_events = False
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    final_loss = float(loss.item())

    # dominant kernel: the fp64 MFMA GEMM of the forward panel solve A = L^-1 K_ZX (one launch when nb >= M')
    Mp = M * (p + 1)
    Bp_local = (B // world) * (p + 1)
    durs = [s.elapsed_time(e) * 1e-3 for (name, s, e) in eng.events if name == "solve_fwd"]
    roof = None
    if durs:
        avg = sum(durs) / len(durs)
        flops = float(Mp) * Mp * Bp_local           # F_trsm = M'^2 B' per solve
        ach = flops / avg / 1e12
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in
        # separate runs, gfx950 corrections applied by tools/summarize_pmc.py); only valid for the 1-GPU C4 shape
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
        if world == 1 and args.config == "c4" and args.trsm_nb >= Mp and os.path.exists(pmc):
            cands = [k for k in json.load(open(pmc))["kernels"] if k["kernel"].startswith("gemm64_kernel<float>")]
            if cands:   # the forward solve is the largest launch of that instantiation
                traffic = max(cands, key=lambda k: k["hbm_bytes_per_launch"])["hbm_bytes_per_launch"]
        roof = dict(bound="mfma", kernel="gemm64_kernel<float> (panel solve A = L^-1 K_ZX: lower-triangular fp64 inverse x fp32 K_ZX on v_mfma_f64_16x16x4, fp32 result)",
                    achieved=ach, peak=PEAK_F64_MFMA_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_F64_MFMA_TFLOPS,
                    traffic=traffic, launches=len(durs), avg_ms=avg * 1e3, flops_per_launch=flops)

    if rank == 0:
        out = {
            "metric": "ELBO steps/sec, DSVGP d=20 N=1M M=500 p=5" if args.config == "c4" else "ELBO steps/sec, " + cfg["name"],
            "value": args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32 (f64 Cholesky/solves)", "data": "synthetic",
            "config": {"workload": cfg["name"], "global_batch": B, "per_gpu_batch": B // world, "M_prime": Mp,
                       "parallelism": "dp%d rows" % world, "trsm_nb": args.trsm_nb, "final_loss": final_loss},
            "roofline": roof,
        }
        if cfg.get("ciq"):
            out["config"]["ciq"] = dict(eng.ciq_stats)
        if world == 1 and not args.no_cpu_baseline and not cfg.get("ciq"):
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
