#!/usr/bin/env python3
"""Benchmark of the DSVGP ELBO training step (BASELINE.json metric) on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: this process starts ``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a
child BEFORE anything touches the GPU, relays its output (rank 0 prints the JSON line) and exits with its status; under a
launcher (WORLD_SIZE set) it is one rank.  Any other WORLD_SIZE / --gpus mismatch is an error.

A "step" is one iteration of ``directional_vi.train_gp``'s inner loop (reference
directionalvi/directional_vi.py:229-254): minibatch gather, fused ELBO forward + backward, both Adam
steps and both LR-scheduler steps.  Workload = BASELINE config 4: d=20, N=1M, M=500, p=5, global
minibatch 4096 (sharded by rows over the ranks, packed-triangle RCCL all-reduce per step: strong scaling).
Synthetic data (X ~ U[0,1]^d, y=[f, grad f], f=sin(2 pi |x|^2), reference tests/testfun.py) is resident in
HBM before the timed region.  Rank 0 prints ONE JSON line with
  roofline           the dominant kernel (fp64 MFMA panel-solve GEMM), timed live with HIP events on its launch stream;
  roofline_assembly  the HBM-bound kernel assembly forward (K_ZX) and backward (K_ZX-bar), same method (north_star);
  cpu_baseline       the oracle's reference-op-sequence training step on this host's cores (N = 1 only).
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    "c4": dict(name="DSVGP d=20 N=1M M=500 p=5 B=4096", d=20, N=1_000_000, M=500, p=5, B=4096),
    "c2": dict(name="DSVGP d=5 N=10k M=200 p=2 B=512", d=5, N=10_000, M=200, p=2, B=512),
    # BASELINE config 3: full-gradient SVGP (grad_svgp harness), p = d, M' = M (d + 1) = 3300
    "c3": dict(name="GradSVGP d=10 N=50k M=300 (M'=3300) B=512", d=10, N=50_000, M=300, p=10, B=512, grad=True),
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
PEAK_HBM_TBPS = 8.0             # HBM3E (MI355X_MICROARCH.md)


def _self_launch(args):
    """--gpus N > 1 from a plain invocation: one child launcher, N ranks, before any GPU call in this process.
    The rendezvous port is picked by the launcher itself (``--standalone``: c10d store on an ephemeral port of 127.0.0.1),
    so two benches starting on one node cannot race for a port probed here."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def synthetic_data(N, d, device):
    import torch
    g = torch.Generator(device=device).manual_seed(0)
    X = torch.rand(N, d, device=device, generator=g)
    sq = (X * X).sum(1)
    Y = torch.cat([torch.sin(2 * math.pi * sq)[:, None], 4 * math.pi * torch.cos(2 * math.pi * sq)[:, None] * X], 1)
    return X.contiguous(), Y.contiguous()


def usable_cpus():
    """CPUs this process may actually run on: the affinity mask and the cgroup CPU quota, not the host's core count (a GPU box
    shows 256 logical CPUs to a container limited to a share of them; 256 threads on that share thrash)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()])):
        try:
            quota, period = parse(open(path).read())
            if quota != "max" and int(quota) > 0:
                n = min(n, max(1, int(math.ceil(int(quota) / int(period)))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(cfg, budget_s=150.0, min_steps=3, fp64=False):
    """The oracle's reference-op-sequence training step (oracle/train_ref.py: DataLoader over the FULL dataset, the four
    kernel assemblies in the reference's matmul / gather / shuffle form, fp64 Cholesky + 2 solves, autograd backward, two
    Adam steps) on this host: 1 untimed warm-up step, then >= 3 timed full-size steps (more while the budget lasts)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import train_ref
    cores = usable_cpus()
    torch.set_num_threads(cores)
    st = train_ref.RefTrainer(cfg["N"], cfg["d"], cfg["M"], cfg["p"], cfg["B"], full_gradient=bool(cfg.get("grad")),
                              dtype=torch.float64 if fp64 else torch.float32)
    t0 = time.time()
    st.step()
    first = time.time() - t0
    st.assembly_seconds = 0.0
    steps, t_acc = 0, 0.0
    while steps < min_steps or (t_acc + first * 0.9 < budget_s and steps < 10):
        t0 = time.time()
        st.step()
        t_acc += time.time() - t0
        steps += 1
        if steps >= 1 and t_acc > 2.0 * budget_s:          # a very slow host: report what was timed, say so in `sample`
            break
    return dict(value=steps / t_acc, unit="steps/s", cores=cores, threads=torch.get_num_threads(),
                host_logical_cpus=os.cpu_count(), kind="port",
                assembly="reference-sequence",
                assembly_fwd_s_per_step=st.assembly_seconds / steps,
                sample="%d timed full-size steps (B=%d, M'=%d, DataLoader over all N=%d rows) after 1 warm-up step "
                       "(%.1f s); %.1f s/step, of which %.2f s in the forward of the 4 kernel assemblies"
                       % (steps, cfg["B"], cfg["M"] * (cfg["p"] + 1), cfg["N"], first, t_acc / steps,
                          st.assembly_seconds / steps))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--trsm-nb", type=int, default=int(os.environ.get("DSVGP_TRSM_NB", "0")),
                    help="panel width of the triangular solve; 0 = the engine's automatic regime (explicit inverse up to M' = 8192)")
    ap.add_argument("--no-overlap", action="store_true", help="diagnostics: no side stream under the Cholesky chain")
    ap.add_argument("--no-fused-inverse", action="store_true", help="diagnostics: potrf + trtri recursion instead")
    ap.add_argument("--emulate-world", type=int, default=1,
                    help="diagnostics (N = 1): rank 0's work of a W-rank job, collectives skipped")
    ap.add_argument("--no-global-gram", action="store_true",
                    help="diagnostics (N > 1): all-reduce the variational gradients instead of [G ; b^T], replicated Cholesky backward")
    ap.add_argument("--lib-gemm", action="store_true",
                    help="diagnostics: the dense K_ZX-bar product through rocBLAS sgemm instead of the hand-written kernel")
    ap.add_argument("--no-pack-reduce", action="store_true", help="diagnostics (N > 1): dense instead of packed-triangle all-reduce")
    ap.add_argument("--fp64", action="store_true",
                    help="the reference's experiment-script mode (torch.set_default_dtype(torch.float64), "
                         "experiments/synthetic/exp_script.py:56): data, model and every kernel in double precision")
    ap.add_argument("--graph", default="off", choices=["on", "off"],
                    help="HIP-graph replay of the step (one rank, ELBO fast path); measured no faster than eager at C2, see DESIGN.md")
    ap.add_argument("--dp-algo", default=os.environ.get("DSVGP_DP_ALGO", "allreduce"), choices=["allreduce", "rs_ag"],
                    help="(N > 1) collective of the large operand: RCCL all-reduce, or reduce-scatter + all-gather")
    ap.add_argument("--split-bf16", action="store_true",
                    help="OPT-IN: the Gram product and the dense K_ZX-bar product as bf16 x 3 split products on the bf16 matrix pipe "
                         "(six bf16 MFMA products per fp32 product, fp32 accumulation; csrc/gemm3b.hip) instead of fp32 MFMA")
    ap.add_argument("--no-other-configs", action="store_true", help="c4 on one GPU: skip the C2 / C3 / C5 / float64 windows")
    ap.add_argument("--no-extras", action="store_true",
                    help="c4 on one GPU: skip the repeated windows, the reporting-step timing and the other BASELINE configurations")
    ap.add_argument("--eval-only", action="store_true", help="tools: only the eval_gp throughput figure of --config (other_configs.eval_c4)")
    ap.add_argument("--event-every", type=int, default=0,
                    help="HIP events (dominant-kernel / assembly timings) on every K-th timed step; 0 = every step, except on the "
                         "sub-millisecond configuration c2 where six event records are 5 %% of the step: every 8th there")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.event_every <= 0:
        args.event_every = 8 if args.config == "c2" else 1

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if args.gpus > 1:
            sys.exit(_self_launch(args))
        world, rank, local_rank = 1, 0, 0
    else:
        world, rank, local_rank = int(env_world), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            raise SystemExit("bench.py: WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # before the HIP runtime comes up in this process

    import torch
    import torch.distributed as dist
    # rehearsal on a one-GPU box: DSVGP_REHEARSE_GLOO=1 runs all ranks on cuda:0 over gloo (exercises the sharded path,
    # not a measurement); the driver's multi-GPU runs use one GPU per rank over RCCL
    if args.fp64:
        torch.set_default_dtype(torch.float64)
    rehearse = os.environ.get("DSVGP_REHEARSE_GLOO") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import dsvgp_amd

    def build(cfg, fp64):
        """dataset + training loop of one configuration, engine switches applied; returns (loop, engine, batch(k))"""
        d, N, M, p, B = cfg["d"], cfg["N"], cfg["M"], cfg["p"], cfg["B"]
        prev = torch.get_default_dtype()
        if fp64:
            torch.set_default_dtype(torch.float64)
        try:
            X, Y = synthetic_data(N, d, device)
            if fp64:
                X, Y = X.double(), Y.double()
            if cfg.get("grad"):
                loop = dsvgp_amd.grad_svgp.setup_training(None, d, num_inducing=M, minibatch_size=B, num_epochs=1,
                                                          learning_rate_hypers=0.01, seed=0, tensors=(X, Y))
            else:
                loop = dsvgp_amd.setup_training(None, num_inducing=M, num_directions=p, minibatch_size=B, minibatch_dim=p,
                                                num_epochs=1, learning_rate_hypers=0.01, inducing_data_initialization=True,
                                                seed=0, tensors=(X, Y), use_ciq=bool(cfg.get("ciq")))
        finally:
            torch.set_default_dtype(prev)
        loop.graph = args.graph == "on"
        eng = loop.model.engine
        if args.trsm_nb > 0:
            eng.trsm_nb = args.trsm_nb
        if args.no_overlap:
            eng.overlap = False        # (default None: automatic by problem size)
        if os.environ.get("DSVGP_OVERLAP") == "1":
            eng.overlap = True         # diagnostics: the side stream also below M' = 2048
        eng.fused_inverse = not args.no_fused_inverse
        if args.split_bf16:
            eng.split_bf16 = True
        eng.lib_dense_gemm = bool(args.lib_gemm)
        eng.global_gram = not args.no_global_gram
        eng.pack_reduce = not args.no_pack_reduce
        if loop.dp is not None:
            loop.dp.algo = args.dp_algo
            # replica check on by default under the bench (every 64 steps; DSVGP_DP_CHECK=N / 0 overrides): the replicas' L_S, m and Adam
            # moments are never re-broadcast, so the first run on real multi-GPU hardware reports whether they stayed bitwise equal
            if os.environ.get("DSVGP_DP_CHECK") is None:
                loop.dp.check_every = 64
        perm = loop.epoch_permutation()
        nbatches = N // B
        return loop, eng, (lambda k: perm[(k % nbatches) * B:(k % nbatches + 1) * B])

    def window(loop, batch, first, steps):
        """`steps` optimisation steps bracketed by barrier + synchronize on both sides; returns (seconds, last loss)"""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            loss, _, _ = loop.step(batch(first + k))
        loop.finish()                    # (graph replay: the status of the last replayed step is read here)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, loss

    def events_on(eng, every):
        eng.record_events = True
        eng.events = []
        if hasattr(eng, "c_step_timed"):
            eng.c_step_timed = []
            eng.record_every, eng._rec_count = every, 0

    def event_durations(eng, names):
        has_dur = hasattr(eng, "event_durations")
        return {nm: (eng.event_durations(nm) if has_dur else [s_.elapsed_time(e_) * 1e-3 for (n2, s_, e_) in eng.events if n2 == nm])
                for nm in names}

    d, N, M, p, B = cfg["d"], cfg["N"], cfg["M"], cfg["p"], cfg["B"]
    loop, eng, batch = build(cfg, args.fp64)
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

            def all_gather_async(self, out, inp):       # (every block = rank 0's: finite stand-ins for the other ranks' data)
                out.view(self.world, -1).copy_(inp.reshape(1, -1).expand(self.world, -1))
                return self.all_reduce_async(None)
        eng.collective = _NoComm()
        # the stand-in blocks make the gradients meaningless: freeze the parameters (the kernels run unchanged)
        for opt, sch in ((loop.variational_optimizer, loop.variational_scheduler),
                         (loop.hyperparameter_optimizer, loop.hyperparameter_scheduler)):
            for grp in opt.param_groups:
                grp["lr"] = 0.0
            if hasattr(sch, "base_lrs"):
                sch.base_lrs = [0.0] * len(sch.base_lrs)

    for k in range(args.warmup):
        loop.step(batch(k))
    events_on(eng, args.event_every)
    if (world > 1 or args.emulate_world > 1) and hasattr(eng, "dp_host_trace"):
        eng.dp_host_trace = []           # host time stamps of the five C calls + collectives of every rank step in the window
    elapsed, loss = window(loop, batch, args.warmup, args.steps)
    dp_trace, replica_check = None, None
    if getattr(eng, "dp_host_trace", None):
        tr_ = eng.dp_host_trace
        names_ = ("phase0_front_gram", "issue_allreduce_G", "phase1_q_columns", "issue_allgather_q", "wait_allreduce_G", "phase2_variational_lbar_rows",
                  "issue_allgather_lbar", "wait_allgather_q", "phase3_dense_kernel_bwd", "wait_allgather_lbar", "phase4_chol_backward_tail")
        n_ = len(tr_)
        dp_trace = dict(steps=n_, unit="us of host time per rank step (rank 0), mean over the timed window",
                        total=1e6 * sum(t_[-1] - t_[0] for t_ in tr_) / n_,
                        **{nm_: 1e6 * sum(t_[i_ + 1] - t_[i_] for t_ in tr_) / n_ for i_, nm_ in enumerate(names_)})
    if hasattr(eng, "dp_host_trace"):
        eng.dp_host_trace = None
    if loop.dp is not None:
        if loop.dp.check_every > 0:
            loop._check_replicas(loop.dp, force=True)       # one more check at the end of the window, whatever the step count
        replica_check = dict(every=loop.dp.check_every, checks=loop.dp.checks, divergences=loop.dp.divergences,
                             note="bitwise checksum of every parameter across the ranks after the Adam step (parallel.DataParallel.check_replicas); "
                                  "a divergence re-broadcasts rank 0's parameters and Adam moments")
    if hasattr(eng, "record_every"):
        eng.record_every = 1             # (the untimed passes below: every step)
    if loop._graphs:
        # the timed steps were graph replays (no HIP events inside a graph): the per-kernel timings of the roofline entries
        # come from three more, untimed, eager steps of the same loop
        loop.graph = False
        eng.events = []
        if hasattr(eng, "c_step_timed"):
            eng.c_step_timed = []
        for k in range(3):
            loop.step(batch(args.warmup + args.steps + k))
        torch.cuda.synchronize()
    # the kernel assembly forward alone on the GPU (it shares the CUs with the Cholesky chain in the step when M' >= 2048):
    # three more untimed steps with the side stream off, HIP events around the same launch
    EVENT_NAMES = ("solve_fwd", "assemble_fwd", "assemble_bwd", "gram", "dense", "early_reduce_wait", "final_reduce", "ciq_stacked_backward")
    durations = event_durations(eng, EVENT_NAMES)
    iso_fwd = iso_bwd = None
    if not cfg.get("ciq") and world == 1 and not args.fp64:
        saved = eng.overlap
        eng.overlap, eng.events = False, []
        if hasattr(eng, "c_step_timed"):
            eng.c_step_timed = []
        for k in range(3):
            loop.step(batch(args.warmup + args.steps + 3 + k))
        torch.cuda.synchronize()
        iso = event_durations(eng, ("assemble_fwd", "assemble_bwd"))
        durs = iso["assemble_fwd"]
        iso_fwd = sum(durs) / len(durs) if durs else None
        durs = iso["assemble_bwd"]
        iso_bwd = sum(durs) / len(durs) if durs else None
        eng.overlap = saved
    eng.record_events = False
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    final_loss = float(loss.item())

    # ---- the same window twice more (box-to-box and run-to-run spread of a 0.3 s timed region is +-3 %): min / median / max of the
    # three, `steps` / `ms_per_step` stay those of the FIRST window
    repeat_ms = None
    extras = args.config == "c4" and world == 1 and args.emulate_world == 1 and not args.fp64 and not args.no_extras
    if extras:
        ws_ = [elapsed]
        for r in range(2):
            e_r, _ = window(loop, batch, args.warmup + args.steps * (r + 1) + 6, args.steps)
            ws_.append(e_r)
        ws_ = sorted(1e3 * w / args.steps for w in ws_)
        repeat_ms = [ws_[0], ws_[1], ws_[2]]
    # ---- second figure (opt-in mode, never the headline): the same window with the two big fp32 products as bf16 x 3 split products
    split_fig = None
    if extras and hasattr(eng, "split_bf16") and not eng.split_bf16:
      try:
        eng.split_bf16 = True
        for k in range(3):
            loop.step(batch(args.warmup + 3 * args.steps + 20 + k))
        e_s, loss_s = window(loop, batch, args.warmup + 3 * args.steps + 23, args.steps)
        eng.split_bf16 = False
        split_fig = dict(ms_per_step=1e3 * e_s / args.steps, steps=args.steps, final_loss=float(loss_s.item()),
                         dtype="f32 model; the Gram product and the dense K_ZX-bar product as six bf16 MFMA products per fp32 product "
                               "(operands split into three bf16 planes, fp32 accumulation, v_mfma_f32_32x32x16_bf16); f64 Cholesky/solves",
                         note="opt-in (`--split-bf16` / DSVGP_SPLIT_BF16=1); passes every reference-text case at the fp32 path's "
                              "tolerances, worst error per configuration equal to the fp32 path's, individual scalar gradients up to "
                              "3-27x further out (<= 1.2e-5): tests/test_gpu_reftext.py::test_split_bf16_step_*")
      except Exception as ex:                # (an extra must never take the headline line down)
        eng.split_bf16 = False
        split_fig = dict(error="%s: %s" % (type(ex).__name__, ex))
    # ---- the reference's every-50th-step report (directional_vi.py:255-260: loss.item(), nll of the function values from
    # output.mean / output.variance of that forward pass): one reporting step costs a host synchronisation (the step pipeline
    # drains) plus the value-row variances; measured over 5 reporting steps in a row, 1/50 of the extra goes into ms_per_step
    report_extra_ms = None
    if extras and hasattr(eng, "value_variances"):
        q_ = p + 1

        def report_step(k):
            l_, out_, yb_ = loop.step(batch(args.warmup + 3 * args.steps + 6 + k), need_variance="values")
            means_ = out_.mean[::q_]
            stds_ = out_.value_variance.sqrt()
            nll_ = -torch.distributions.Normal(means_, stds_).log_prob(yb_[::q_]).mean()
            return "loss: %s, nll: %s" % (l_.item(), nll_.item())

        try:
            report_step(0)               # (untimed: first use of the value-row buffers and of torch.distributions' kernels)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(5):
                report_step(1 + k)
            torch.cuda.synchronize()
            report_extra_ms = max(0.0, 1e3 * (time.perf_counter() - t0) / 5 - 1e3 * elapsed / args.steps)
        except Exception:                    # (an extra must never take the headline line down)
            report_extra_ms = None

    # N > 1: what the collectives this design could use cost on THIS node, measured after the timed region (all ranks take part;
    # priced for the next design step: sharding the replicated M' x M' products would add two 72 MB all-gathers per step)
    coll_probe = None
    if world > 1 and not rehearse and os.environ.get("DSVGP_BENCH_COLL_PROBE", "1") == "1":
        coll_probe = {}
        Mp_ = M * (p + 1)
        for name, fn, numel, dt in (("allreduce_packed_G_ms", "ar", Mp_ * (Mp_ + 1) // 2 + Mp_, torch.float32),
                                    ("allgather_MxM_f64_ms", "ag", Mp_ * Mp_ // world * world, torch.float64),
                                    ("allgather_MxM_f32_ms", "ag", Mp_ * Mp_ // world * world, torch.float32)):
            try:
                buf = torch.zeros(numel, dtype=dt, device=device)
                shard = buf[:numel // world].clone()
                for it in range(13):
                    if it == 3:
                        torch.cuda.synchronize()
                        dist.barrier()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                    if fn == "ar":
                        dist.all_reduce(buf)
                    else:
                        dist.all_gather_into_tensor(buf, shard)
                e1.record()
                torch.cuda.synchronize()
                coll_probe[name] = e0.elapsed_time(e1) / 10.0
            except Exception as ex:      # (a backend without the collective: report, do not fail the bench line)
                coll_probe[name] = "unavailable: %s" % type(ex).__name__
    def rooflines(cfg, eng, durations, iso_fwd, fp64, cfg_name, headline, iso_bwd=None):
        """(roofline of the dominant kernel, assembly roofline) of one configuration from its HIP-event durations"""
        d, N, M, p, B = cfg["d"], cfg["N"], cfg["M"], cfg["p"], cfg["B"]
        # dominant kernel: the fp64 MFMA GEMM of the forward panel solve A = L^-1 K_ZX (one launch when nb >= M')
        Mp = M * (p + 1)
        B_local = B // world
        Bp_local = B_local * (p + 1)

        def avg(name):
            durs = durations.get(name, [])
            return (sum(durs) / len(durs), len(durs)) if durs else (None, 0)

        roof = None
        t_solve, n_solve = avg("solve_fwd")
        if t_solve:
            flops = float(Mp) * Mp * Bp_local           # F_trsm = M'^2 B' per solve
            ach = flops / t_solve / 1e12
            # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in
            # separate runs, gfx950 corrections applied by tools/summarize_pmc.py); only valid for the 1-GPU C4 shape
            traffic = traffic_src = pmc_busy = None
            if world == 1 and cfg_name == "c4" and eng.trsm_nb >= Mp and not fp64 and headline:
                import glob
                # NOT live: both figures are read from the committed rocprofv3 PMC summaries of this same command (profiles/), newest round first
                # (the kernel that runs the forward solve NOW decides which committed profile applies: the pipelined wide kernel of gemm64.hip first)
                families = ("gemm64p_kernel", "gemm64w_kernel<float", "gemm64_kernel<float")
                for fam in families:
                    for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")), reverse=True):
                        cands = [k for k in json.load(open(pmc))["kernels"] if k["kernel"].startswith(fam)]
                        if cands:   # the forward solve is the largest launch of that instantiation
                            traffic = max(cands, key=lambda k: k["hbm_bytes_per_launch"])["hbm_bytes_per_launch"]
                            traffic_src = "profiles/" + os.path.basename(pmc)
                            break
                    if traffic is not None:
                        break
                for fam in families:
                    for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_step_mfma_busy.txt")), reverse=True):
                        best = None
                        lines = open(pmc).read().splitlines()
                        for i, ln in enumerate(lines[:-1]):
                            if ln.startswith(fam) and " grid=" in ln and "mfma_busy=" in lines[i + 1]:
                                vals = dict(kv.split("=") for kv in lines[i + 1].replace(" GHz", "").split() if "=" in kv)
                                ms = float(ln.split("avg_ms=")[1])
                                if best is None or ms > best[2]:            # (the forward solve is the longest launch of its family)
                                    best = (ln, vals, ms)
                        if best:
                            pmc_busy = dict(kernel=best[0].split(" grid=")[0], mfma_busy=float(best[1]["mfma_busy"]),
                                            wait_any_per_wave=float(best[1]["wait_any/wave"]),
                                            wait_inst_any_per_wave=float(best[1]["wait_inst_any/wave"]), clock_ghz=float(best[1]["clk"]),
                                            profiled_avg_ms=best[2], source="profiles/" + os.path.basename(pmc),
                                            note="rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES) over `python3 bench.py "
                                                 "--config c4`, forward-solve launches only; committed profile, not collected by this run")
                            break
                    if pmc_busy is not None:
                        break
            wide = Mp >= 64 and (Mp + 63) // 64 * ((Bp_local + 63) // 64) >= 8192          # (gemm64.hip: 64 x 192 / 64 x 128 tiles from 8192 tiles of 64 x 64 up)
            roof = dict(bound="mfma", kernel=("%s (panel solve A = L^-1 K_ZX: lower-triangular fp64 inverse x fp64 K_ZX on v_mfma_f64_16x16x4)"
                                               % ("gemm64p_kernel<double, 128> (64 x 128 tiles, LDS-DMA stages)" if wide else "gemm64_kernel<double>") if fp64 else
                                               "%s (panel solve A = L^-1 K_ZX: lower-triangular fp64 inverse x fp32 K_ZX on v_mfma_f64_16x16x4, fp32 result)"
                                               % ("gemm64p_kernel<float, 192> (64 x 192 tiles, LDS-DMA stages)" if wide else "gemm64l_kernel (64 x 64 tiles, LDS-DMA stages)")),
                        achieved=ach, peak=PEAK_F64_MFMA_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_F64_MFMA_TFLOPS,
                        traffic=traffic, traffic_source=(traffic_src + " (committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes, not live)") if traffic_src else None,
                        mfma_utilisation=pmc_busy, launches=n_solve, avg_ms=t_solve * 1e3, flops_per_launch=flops)
            if rank == 0 and headline:
                # what THIS card sustains on the same instruction with no memory traffic (40 ms of back-to-back MFMAs after the
                # timed region): `peak` above is the data-sheet figure at 2.4 GHz, which no MI355X of this pool holds under matrix load
                sus = dsvgp_amd._ops.mfma_rate(dsvgp_amd._ops.Context.get(device), True, 40)
                roof["sustained"] = dict(measured_mfma_only=sus, unit="TFLOP/s", frac_of_sustained=ach / sus,
                                         note="v_mfma_f64_16x16x4_f64 from registers on all CUs (dsvgp_mfma_rate)")
                if not fp64:
                    # the same for the fp32 shape of the step's other two large products (dense K_ZX-bar, Gram: gemm32.hip), for the reader of
                    # profiles/*kernel_stats_c4.txt: their 125 / 108 TF are to be read against this figure, not against 157.3
                    roof["sustained"]["fp32_mfma_only"] = dsvgp_amd._ops.mfma_rate(dsvgp_amd._ops.Context.get(device), False, 40)
                    # ... and the probe in the hardware guide's own form (ONE wave per SIMD; MI355X_MICROARCH.md: 155 TF fp32), with the rate of
                    # its first ~2 ms launch from a cooler card and the in-kernel clock of its last launch: the gap between the data-sheet
                    # peak (2.4 GHz) and what a 40 ms run holds is the clock, not the instruction stream
                    try:
                        ctx_ = dsvgp_amd._ops.Context.get(device)
                        probes = {}
                        for nm_, dbl_, one_ in (("fp32_one_wave_per_simd", False, True), ("fp32_four_waves_per_simd", False, False),
                                                ("fp64_one_wave_per_simd", True, True), ("fp64_four_waves_per_simd", True, False)):
                            r_, b_, c_ = dsvgp_amd._ops.mfma_rate2(ctx_, dbl_, one_, 40)
                            peak_ = PEAK_F64_MFMA_TFLOPS if dbl_ else 157.3
                            probes[nm_] = dict(sustained_tflops=r_, first_launch_tflops=b_, clock_ghz_last_launch=c_,
                                               tflops_at_2p4_ghz=(r_ * 2.4 / c_) if c_ > 0 else None,
                                               frac_of_peak_at_that_clock=(r_ / (peak_ * c_ / 2.4)) if c_ > 0 else None)
                        roof["sustained"]["probes"] = probes
                    except Exception as ex:            # (an extra must never take the headline line down)
                        roof["sustained"]["probes"] = dict(error="%s: %s" % (type(ex).__name__, ex))

        # the step's two fp32 [M', B'] products, HIP events of the one-call step (csrc/step.hip, dsvgp_elbo_step_timings5): the dense product
        # K_ZX-bar = [Q' | a][A ; mu_bar^T] (2 M' (M'+1) B' flop) and the Gram product [tril(G) ; b^T] = tril([A ; mu_bar^T] A^T) (its lower
        # triangle: (M'+1) M' B' flop), both on v_mfma_f32_32x32x2_f32 (gemm32.hip); peak 157.3 TF (MI355X_MICROARCH.md)
        roof_dense = roof_gram = None
        t_dense, n_dense = avg("dense")
        t_gram, n_gram = avg("gram")
        if t_dense and not fp64:
            fl = 2.0 * Mp * (Mp + 1) * Bp_local
            roof_dense = dict(bound="mfma", kernel="gemm32_dma_kernel<32,true,false,32> (dense K_ZX-bar = [Q' | a][A ; mu_bar^T], fp32 MFMA, LDS-DMA stages)",
                              achieved=fl / t_dense / 1e12, peak=157.3, unit="TFLOP/s", frac=fl / t_dense / 1e12 / 157.3, launches=n_dense,
                              avg_ms=t_dense * 1e3, flops_per_launch=fl,
                              note="HIP events around the launch on its stream; includes the launch's own clear of nothing (no split-K)")
        if t_gram and not fp64:
            fl = float(Mp + 1) * Mp * Bp_local
            roof_gram = dict(bound="mfma", kernel="gemm32_dma_kernel<32,true,true,32> (Gram [tril(G) ; b^T] = tril([A ; mu_bar^T] A^T), fp32 MFMA, split-K, fp32 atomics)",
                             achieved=fl / t_gram / 1e12, peak=157.3, unit="TFLOP/s", frac=fl / t_gram / 1e12 / 157.3, launches=n_gram,
                             avg_ms=t_gram * 1e3, flops_per_launch=fl,
                             note="lower-triangle flops only; the event pair also spans the clear of the split-K target when the launcher makes one")
        if roof is not None:
            roof["dense"], roof["gram"] = roof_dense, roof_gram

        t_ciq, n_ciq = avg("ciq_stacked_backward")
        if cfg.get("ciq") and t_ciq:
            # CIQ (BASELINE config 5): the largest launch of the step is the stacked backward product of sqrt_inv_matmul,
            # K_ZZ-bar = -sym stack_i(U_i)^T stack_i(Z_i): [M', depth] x [depth, M'] in fp32 on v_mfma_f32_32x32x2_f32
            depth = float(eng.ciq_stats.get("stacked_depth", 0))
            flops = 2.0 * Mp * Mp * depth
            roof = dict(bound="mfma", kernel="gemm32 (stacked backward product of the CIQ step: [M', depth]^T x [depth, M'], fp32 MFMA)",
                        achieved=flops / t_ciq / 1e12, peak=157.3, unit="TFLOP/s", frac=flops / t_ciq / 1e12 / 157.3, traffic=None,
                        launches=n_ciq, avg_ms=t_ciq * 1e3, flops_per_launch=flops, depth=depth)
        # kernel assembly (north_star: HBM GB/s of the assembly): algorithmic bytes per launch =
        # 4 [M' B' + (M + B) d (p + 1)] -- write the block matrix once (forward) / read its gradient once (backward) plus the
        # points and directions -- over the live HIP-event duration on the stream the kernel was queued on
        roof_asm = None
        t_f, n_f = avg("assemble_fwd")
        t_b, n_b = avg("assemble_bwd")
        if t_f and t_b and not cfg.get("ciq"):
            nbytes = 4.0 * (float(Mp) * Bp_local + (M + B_local) * d * (p + 1))
            # (the training loop states its one-hot directions as an index list: the canonical-direction kernels where they take (d, p))
            canon = "_canon" if dsvgp_amd._ops.canon_supported(d, p) and not cfg.get("dfree_values") else ""
            roof_asm = dict(bound="hbm", peak=PEAK_HBM_TBPS, unit="TB/s", bytes_per_launch=nbytes,
                            forward=dict(kernel="kernel_fwd%s (K_ZX, %d x %d fp32, interleaved block layout)" % (canon, Mp, Bp_local),
                                         avg_ms=t_f * 1e3, launches=n_f, achieved=nbytes / t_f / 1e12,
                                         frac=nbytes / t_f / 1e12 / PEAK_HBM_TBPS,
                                         note="queued on the side stream under the Cholesky chain when M' >= 2048: shares the CUs",
                                         alone_avg_ms=iso_fwd * 1e3 if iso_fwd else None,
                                         alone_achieved=nbytes / iso_fwd / 1e12 if iso_fwd else None,
                                         alone_frac=nbytes / iso_fwd / 1e12 / PEAK_HBM_TBPS if iso_fwd else None),
                            backward=dict(kernel="kernel_bwd%s + kernel_bwd_points (reads K_ZX-bar once -> dZ, dV, d ell, d s)" % canon,
                                          avg_ms=t_b * 1e3,
                                          launches=n_b, achieved=nbytes / t_b / 1e12, frac=nbytes / t_b / 1e12 / PEAK_HBM_TBPS,
                                          note="queued on the side stream beside the Cholesky backward's tail when the second stream is on",
                                          alone_avg_ms=iso_bwd * 1e3 if iso_bwd else None,
                                          alone_achieved=nbytes / iso_bwd / 1e12 if iso_bwd else None,
                                          alone_frac=nbytes / iso_bwd / 1e12 / PEAK_HBM_TBPS if iso_bwd else None))

        return roof, roof_asm

    roof, roof_asm = rooflines(cfg, eng, durations, iso_fwd, args.fp64, args.config, True, iso_bwd)
    Mp = M * (p + 1)
    hl = (eng.trsm_nb, eng.lib_dense_gemm, bool(getattr(eng, "c_step_used", False)), getattr(eng, "pack_reduce", None),
          int(getattr(eng, "early_wire_numel", 0) or 0))
    graph_replay = bool(loop._graphs)
    ciq_stats = dict(eng.ciq_stats) if cfg.get("ciq") else None

    def time_other(cfg_o, name, fp64_, steps_, warm_):
        loop_o, eng_o, batch_o = build(cfg_o, fp64_)
        for k in range(warm_):
            loop_o.step(batch_o(k))
        events_on(eng_o, 8 if name == "c2" else 1)
        e_o, loss_o = window(loop_o, batch_o, warm_, steps_)
        eng_o.record_events = False
        roof_o, asm_o = rooflines(cfg_o, eng_o, event_durations(eng_o, EVENT_NAMES), None, fp64_, name, False)
        res = dict(workload=cfg_o["name"] + (" (float64 model mode)" if fp64_ else ""), steps=steps_, warmup=warm_,
                   ms_per_step=1e3 * e_o / steps_, steps_per_s=steps_ / e_o, final_loss=float(loss_o.item()),
                   one_call_step=bool(getattr(eng_o, "c_step_used", False)),
                   roofline_kernel=(roof_o or {}).get("kernel"), roofline_frac=(roof_o or {}).get("frac"),
                   roofline_avg_ms=(roof_o or {}).get("avg_ms"),
                   dense_frac=((roof_o or {}).get("dense") or {}).get("frac"), dense_avg_ms=((roof_o or {}).get("dense") or {}).get("avg_ms"),
                   gram_frac=((roof_o or {}).get("gram") or {}).get("frac"), gram_avg_ms=((roof_o or {}).get("gram") or {}).get("avg_ms"),
                   assembly_fwd_frac=((asm_o or {}).get("forward") or {}).get("frac"),
                   assembly_bwd_frac=((asm_o or {}).get("backward") or {}).get("frac"))
        del loop_o, eng_o, batch_o
        # the same reference-op-sequence CPU port as the headline's cpu_baseline, on a bounded sample (two to ten timed steps) of this configuration;
        # the CIQ configuration has no CPU training loop to time (the oracle evaluates single steps: 190 s for the C5 `init` state
        # on 8 cores, tests/golden/c5_step_init64.npz)
        if not args.no_cpu_baseline and not fp64_ and not cfg_o.get("ciq"):
            try:
                res["cpu_baseline"] = cpu_baseline(cfg_o, budget_s=10.0, min_steps=2)
            except Exception as ex:
                res["cpu_baseline"] = dict(error="%s: %s" % (type(ex).__name__, ex))
        return res

    def time_eval(cfg_e, n_test=100_000, eval_batch=4096):
        """eval_gp-equivalent throughput (reference directionalvi/directional_vi.py:271-305): predictive mean and
        variance (with likelihood noise) of all p + 1 outputs of `n_test` test rows in batches of `eval_batch`, eval mode (the
        Cholesky factor and its inverse cached across batches like the reference's @cached _cholesky_factor, DGVS.py:72), canonical
        derivative directions tiled per batch (:292-294), every batch's two vectors copied to the host (:297-298).  Forward only."""
        loop_e, eng_e, _ = build(cfg_e, False)
        d_, p_, M_ = cfg_e["d"], cfg_e["p"], cfg_e["M"]
        Mp_ = M_ * (p_ + 1)
        Xt = torch.rand(n_test, d_, device=device, generator=torch.Generator(device=device).manual_seed(7))
        params_e = loop_e.model._param_dict(loop_e.likelihood)
        Dcan = torch.eye(d_, device=device)[:p_]
        def run(collect):
            outs = []
            with torch.no_grad():
                for s0 in range(0, n_test, eval_batch):
                    xb = Xt[s0:s0 + eval_batch]
                    mu_, varn_ = eng_e.predict(params_e, xb, Dcan.repeat(xb.shape[0], 1), cache=True)
                    if collect:
                        outs.append((mu_.cpu(), varn_.cpu()))
            return outs
        run(False)                                   # warm-up: factorisation + inverse cached, buffers allocated
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(False); e1.record()
        torch.cuda.synchronize()
        dev_s = e0.elapsed_time(e1) * 1e-3
        t0 = time.perf_counter()
        outs = run(True)
        torch.cuda.synchronize()
        wall_s = time.perf_counter() - t0
        nb_ = (n_test + eval_batch - 1) // eval_batch
        # the dominant kernel of a batch is the forward solve A = L^-1 K_ZX of the training step (fp64 MFMA, M'^2 B' flop), followed by
        # the triangular W = L_S^T A (fp32 MFMA, M'^2 B'); roofline of the pair against the mixed peak, as for the step
        flop_solve = float(Mp_) ** 2 * n_test * (p_ + 1)
        mixed_s = flop_solve / 78.6e12 + flop_solve / 157.3e12
        res = dict(workload="eval_gp of %s: %d test rows, batches of %d, cached L^-1" % (cfg_e["name"], n_test, eval_batch),
                   rows_per_s=n_test / dev_s, outputs_per_s=n_test * (p_ + 1) / dev_s, ms_per_batch=1e3 * dev_s / nb_,
                   rows_per_s_with_host_copies=n_test / wall_s,
                   roofline=dict(bound="mfma", note="fp64 solve M'^2 B' + fp32 triangular product M'^2 B' per batch against 78.6 / 157.3 TF",
                                 achieved_tflops=2 * flop_solve / dev_s / 1e12, frac=mixed_s / dev_s),
                   mean_finite=bool(all(torch.isfinite(m_).all() for m_, _ in outs)),
                   variance_min=float(min(v_.min().item() for _, v_ in outs)))
        if not args.no_cpu_baseline:
            try:
                sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle"))
                import dsvgp_oracle as O_
                torch.set_num_threads(usable_cpus())
                Pc = {k: v.detach().cpu() for k, v in params_e.items()}
                nrows = 1024                                         # bounded sample: one 1024-row batch, factorisation included once
                xb = Xt[:nrows].cpu()
                Db = torch.eye(d_)[:p_].repeat(nrows, 1)
                t0 = time.perf_counter()
                with torch.no_grad():
                    O_.predictive(Pc, xb, Db, assembly=O_.kernel_matrix_refseq)
                t1 = time.perf_counter() - t0
                res["cpu_baseline"] = dict(value=nrows / t1, unit="rows/s", cores=usable_cpus(), kind="port",
                                           sample="oracle predictive (reference op sequence, fp64 Cholesky + solves) on one %d-row batch, "
                                                  "factorisation included (the reference caches it across batches): %.2f s" % (nrows, t1))
            except Exception as ex:
                res["cpu_baseline"] = dict(error="%s: %s" % (type(ex).__name__, ex))
        del loop_e, eng_e
        return res

    if args.eval_only and rank == 0:
        print(json.dumps(time_eval(cfg)), flush=True)
        return

    # ---- every other BASELINE configuration, timed by this same command (one GPU, default run): C2, C3, C5 and the float64 model
    # mode at C4 and C5 (fp64 msMINRES), each with its own warm-up and ONE timed window of the given number of steps (same bracketing as above)
    other = None
    if extras and not args.no_other_configs:
        other = {}
        del loop, eng, batch
        import gc
        for key, name, fp64_, steps_, warm_ in (("c2", "c2", False, 300, 20), ("c3", "c3", False, 30, 5), ("c5", "c5", False, 8, 3),
                                                  ("c4_fp64", "c4", True, 10, 3), ("c5_fp64", "c5", True, 4, 2)):
            gc.collect()
            torch.cuda.empty_cache()
            cfg_o = CONFIGS[name]
            try:
                other[key] = time_other(cfg_o, name, fp64_, steps_, warm_)
            except Exception as ex:          # (an extra must never take the headline line down)
                other[key] = dict(workload=cfg_o["name"], error="%s: %s" % (type(ex).__name__, ex))
        gc.collect()
        torch.cuda.empty_cache()
        try:
            other["eval_c4"] = time_eval(CONFIGS["c4"])
        except Exception as ex:
            other["eval_c4"] = dict(workload="eval_gp at C4", error="%s: %s" % (type(ex).__name__, ex))
        eng = None

    # ms_per_step = the timed window / steps, plus 1/50 of what a reporting step costs on top of a plain one (the reference reports
    # on every 50th step, directional_vi.py:255)
    ms_step = 1e3 * elapsed / args.steps + (report_extra_ms / 50.0 if report_extra_ms is not None else 0.0)
    trsm_nb_, lib_dense_, c_used_, pack_, early_numel_ = hl
    if rank == 0:
        out = {
            "metric": "ELBO steps/sec, DSVGP d=20 N=1M M=500 p=5" if args.config == "c4" else "ELBO steps/sec, " + cfg["name"],
            "value": 1e3 / ms_step, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64" if args.fp64 else ("f32 via bf16 x 3 split products (f64 Cholesky/solves)" if args.split_bf16 else "f32 (f64 Cholesky/solves)"),
            "data": "synthetic",
            "config": {"workload": cfg["name"], "global_batch": B, "per_gpu_batch": B // world, "M_prime": Mp,
                       "parallelism": "dp%d rows" % world, "trsm_nb": trsm_nb_, "final_loss": final_loss,
                       "graph_replay": graph_replay,
                       "one_call_step": c_used_,
                       "kernel_events": "HIP events around the roofline kernels on every %s timed step" % (
                           "single" if args.event_every == 1 else "%d-th" % args.event_every),
                       "dense_product": "rocBLAS sgemm" if lib_dense_ else "hand-written (gemm32.hip, v_mfma_f32_32x32x2_f32)",
                       "timed_window_ms_per_step": 1e3 * elapsed / args.steps,
                       "reporting_step_extra_ms": report_extra_ms,
                       "timed_step": "TrainLoop.step(): ELBO fast path every step, timed window = `steps` plain steps; "
                                     "ms_per_step = timed_window_ms_per_step + reporting_step_extra_ms / 50, where "
                                     "reporting_step_extra_ms is what the reference's every-50th-step report (loss.item(), nll "
                                     "from output.mean / output.variance[::p+1] of that forward pass: a host synchronisation + "
                                     "one [M', M'] x [M', B] product) costs on top of a plain step, measured over 5 reporting "
                                     "steps in a row after the window" if report_extra_ms is not None else
                                     "TrainLoop.step(): ELBO fast path every step; the reference's every-50th-step report is "
                                     "outside the timed region",
                       "repeat_ms": repeat_ms,
                       "repeat_note": "min / median / max ms per step of three windows of `steps` steps (the first one is the "
                                      "headline window)" if repeat_ms else None,
                       "split_bf16_second_figure": split_fig,
                       "other_configs": other},
            "roofline": roof,
            "roofline_dense": (roof or {}).get("dense"),
            "roofline_gram": (roof or {}).get("gram"),
            "roofline_assembly": roof_asm,
        }
        if roof:
            # which launch of the step is the largest (the `roofline` entry keeps the forward solve, the kernel north_star names)
            cands = [("forward solve (roofline)", roof.get("avg_ms")), ("dense K_ZX-bar product (roofline_dense)", (roof.get("dense") or {}).get("avg_ms")),
                     ("Gram product (roofline_gram)", (roof.get("gram") or {}).get("avg_ms"))]
            cands = [c for c in cands if c[1]]
            if cands:
                out["config"]["largest_launch"] = dict(name=max(cands, key=lambda c: c[1])[0], avg_ms={n_: t_ for n_, t_ in cands})
            roof.pop("dense", None)
            roof.pop("gram", None)
        if world > 1:
            out["rccl_ranks"] = dist.get_world_size()
            def avg_ms(name):
                durs = durations.get(name, [])
                return (sum(durs) / len(durs)) if durs else None
            t_w, t_r = avg_ms("early_reduce_wait"), avg_ms("final_reduce")
            out["config"]["collective"] = dict(backend=dist.get_backend(), algo=args.dp_algo, packed_triangle=pack_,
                                               early_operand_floats=early_numel_,
                                               exposed_early_reduce_wait_ms=t_w * 1e3 if t_w else None,
                                               final_reduce_ms=t_r * 1e3 if t_r else None,
                                               note="rank 0, HIP events on the main stream: the time the step stalls for the "
                                                    "[tril(G) | b] sum and the time of the closing gradient all-reduce",
                                               probe=coll_probe if coll_probe is not None else ("skipped: gloo rehearsal on one card" if rehearse else None),
                                               replica_check=replica_check, host_us_per_rank_step=dp_trace)
        elif dp_trace is not None:
            out["config"]["host_us_per_rank_step"] = dp_trace
        if cfg.get("ciq"):
            out["config"]["ciq"] = ciq_stats
        if world == 1 and not args.no_cpu_baseline and not cfg.get("ciq"):
            out["cpu_baseline"] = cpu_baseline(cfg, fp64=args.fp64)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
