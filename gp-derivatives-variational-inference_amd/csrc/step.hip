// One C entry point for the whole ELBO step (round 3; SURVEY.md section 8b's `dsvgp_elbo_terms` proposal taken to its end).
//
// dsvgp_elbo_step_f32 queues forward + backward of one DSVGP minibatch ELBO evaluation -- the iteration body of the reference's
// train_gp (directionalvi/directional_vi.py:245-249: output = model(x, derivative_directions=D); loss = -mll(output, y);
// loss.backward()) with the composition of DirectionalGradVariationalStrategy.forward (DGVS.py:89-208) -- from ONE host call:
// ~70 kernel launches on two HIP streams with events between them, no host synchronisation, every intermediate in ONE
// caller-owned workspace.  It is the same sequence of library calls that the Python engine (`_step.ElboEngine._elbo_fast`)
// issues through ~100 ctypes calls (1.05 ms of host time per step; the C2 step of 0.75 ms is bound by that); the arithmetic is
// shared (the dsvgp_* entry points below), so the parity tests of either path cover both.
//
// Scope: the ELBO fast path (Gram formulation) of the Cholesky-whitened strategy with every data point carrying its p
// directional derivatives, explicit-inverse regime (M' <= 8192), one rank or a data-parallel rank's "local" part
// (global_rows / include_kl).  PLL, per-output variances, CIQ, shared directions, derivative-free data and the jitter
// ladder after a failed factorisation stay on the Python-orchestrated path (the caller reads the status word with
// dsvgp_elbo_step_status and falls back).
#include "common.h"

#include <stdlib.h>
#include <string.h>

#include <new>

namespace {

struct Carve {
    size_t off = 0;
    size_t take(size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    }
};

inline int pad4(int n) { return (n + 3) / 4 * 4; }
inline int auto_nb(int Mp) {          // _step.ElboEngine._problem_size: the explicit-inverse regime up to M' = 8192
    int b = 64;
    while (b < Mp) b <<= 1;
    return b;
}

}  // namespace

struct dsvgp_step_plan {
    int M, d, p, B, Mp, Bp, DP, nb;
    bool per_output = false;          // plan of the per-output step (dsvgp_elbo_step_po_f32: PLL objective / per-output variances)
    size_t o_A64 = 0, o_U32 = 0, o_Abar = 0, o_Kb64 = 0, o_var = 0, o_varbar = 0;
    int world = 1;                    // > 1: a data-parallel rank's plan (dsvgp_elbo_step_dp_f32); sizes the buffers below
    int wq = 0, wr = 0;               // column block of [Q' | a] / row block of L-bar per rank
    size_t o_Qfull = 0, o_qcol64 = 0, o_cbT = 0, o_cbK = 0, o_slab = 0, slab_bytes = 0, o_lrow64 = 0;
    hipEvent_t ev_dp = nullptr;
    size_t bytes;
    // workspace offsets (bytes)
    size_t o_zero, zero_bytes;        // region cleared at the start of every step: info, sums, kl_buf
    size_t o_info, o_sums, o_klbuf, o_scal, o_hyp, o_center;
    size_t o_PZ, o_sZ, o_vZ, o_PX, o_sX, o_vX;
    size_t o_phi32 = 0; bool phi32_own = false;
    bool precleared_ge = false;
    bool precleared = false;                  // this step's targets of the main stream's split-K / OUT_LOWER products were cleared on the side stream (step_front)
    size_t o_L, o_trsm, o_potrf, o_Kzx, o_A32e, o_S32e, o_var0, o_stats, o_Ge, o_Qe64, o_S64e, o_Qe32, o_Kb32, o_G1, o_Yt, o_Kbar, o_kbwd, o_kbwd2;
    size_t o_arena, arena_bytes;     // contiguous region of everything a launcher would clear (see step_layout)
    int ldS, ldQ32;
    const void* pad_ready_for = nullptr;          // the workspace whose Qe32 pad columns have been zeroed
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_side = nullptr, ev_status = nullptr, ev_fork2 = nullptr, ev_var = nullptr, ev_dense = nullptr,
               ev_zx = nullptr, ev_pipe1 = nullptr, ev_pipe2 = nullptr, ev_s = nullptr;
    // timing pairs of the last TM_RING timed steps: forward solve, K_ZX assembly, K_ZX-bar kernel backward, Gram product, dense K_ZX-bar product
    static constexpr int TM_RING = 128, TM_PAIRS = 5;
    hipEvent_t tm_ring[TM_RING][2 * TM_PAIRS] = {};
    hipEvent_t* tm = tm_ring[0];
    long timed_steps = 0;
    bool timed = false;
    float* host_status = nullptr;                 // pinned: hyp[4] + info (as float bits)
};

// workspace layout of one (M, d, p, B); returns the total byte count (0: unsupported shape)
#ifndef STEP_KZZ_LOWER
#define STEP_KZZ_LOWER 1            // K_ZZ assembled on and below its 64-wide block diagonal only (what potrf.hip reads)
#endif
#ifndef STEP_S_LATE
#define STEP_S_LATE 2               // [S - I | m'] behind the chain, beside the forward solve: 1 as a one-workgroup-per-CU filler, 2 at full grid
#define STEP_S_LATE_MP 1024         // ... from this M' up (below: under the chain, as in rounds 2-5)
#endif
static size_t step_layout(int M, int d, int p, int B, dsvgp_step_plan* pl, int world = 1) {
    if (M <= 0 || d <= 0 || p < 0 || B <= 0 || world < 1 || world > 64 || (world > 1 && M < world)) return 0;
    const int q = p + 1, Mp = M * q, Bp = B * q, DP = dsvgp_packed_width(d);
    if (DP <= 0 || Mp > 8192 || (int64_t)Mp * Bp >= ((int64_t)1 << 31)) return 0;
    pl->M = M; pl->d = d; pl->p = p; pl->B = B; pl->Mp = Mp; pl->Bp = Bp; pl->DP = DP; pl->nb = auto_nb(Mp);
    pl->ldS = pad4(Mp + 1); pl->ldQ32 = pad4(Mp + 1);
    Carve c;
    pl->o_scal = c.take(8 * sizeof(float)); pl->o_center = c.take((size_t)d * sizeof(float));
    pl->o_PZ = c.take((size_t)Mp * DP * 4); pl->o_sZ = c.take((size_t)Mp * 4); pl->o_vZ = c.take((size_t)(M * p > 0 ? M * p : 1) * 4);
    pl->o_PX = c.take((size_t)Bp * DP * 4); pl->o_sX = c.take((size_t)Bp * 4); pl->o_vX = c.take((size_t)(B * p > 0 ? B * p : 1) * 4);
    pl->o_L = c.take((size_t)Mp * Mp * 8);
    pl->o_potrf = c.take(potrf_blocked_workspace_bytes(Mp));
    pl->o_Kzx = c.take((size_t)Mp * Bp * 4); pl->o_A32e = c.take((size_t)(Mp + 1) * Bp * 4);
    pl->o_var0 = c.take((size_t)Bp * 4); pl->o_stats = c.take(dsvgp_stats_workspace_bytes(Mp, Bp) + 16);
    pl->o_Qe32 = c.take((size_t)Mp * pl->ldQ32 * 4);
    pl->o_S64e = c.take((size_t)(Mp + 1) * ((Mp + 1) / 2 * 2) * 8);      // fp64 [S - I ; m^T / (2 vbar)], (M'+1) x M' (rewritten every step)
    const size_t kb = dsvgp_kernel_bwd_workspace_bytes(M, B, d, p), kz = dsvgp_kernel_bwd_workspace_bytes(M, M, d, p);
    pl->o_kbwd = c.take(kb > kz ? kb : kz); pl->o_kbwd2 = c.take(kb);
    // ---- the "arena": every buffer that some launcher clears before use (split-K targets, OUT_LOWER outputs), contiguous, so that
    // a small problem clears all of them with ONE memset (prezeroed mode below).  The solve workspace comes first: its scratch
    // T (the fp64 split-K target of a small fp32-only solve) is its tail, [trsm end - T bytes, trsm end).
    const int nrhs_max = Bp > Mp + 1 ? Bp : Mp + 1;
    const size_t trsm_bytes = dsvgp_trsm_workspace_bytes(Mp, nrhs_max, pl->nb);
    pl->o_trsm = c.take(trsm_bytes);
    // (dsvgp_trsm's layout: [L^-1 | L^-T | tmp (n + b) (b / 2) | T b x nrhs] doubles, b = nb here since nb / 2 < M' <= nb)
    pl->o_arena = (pl->o_trsm + sizeof(double) * ((size_t)2 * Mp * Mp + (size_t)(Mp + pl->nb) * (pl->nb / 2))) / 256 * 256;   // (aligned down into
    // the trtri scratch `tmp`, which the fused factorisation + inverse never uses: one aligned fill kernel)
    pl->o_S32e = c.take((size_t)Mp * pl->ldS * 4);
    pl->o_Ge = c.take((size_t)(Mp + 1) * Mp * 4); pl->o_Qe64 = c.take((size_t)Mp * (Mp + 2) * 8 + 64);
    pl->o_Kb32 = c.take((size_t)Mp * Bp * 4); pl->o_G1 = c.take((size_t)Mp * Mp * 8);
    pl->o_Yt = c.take((size_t)Mp * Mp * 8); pl->o_Kbar = c.take((size_t)Mp * Mp * 8);
    // small problems: the fp32 argument of Phi (tril([S - I | m'] [G ; b^T]), a split-K / OUT_LOWER target) gets its own place INSIDE the arena
    // -- cleared by the step's one memset -- instead of the [Q' | a] scratch, which would need a clearing launch between the dense product
    // that reads it and this product (round 6: 5 us of the M' = 600 step)
    // (round 6, second half: at every size -- large problems clear it on the side stream under the chain, `preclear` below)
    pl->phi32_own = true;
    pl->o_phi32 = c.take((size_t)Mp * pl->ldQ32 * 4);
    // head: hyp[4] | info[4 ints] | sums[4] | kl_buf[2 M' + 1]   (cleared every step; hyp + info go to the host in ONE copy).
    // It closes the arena, so that the small-problem mode clears both with one memset.
    pl->o_zero = c.off;
    pl->zero_bytes = (4 * sizeof(float) + 4 * sizeof(int) + 4 * sizeof(float) + (size_t)(2 * Mp + 1) * sizeof(float) + 255) / 256 * 256;
    pl->o_hyp = c.take(pl->zero_bytes);
    pl->o_info = pl->o_hyp + 4 * sizeof(float);
    pl->o_sums = pl->o_info + 4 * sizeof(int);
    pl->o_klbuf = pl->o_sums + 4 * sizeof(float);
    pl->arena_bytes = c.off - pl->o_arena;            // (T tail of the solve workspace ... head)
    pl->world = world;
    if (world > 1) {
        // data-parallel rank (global-Gram schedule with the replicated M'^3 stage sharded): its columns of
        // [Q' | a] (fp64 result of the column solve, fp32 copy in the caller's all-gather operand), the gathered [Q' | a] in
        // row-major form, the column block of the Cholesky backward, one fp64 row block of tril(L^T L-bar), and the slab of the fixed-order G L_S product (the replicas' L_S-bar must agree
        // bit for bit: nobody reduces it again)
        const int q1 = p + 1;
        pl->wq = ((Mp + 1 + world - 1) / world + 3) / 4 * 4;
        pl->wr = ((Mp + world - 1) / world + 1) / 2 * 2;            // (even: 16-byte aligned column offsets into the fp64 [S - I ; m^T])
        const int wc = ((M + world - 1) / world) * q1;              // widest column block of K_ZZ-bar
        const int ldQ64 = (Mp + 2) / 2 * 2;
        pl->o_Qfull = c.take((size_t)Mp * world * pl->wq * 4);
        pl->o_qcol64 = c.take((size_t)Mp * pl->wq * 8);
        pl->o_lrow64 = c.take((size_t)pl->wr * Mp * 8);
        pl->o_cbT = c.take((size_t)Mp * wc * 8);
        pl->o_cbK = c.take((size_t)Mp * wc * 8);
        pl->slab_bytes = (size_t)4 * Mp * Mp * 4 + 4096;
        pl->o_slab = c.take(pl->slab_bytes);
    }
    if (pl->per_output) {
        // per-output step: the fp64 copy of A = L^-1 K_ZX (left factor of L-bar = -tril(K_ZX-bar A^T)), U = L_S W, A-bar, the fp64 K_ZX-bar, the per-output variance and its gradient; W = L_S^T A reuses the K_ZX buffer (dead after the solve)
        pl->o_A64 = c.take((size_t)Mp * Bp * 8);
        pl->o_U32 = c.take((size_t)Mp * Bp * 4);
        pl->o_Abar = c.take((size_t)Mp * Bp * 4);
        pl->o_Kb64 = c.take((size_t)Mp * Bp * 8);
        pl->o_var = c.take((size_t)Bp * 4);
        pl->o_varbar = c.take((size_t)Bp * 4);
    }
    pl->bytes = c.off + 256;
    return pl->bytes;
}

// flag 32: bf16 plane triples of [A ; mu_bar^T] ((M'+1) x B'), of its transpose (B' x (M'+1)) and of [Q' | a] (M' x (M'+1))
static size_t split_offsets(int Mp, int Bp, size_t* o_pat, size_t* o_pq) {
    size_t off = (dsvgp_split3_bytes(Mp + 1, Bp) + 255) / 256 * 256;
    if (o_pat) *o_pat = off;
    off += (dsvgp_split3_bytes(Bp, Mp + 1) + 255) / 256 * 256;
    if (o_pq) *o_pq = off;
    off += (dsvgp_split3_bytes(Mp, Mp + 1) + 255) / 256 * 256;
    return off;
}
extern "C" size_t dsvgp_elbo_step_split_bytes(int M, int d, int p, int B) {
    if (M <= 0 || d <= 0 || p < 0 || B <= 0) return 0;
    return split_offsets(M * (p + 1), B * (p + 1), nullptr, nullptr);
}

extern "C" size_t dsvgp_elbo_step_workspace_bytes(int M, int d, int p, int B) {
    dsvgp_step_plan pl{};
    return step_layout(M, d, p, B, &pl);
}

extern "C" size_t dsvgp_elbo_step_dp_workspace_bytes(int M, int d, int p, int B, int world) {
    dsvgp_step_plan pl{};
    return step_layout(M, d, p, B, &pl, world);
}

static int plan_create(dsvgp_ctx* ctx, int M, int d, int p, int B, int world, dsvgp_step_plan** out, bool per_output = false) {
    if (!ctx || !out) return DSVGP_EINVAL;
    dsvgp_step_plan* pl = new (std::nothrow) dsvgp_step_plan();
    if (!pl) return DSVGP_EINVAL;
    pl->per_output = per_output;
    if (!step_layout(M, d, p, B, pl, world)) { delete pl; return DSVGP_EINVAL; }
    bool ok = hipStreamCreateWithFlags(&pl->side, hipStreamNonBlocking) == hipSuccess;
    hipEvent_t* evs[] = {&pl->ev_fork, &pl->ev_side, &pl->ev_status, &pl->ev_fork2, &pl->ev_var, &pl->ev_dense, &pl->ev_zx, &pl->ev_dp,
                         &pl->ev_pipe1, &pl->ev_pipe2, &pl->ev_s};
    for (hipEvent_t* e : evs) ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    for (auto& slot : pl->tm_ring) for (hipEvent_t& e : slot) ok = ok && hipEventCreate(&e) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&pl->host_status, 8 * sizeof(float), hipHostMallocDefault) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        for (hipEvent_t* e : evs) if (*e) (void)hipEventDestroy(*e);
        for (auto& slot : pl->tm_ring) for (hipEvent_t e : slot) if (e) (void)hipEventDestroy(e);
        if (pl->side) (void)hipStreamDestroy(pl->side);
        if (pl->host_status) (void)hipHostFree(pl->host_status);
        delete pl;
        return 1000 + (int)hipErrorOutOfMemory;
    }
    pl->host_status[4] = 0.f;
    *out = pl;
    return 0;
}
extern "C" int dsvgp_elbo_step_plan_create(dsvgp_ctx* ctx, int M, int d, int p, int B, dsvgp_step_plan** out) {
    return plan_create(ctx, M, d, p, B, 1, out);
}
// plan of ONE RANK of a `world`-rank data-parallel job (B = this rank's rows of the minibatch): dsvgp_elbo_step_dp_f32
extern "C" int dsvgp_elbo_step_dp_plan_create(dsvgp_ctx* ctx, int M, int d, int p, int B, int world, dsvgp_step_plan** out) {
    if (world < 2) return DSVGP_EINVAL;
    return plan_create(ctx, M, d, p, B, world, out);
}

// plan / workspace of the PER-OUTPUT step (dsvgp_elbo_step_po_f32): one rank
extern "C" int dsvgp_elbo_step_po_plan_create(dsvgp_ctx* ctx, int M, int d, int p, int B, dsvgp_step_plan** out) {
    return plan_create(ctx, M, d, p, B, 1, out, true);
}
extern "C" size_t dsvgp_elbo_step_po_workspace_bytes(int M, int d, int p, int B) {
    dsvgp_step_plan pl{};
    pl.per_output = true;
    return step_layout(M, d, p, B, &pl);
}

extern "C" int dsvgp_elbo_step_plan_destroy(dsvgp_step_plan* pl) {
    if (!pl) return DSVGP_EINVAL;
    hipEvent_t evs[] = {pl->ev_fork, pl->ev_side, pl->ev_status, pl->ev_fork2, pl->ev_var, pl->ev_dense, pl->ev_zx, pl->ev_dp, pl->ev_pipe1,
                        pl->ev_pipe2, pl->ev_s};
    for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    for (auto& slot : pl->tm_ring) for (hipEvent_t e : slot) if (e) (void)hipEventDestroy(e);
    if (pl->side) (void)hipStreamDestroy(pl->side);
    if (pl->host_status) (void)hipHostFree(pl->host_status);
    delete pl;
    return 0;
}

// HIP-event durations (ms) of a step queued with flag 4 -- `back` steps before the most recent one (the plan keeps the last 128) --
// each measured on the stream its kernel ran on: ms[0] forward panel solve A = L^-1 K_ZX, ms[1] K_ZX assembly, ms[2] K_ZX-bar
// kernel backward.  Waits for that step.
static int step_timings(dsvgp_step_plan* pl, int back, float* ms, int n) {
    if (!pl || !ms || n < 1 || n > dsvgp_step_plan::TM_PAIRS || back < 0 || back >= dsvgp_step_plan::TM_RING || back >= pl->timed_steps)
        return DSVGP_EINVAL;
    hipEvent_t* tm = pl->tm_ring[(pl->timed_steps - 1 - back) % dsvgp_step_plan::TM_RING];
    for (int k = 0; k < n; ++k) {
        hipError_t e = hipEventSynchronize(tm[2 * k + 1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms[k], tm[2 * k], tm[2 * k + 1]);
        if (e != hipSuccess) return 1000 + (int)e;
    }
    return 0;
}
extern "C" int dsvgp_elbo_step_timings(dsvgp_step_plan* pl, int back, float* ms3) { return step_timings(pl, back, ms3, 3); }
// ... and the two fp32 [M', B'] products of the step as well: ms5[3] the Gram product [tril(G) ; b^T] = tril([A ; mu_bar^T] A^T), ms5[4] the dense
// product K_ZX-bar = [Q' | a][A ; mu_bar^T] (bench.py: roofline_gram / roofline_dense)
extern "C" int dsvgp_elbo_step_timings5(dsvgp_step_plan* pl, int back, float* ms5) { return step_timings(pl, back, ms5, 5); }

extern "C" size_t dsvgp_elbo_step_plan_bytes(const dsvgp_step_plan* pl) { return pl ? pl->bytes : 0; }
// number of steps queued with flag 4 so far (the step queued last with that flag has index count - 1: callers that keep an index
// per step read its timings with back = count - 1 - index, whatever was queued in between)
extern "C" long dsvgp_elbo_step_timed_count(const dsvgp_step_plan* pl) { return pl ? pl->timed_steps : 0; }

// Where an intermediate of the step queued last lies in the caller's workspace (valid until the next step on that workspace):
//   which = 0: [A ; mu_bar^T] = [L^-1 K_ZX ; residual row], float [M' + 1, B'];  1: K_ZX, float [M', B'];  2: the Cholesky factor L,
//   double [M', M'] (lower);  3: L^-1, double [M', M'] (lower);  4: {lengthscale, outputscale, noise, 0}, float [1, 4].   Used for the reference's every-50th-step nll print
//   (directional_vi.py:255-260), which needs the predictive variance of the function-value rows of THAT forward pass.
extern "C" int dsvgp_elbo_step_locate(const dsvgp_step_plan* pl, int which, size_t* offset_bytes, int* rows, int* cols, int64_t* ld) {
    if (!pl || !offset_bytes || !rows || !cols || !ld) return DSVGP_EINVAL;
    switch (which) {
        case 0: *offset_bytes = pl->o_A32e; *rows = pl->Mp + 1; *cols = pl->Bp; *ld = pl->Bp; return 0;
        case 1: *offset_bytes = pl->o_Kzx; *rows = pl->Mp; *cols = pl->Bp; *ld = pl->Bp; return 0;
        case 2: *offset_bytes = pl->o_L; *rows = pl->Mp; *cols = pl->Mp; *ld = pl->Mp; return 0;
        case 3: *offset_bytes = pl->o_trsm; *rows = pl->Mp; *cols = pl->Mp; *ld = pl->Mp; return 0;
        case 4: *offset_bytes = pl->o_hyp; *rows = 1; *cols = 4; *ld = 4; return 0;
        case 5: *offset_bytes = pl->o_info; *rows = 1; *cols = 1; *ld = 1; return 0;     // the factorisation's status word (int32; 0 = positive definite)
        default: return DSVGP_EINVAL;
    }
}

// Wait for the factorisation of the step queued last (NOT for the rest of the step) and return its status word (0 = positive
// definite; k > 0: pivot k failed -- the caller runs psd_safe_cholesky's jitter ladder on the piecewise path) and the constrained
// hyper-parameters {lengthscale, outputscale, noise, 0} of that step.
extern "C" int dsvgp_elbo_step_status(dsvgp_step_plan* pl, float* hyp4, int* info) {
    if (!pl || !info) return DSVGP_EINVAL;
    hipError_t e = hipEventSynchronize(pl->ev_status);
    if (e != hipSuccess) return 1000 + (int)e;
    if (hyp4) for (int i = 0; i < 4; ++i) hyp4[i] = pl->host_status[i];
    int v;
    memcpy(&v, &pl->host_status[4], sizeof(int));
    *info = v;
    return 0;
}

#ifndef STEP_VAR_LATE
#define STEP_VAR_LATE 2             // 1: (B' >= STEP_Q_CLASSIC_BP) / 2: (always) the variational block behind the dense product, see dsvgp_elbo_step_f32
#endif
#ifndef STEP_VAR_LATE_MP
#define STEP_VAR_LATE_MP 1536       // ... from this M' on (at M' = 600 the Cholesky backward is too short to hide it: 0.549 -> 0.560 ms)
#endif
#ifndef STEP_Q_CLASSIC_BP
#define STEP_Q_CLASSIC_BP 16384     // minibatch columns B' from which the [Q' | a] solve of the one-call step keeps the register-staged lean kernel
#endif
#ifndef STEP_PIPE_K1
#define STEP_PIPE_K1 450    // forward solve under the chain (flag 128): the side stream starts rows [0, r1) after launch k1 = 45 % of the block rows,
#define STEP_PIPE_K2 750    // rows [r1, r2) after launch k2 = 75 %
#define STEP_PIPE_PAD 49152 // unused dynamic LDS of the side-stream pieces: one 36 KB + 48 KB workgroup per CU beside one 70 KB chain workgroup
#endif
#ifndef STEP_PHI64
#define STEP_PHI64 0        // 1: tril(L^T L-bar) always with fp64 accumulation (probes; flag 64 of the step does the same at run time)
#endif
#define STEP_CALL(expr)                 \
    do {                                \
        const int rc__ = (expr);        \
        if (rc__) { ctx->stream = main; return rc__; } \
    } while (0)
#define STEP_HIP(expr)                  \
    do {                                \
        const hipError_t e__ = (expr);  \
        if (e__ != hipSuccess) { ctx->stream = main; return 1000 + (int)e__; } \
    } while (0)

// flags: bit 0 = overlap (second stream: K_ZX assembly + S = L_S L_S^T under the Cholesky chain, the L_S / m gradients next to
//                the Q' solve and the dense product, K_ZX-bar's kernel backward next to the Cholesky backward when B' <= 2 M');
//        bit 1 = include the KL term (a data-parallel rank other than 0 leaves it out);
//        bit 2 = record HIP-event timings; bit 3 = the workspace may have been written by somebody else (re-clear the paddings).
// io->flat .. flat + flat_floats is cleared here (the gradient slots must start from zero); every gradient pointer of io points
// into it.  All pointers are device pointers; nothing is read back.
// K_ZX and its backward: on the canonical-direction kernels when the caller states the minibatch's directions as an index list
// (io->dir_idx, include/dsvgp.h) and they take the geometry, on the general kernels otherwise
static inline bool zx_canon(const dsvgp_elbo_step_io* io, int d, int p) { return io->dir_idx && dsvgp_kernel_canon_supported(d, p); }
// ... and the both-sides one-hot kernels (dsvgp_kernel_fwd_canon2 / _bwd_canon2) when the caller also states that the inducing directions are
// the same unit vectors (io->v_one_hot: the full-gradient SVGP) -- K_ZZ and K_ZX, forward and backward
static inline bool zz_canon2(const dsvgp_elbo_step_io* io, int d, int p) {
    return io->dir_idx && io->v_one_hot && dsvgp_kernel_canon2_supported(d, p);
}
static inline bool rows_in_16_byte_pieces(const void* G, int64_t ldg, int esz) { return ldg % (16 / esz) == 0 && (uintptr_t)G % 16 == 0; }
static int zz_fwd(dsvgp_ctx* ctx, const dsvgp_elbo_step_io* io, const float* PZ, const float* sZ, int M, int d, int p, const float* hyp,
                  double* L, int64_t ld) {
    // (the result is factored in place by the blocked Cholesky chain, which reads the 64 x 64 blocks on and below the block diagonal only:
    //  the assembly skips the rest -- common.h: fwd_lower_only)
    struct LowerOnly {
        dsvgp_ctx* c; bool prev;
        LowerOnly(dsvgp_ctx* c_) : c(c_), prev(c_->fwd_lower_only) { c->fwd_lower_only = STEP_KZZ_LOWER != 0; }
        ~LowerOnly() { c->fwd_lower_only = prev; }
    } lower_only(ctx);
    if (zz_canon2(io, d, p))
        return dsvgp_kernel_fwd_canon2(ctx, PZ, M, PZ, M, d, p, io->dir_idx, io->dir_idx_base, hyp, io->kzz_jitter, L, ld, 1);
    return dsvgp_kernel_fwd(ctx, PZ, sZ, M, PZ, sZ, M, d, p, hyp, io->kzz_jitter, L, ld, 1);
}
// K_ZZ-bar (double, symmetric) or a column block of it (data-parallel phase 4: P2 / s2 / n2 name the block's points)
static int zz_bwd(dsvgp_ctx* ctx, const dsvgp_elbo_step_io* io, const double* Kbar, int64_t ld, const float* PZ, const float* sZ,
                  const float* vZ, int M, const float* P2, const float* s2, int n2, int d, int p, const float* hyp, void* ws) {
    if (zz_canon2(io, d, p) && rows_in_16_byte_pieces(Kbar, ld, 8))
        return dsvgp_kernel_bwd_canon2(ctx, Kbar, ld, 1, PZ, vZ, M, P2, n2, d, p, io->dir_idx, io->dir_idx_base, hyp, 1, io->dZ, io->dV,
                                       io->d_hyp, ws);
    return dsvgp_kernel_bwd(ctx, Kbar, ld, 1, PZ, sZ, vZ, M, P2, s2, n2, d, p, hyp, 1, io->dZ, io->dV, io->d_hyp, ws);
}
static int zx_fwd(dsvgp_ctx* ctx, const dsvgp_elbo_step_io* io, const float* PZ, const float* sZ, int M, const float* PX, const float* sX,
                  int B, int d, int p, const float* hyp, float* Kzx, int64_t ld) {
    if (zz_canon2(io, d, p)) return dsvgp_kernel_fwd_canon2(ctx, PZ, M, PX, B, d, p, io->dir_idx, io->dir_idx_base, hyp, 0.f, Kzx, ld, 0);
    if (zx_canon(io, d, p)) return dsvgp_kernel_fwd_canon(ctx, PZ, sZ, M, PX, sX, B, d, p, io->dir_idx, io->dir_idx_base, hyp, Kzx, ld);
    return dsvgp_kernel_fwd(ctx, PZ, sZ, M, PX, sX, B, d, p, hyp, 0.f, Kzx, ld, 0);
}
static int zx_bwd(dsvgp_ctx* ctx, const dsvgp_elbo_step_io* io, const float* Kb32, int64_t ld, const float* PZ, const float* sZ,
                  const float* vZ, int M, const float* PX, const float* sX, int B, int d, int p, const float* hyp, void* ws) {
    if (zz_canon2(io, d, p) && rows_in_16_byte_pieces(Kb32, ld, 4))
        return dsvgp_kernel_bwd_canon2(ctx, Kb32, ld, 0, PZ, vZ, M, PX, B, d, p, io->dir_idx, io->dir_idx_base, hyp, 0, io->dZ, io->dV,
                                       io->d_hyp, ws);
    if (zx_canon(io, d, p))
        return dsvgp_kernel_bwd_canon(ctx, Kb32, ld, 0, PZ, sZ, vZ, M, PX, sX, B, d, p, io->dir_idx, io->dir_idx_base, hyp, io->dZ, io->dV,
                                      io->d_hyp, ws);
    return dsvgp_kernel_bwd(ctx, Kb32, ld, 0, PZ, sZ, vZ, M, PX, sX, B, d, p, hyp, 0, io->dZ, io->dV, io->d_hyp, ws);
}

static int step_validate(dsvgp_ctx* ctx, dsvgp_step_plan* pl, const dsvgp_elbo_step_io* io, void* workspace, size_t workspace_bytes) {
    if (!ctx || !pl || !io || !workspace || workspace_bytes < pl->bytes || ((uintptr_t)workspace % 256)) return DSVGP_EINVAL;
    if (!io->Z || !io->m || !io->LS || !io->constant || !io->raw_lengthscale || !io->raw_outputscale || !io->raw_noise || !io->x ||
        !io->y || !io->flat || !io->dZ || !io->dm || !io->dLS || !io->d_hyp || !io->d_constant || !io->d_raw_lengthscale ||
        !io->d_raw_outputscale || !io->d_raw_noise || !io->loss || !io->mu || io->num_data <= 0 || io->global_rows <= 0)
        return DSVGP_EINVAL;
    const int p = pl->p, Mp = pl->Mp;
    if (p > 0 && (!io->V || !io->D || !io->dV)) return DSVGP_EINVAL;
    if (io->ldls < Mp || io->lddls < Mp) return DSVGP_EINVAL;
    return 0;
}

#define STEP_LOCALS \
    const int M = pl->M, d = pl->d, p = pl->p, B = pl->B, Mp = pl->Mp, Bp = pl->Bp, nb = pl->nb; \
    char* w = (char*)workspace; \
    int* info = (int*)(w + pl->o_info); \
    float* sums = (float*)(w + pl->o_sums); \
    float* kl_buf = (float*)(w + pl->o_klbuf); \
    float* scal = (float*)(w + pl->o_scal); \
    float* hyp = (float*)(w + pl->o_hyp); \
    float* center = (float*)(w + pl->o_center); \
    float *PZ = (float*)(w + pl->o_PZ), *sZ = (float*)(w + pl->o_sZ), *vZ = (float*)(w + pl->o_vZ); \
    float *PX = (float*)(w + pl->o_PX), *sX = (float*)(w + pl->o_sX), *vX = (float*)(w + pl->o_vX); \
    double* L = (double*)(w + pl->o_L); \
    void* trsm_ws = w + pl->o_trsm; \
    void* potrf_ws = w + pl->o_potrf; \
    float* Kzx = (float*)(w + pl->o_Kzx); \
    float* A32e = (float*)(w + pl->o_A32e); \
    float* S32e = (float*)(w + pl->o_S32e); \
    float* var0 = (float*)(w + pl->o_var0); \
    void* stats_ws = w + pl->o_stats; \
    float* Ge = (float*)(w + pl->o_Ge); \
    double* Qe64 = (double*)(w + pl->o_Qe64); \
    double* S64e = (double*)(w + pl->o_S64e); \
    float* Qe32 = (float*)(w + pl->o_Qe32); \
    float* Kb32 = (float*)(w + pl->o_Kb32); \
    double* G1 = (double*)(w + pl->o_G1); \
    double* Yt = (double*)(w + pl->o_Yt); \
    double* Kbar = (double*)(w + pl->o_Kbar); \
    void* kbwd_ws = w + pl->o_kbwd; \
    void* kbwd_ws2 = w + pl->o_kbwd2; \
    const int ldS = pl->ldS, ldQ32 = pl->ldQ32, ldQ64 = (Mp + 2) / 2 * 2, ldST = (Mp + 1) / 2 * 2; \
    const double rows = io->global_rows;

// Everything up to and including the Gram product [tril(G) ; b^T] = tril([A ; mu_bar^T] A^T) of this rank's rows (shared by the
// one-GPU step and by phase 0 of a data-parallel rank)
struct ZeroedScope {                          // ctx->prezeroed = true around a call whose target the step has cleared already
    dsvgp_ctx* c; bool prev;
    ZeroedScope(dsvgp_ctx* c_, bool on) : c(c_), prev(c_->prezeroed) { if (on) c->prezeroed = true; }
    ~ZeroedScope() { c->prezeroed = prev; }
};
// preclear (the one-GPU step): problems too large for the one-memset mode clear the four split-K / OUT_LOWER targets of the MAIN stream's later
// products -- [tril(G) ; b^T], the Phi argument, the two products of the Cholesky backward: 216 MB at M' = 3000 -- on the SIDE stream at
// the start of the step, ahead of K_ZX's assembly: the main stream's wait for that assembly (in front of the forward solve) orders them, no
// new join, and the launchers of those products are told so (ZeroedScope).  Before, four fill launches sat on the main stream between those
// products: 23 + 8 + 40 + 12 us and their launch gaps at M' = 3000 (profiles/r06_f_preclear.txt).
static int step_front(dsvgp_ctx* ctx, dsvgp_step_plan* pl, const dsvgp_elbo_step_io* io, void* workspace, int flags, bool preclear = false) {
    STEP_LOCALS
    const bool overlap = (flags & 1) && !ctx->det_slab;           // (deterministic mode: the scratch serves one stream)
    const bool include_kl = flags & 2, timed = flags & 4;
    pl->timed = timed;
    if (timed) { pl->tm = pl->tm_ring[pl->timed_steps % dsvgp_step_plan::TM_RING]; ++pl->timed_steps; }
#define STEP_TIME(slot) do { if (timed) STEP_HIP(hipEventRecord(pl->tm[slot], ctx->stream)); } while (0)
    const hipStream_t main = ctx->stream, side = pl->side;

    // ---- clears: gradient slots + loss, the status / sums / KL scratch; once per workspace the pad columns of [Q' | a] (fp32).
    // Small problems (launch-bound: M' of a few hundred) also clear the whole arena of split-K / OUT_LOWER outputs here, ONCE,
    // and tell the launchers so (ctx->prezeroed): a dozen ~5 us fill launches fewer per step.  Large problems keep the launchers'
    // own clears (only the products that actually split K clear anything there).
    const bool prezero = pl->arena_bytes <= ((size_t)48 << 20) && !ctx->det_slab;
    static const bool preclear_env = !getenv("DSVGP_PRECLEAR") || atoi(getenv("DSVGP_PRECLEAR")) != 0;
    pl->precleared = preclear && preclear_env && overlap && !prezero && !ctx->det_slab && !(flags & (16 | 32 | 64));
    pl->precleared_ge = false;
    STEP_HIP(hipMemsetAsync(io->flat, 0, io->flat_floats * sizeof(float), main));
    if (prezero) STEP_HIP(hipMemsetAsync(w + pl->o_arena, 0, pl->arena_bytes, main));
    else STEP_HIP(hipMemsetAsync(w + pl->o_zero, 0, pl->zero_bytes, main));
    if (pl->pad_ready_for != workspace || (flags & 8)) {
        STEP_HIP(hipMemsetAsync(Qe32, 0, (size_t)Mp * ldQ32 * sizeof(float), main));
        STEP_HIP(hipMemsetAsync(S32e, 0, (size_t)Mp * ldS * sizeof(float), main));       // (pad columns of [S - I | m']: a k-contiguous operand of the fp32 kernels)
        pl->pad_ready_for = workspace;
    }
    struct PrezeroGuard {               // (the flag must not outlive the call, whatever path returns)
        dsvgp_ctx* c; bool prev;
        PrezeroGuard(dsvgp_ctx* c_, bool on) : c(c_), prev(c_->prezeroed) { c->prezeroed = on; }
        ~PrezeroGuard() { c->prezeroed = prev; }
    } prezero_guard(ctx, prezero);
    // ---- hyper-parameters + centre (one launch), packed inducing rows (DGVS.py:128-149 via RBFKernelDirectionalGrad.py:57-107)
    // (round 6: ONE launch -- centre, hyper-parameters, the packed rows of both point sets; three launches before)
    STEP_CALL(launch_pack_both(main, io->Z, io->V, M, io->x, io->D, B, d, p, io->raw_lengthscale, io->raw_outputscale, io->raw_noise, hyp,
                               center, PZ, sZ, vZ, PX, sX, vX));
    // ---- prologue that does not depend on L: K_ZX, [S - I | m / (2 vbar)] -- on the side stream under the Cholesky chain
    // (round 6) with the second stream, [S - I | m'] is formed BEHIND the chain, beside the forward solve, not under the chain's first
    // launches: it is not needed before the [Q' | a] solve, and under the chain its product -- even as a one-workgroup-per-CU filler --
    // shares the critical workgroup's CU: the chain's launches 1 .. 5 took 41 / 80 / 72 / 63 / 36 us instead of 24 at M' = 3000 (175 us
    // of the step's critical path; profiles/r06_*_timeline_c4.txt).  K_ZX's assembly stays under the chain: the solve needs it.
    // Measured (one box, alternating; profiles/r06_d_s_late.txt): C4 12.006 / 12.038 ms under the chain, 11.967 / 11.982 late as a filler,
    // 11.948 / 11.925 late at full grid; C3 6.32 / 6.36 / 6.29; C2 (M' = 600: the chain's launches are short either way) unchanged within
    // its run-to-run spread -- small problems keep the old place.
    static const int s_late_env = getenv("DSVGP_S_LATE") ? atoi(getenv("DSVGP_S_LATE")) : -1;     // 0: under the chain (rounds 2-5); 1: late, filler; 2: late, full grid
    const int s_late = !overlap ? 0 : s_late_env >= 0 ? s_late_env : (Mp >= STEP_S_LATE_MP ? STEP_S_LATE : 0);
    auto prologue_s = [&](bool background) -> int {
        // S = tril(L_S) tril(L_S)^T: lower triangle + mirror (n^3 / 6 multiply-adds)
        int rc = dsvgp_gemm(ctx, 0, DSVGP_GEMM_A_LOWER | DSVGP_GEMM_TRANS_B | DSVGP_GEMM_B_UPPER | DSVGP_GEMM_OUT_LOWER |
                            (background ? DSVGP_GEMM_BACKGROUND : 0), Mp, Mp, Mp, 1.0, io->LS, io->ldls, io->LS, io->ldls, 0.0, nullptr, 0,
                            S32e, ldS, nullptr, 0, nullptr);
        if (rc) return rc;
        // mirror + [S - I | m / (2 vbar)] AND its fp64 copy, TRANSPOSED -- [S - I ; m^T / (2 vbar)], (M'+1) x M': S - I is symmetric, so its
        // rows are copied as they lie and only the extra column becomes a row -- the left operand of the Cholesky backward's first product
        // (chol_tail below) then streams as an mn-contiguous operand (the lean fp64 kernel's faster staging path: 49 against 46 TF)
        rc = launch_mirror_sminus_i_col(ctx->stream, S32e, Mp, ldS, io->m, hyp, (float)rows, S64e, ldST);
        if (rc) return rc;
        const hipError_t e = hipGetLastError();            // (one call: it clears the error it returns)
        return e == hipSuccess ? 0 : 1000 + (int)e;
    };
    auto prologue = [&](bool background) -> int {
        int rc = 0;
        if (timed && hipEventRecord(pl->tm[2], ctx->stream) != hipSuccess) return 1000 + (int)hipGetLastError();
        rc = zx_fwd(ctx, io, PZ, sZ, M, PX, sX, B, d, p, hyp, Kzx, Bp);
        if (rc) return rc;
        if (timed && hipEventRecord(pl->tm[3], ctx->stream) != hipSuccess) return 1000 + (int)hipGetLastError();
        if (s_late) return 0;
        return prologue_s(background);
    };
    // ---- the forward solve PIPELINED under the Cholesky chain (flag 128).  Row block i of A = L^-1 K_ZX needs rows i of L^-1 only,
    // and those are final as soon as the chain's launch i has run (potrf.hip writes row block k of the inverse in launch k);
    // the chain's later launches are bound by ONE workgroup or by a few hundred latency-bound tiles and leave the matrix pipes
    // mostly idle.  So the solve is queued in three row ranges: [0, r1) and [r1, r2) on the side stream behind events the chain
    // records after its launches k1 / k2, one workgroup per CU (LDS padding: the chain's launches must keep finding room), and
    // the rest on the main stream after the chain as before.  Pieces are row ranges of the SAME product (gemm64.hip's wide kernel
    // with a row offset into the triangle): the arithmetic of every output element is unchanged.
    float* A32 = A32e;
    float* mu_bar = A32e + (size_t)Mp * Bp;
    static const int pipe_k1 = getenv("DSVGP_PIPE_K1") ? atoi(getenv("DSVGP_PIPE_K1")) : STEP_PIPE_K1;      // (per mille of the block rows)
    static const int pipe_k2 = getenv("DSVGP_PIPE_K2") ? atoi(getenv("DSVGP_PIPE_K2")) : STEP_PIPE_K2;
    static const int pipe_pad = getenv("DSVGP_PIPE_PAD") ? atoi(getenv("DSVGP_PIPE_PAD")) : STEP_PIPE_PAD;
    const int nblk64 = (Mp + 63) / 64;
    // launches after which the side stream may start its two row ranges: the chain's launches are k = -1 .. nblk64 - 2, so an event is only
    // ever recorded for k <= nblk64 - 2 (a wait on a never-recorded event does not wait), and the last range [r2, M') must not be empty;
    // values outside 0 <= k1 <= k2 <= nblk64 - 2 (DSVGP_PIPE_K1 / K2 are per mille of the block rows, read with atoi) fall back to the serial solve
    const int pk1 = (int)((int64_t)nblk64 * pipe_k1 / 1000), pk2 = (int)((int64_t)nblk64 * pipe_k2 / 1000);
    const bool pipe = overlap && (flags & 128) && nb >= Mp && Mp % 2 == 0 && Bp % 4 == 0 && ((uintptr_t)Kzx % 16) == 0 &&
                      (int64_t)nblk64 * ((Bp + 63) / 64) >= 8192 && nblk64 >= 16 && pipe_k1 >= 0 && pk1 <= pk2 && pk2 <= nblk64 - 2 &&
                      pipe_pad >= 0 && pipe_pad <= 98304;
    const double* LinvT = (const double*)trsm_ws + (size_t)Mp * Mp;
    auto solve_rows = [&](int ra, int rb, int pad) -> int {        // rows [ra, rb) of A = L^-1 K_ZX
        GemmArgs f{};
        f.batch = 1; f.splitk = 1;
        f.M = rb - ra; f.N = Bp; f.K = rb; f.tri_off = ra; f.wide64 = 1; f.lds_pad = pad;
        f.A = LinvT + ra; f.lda = Mp; f.flags = DSVGP_GEMM_TRANS_A | DSVGP_GEMM_A_LOWER | DSVGP_GEMM_B_IS_FLOAT;
        f.B = Kzx; f.ldb = Bp; f.alpha = 1.0; f.beta = 0.0;
        f.C = nullptr; f.C32 = A32 + (size_t)ra * Bp; f.ldc32 = Bp;
        return launch_gemm(ctx->stream, 1, f);
    };
    int r2 = 0;
    if (overlap) STEP_HIP(hipEventRecord(pl->ev_fork, main));
    // ---- K_ZZ + jitter (fp32 values widened, DGVS.py:74,144), Cholesky with the fused inverse (potrf.hip).  K_ZZ's assembly AND the chain
    // are QUEUED ahead of the side stream's prologue (round 6): the host needs 20-40 us to queue those three launches and their events,
    // and the main stream -- the step's critical path from here to the end of the chain -- sat idle for that long at M' = 600; behind the
    // chain's launches the host is ahead of the device and the prologue still starts under the chain's first launches
    STEP_CALL(zz_fwd(ctx, io, PZ, sZ, M, d, p, hyp, L, Mp));
    auto clear_tail_targets = [&]() -> int {        // the Cholesky backward's two products and the Phi argument (read behind step_front's end only)
        STEP_HIP(hipMemsetAsync(Yt, 0, (size_t)Mp * Mp * sizeof(double), side));
        STEP_HIP(hipMemsetAsync(Kbar, 0, (size_t)Mp * Mp * sizeof(double), side));
        STEP_HIP(hipMemsetAsync(w + pl->o_phi32, 0, (size_t)Mp * ldQ32 * sizeof(float), side));
        return 0;
    };
    auto side_prologue = [&]() -> int {      // (queued BEHIND the chain's launches on the host, see above; it only waits for ev_fork on the device)
        if (!overlap) return 0;
        STEP_HIP(hipStreamWaitEvent(side, pl->ev_fork, 0));
        if (pl->precleared && !s_late) {       // (under the chain: measured level at M' = 3000 -- the fills cost the chain what they save later)
            STEP_CALL(clear_tail_targets());
            STEP_HIP(hipMemsetAsync(Ge, 0, (size_t)(Mp + 1) * Mp * sizeof(float), side));
            pl->precleared_ge = true;
        }
        ctx->stream = side;
        const int rc = prologue(true);
        ctx->stream = main;
        if (rc) return rc;
        STEP_HIP(hipEventRecord(pl->ev_side, side));              // (K_ZX and [S - I | m'] are final)
        return 0;
    };
    if (pipe) {
        const int k1 = pk1, k2 = pk2;
        const int r1 = (k1 + 1) * 64;                                              // (< M': k <= nblk64 - 2)
        r2 = (k2 + 1) * 64;
        const PotrfHook hooks[2] = {{k1, pl->ev_pipe1}, {k2, pl->ev_pipe2}};
        double* Dinv = (double*)trsm_ws;
        STEP_CALL(launch_potrf_blocked(main, L, Mp, Mp, info, (double*)potrf_ws, Dinv, Mp, Dinv + (size_t)Mp * Mp, ctx->prezeroed, hooks,
                                       k2 > k1 ? 2 : 1));
        STEP_CALL(side_prologue());
        ctx->stream = side;
        STEP_HIP(hipStreamWaitEvent(side, pl->ev_pipe1, 0));
        STEP_TIME(0);
        STEP_CALL(solve_rows(0, r1, pipe_pad));
        if (k2 > k1) {
            STEP_HIP(hipStreamWaitEvent(side, pl->ev_pipe2, 0));
            STEP_CALL(solve_rows(r1, r2, pipe_pad));
        } else r2 = r1;
        STEP_TIME(1);
        STEP_HIP(hipEventRecord(pl->ev_pipe1, side));             // (re-used: the side stream's pieces are done)
        ctx->stream = main;
    } else {
        STEP_CALL(dsvgp_potrf_inverse(ctx, L, Mp, Mp, info, potrf_ws, nb, trsm_ws));
        STEP_CALL(side_prologue());
    }
    if (overlap) {
        // the status word's copy to the host leaves the main stream (a 5 us blit between the chain and the solve at M' = 600): the side
        // stream makes it behind an event the main stream records after the factorisation
        STEP_HIP(hipStreamWaitEvent(main, pl->ev_side, 0));
        STEP_HIP(hipEventRecord(pl->ev_fork2, main));
        STEP_HIP(hipStreamWaitEvent(side, pl->ev_fork2, 0));
        STEP_HIP(hipMemcpyAsync(pl->host_status, hyp, 5 * sizeof(float), hipMemcpyDeviceToHost, side));     // hyp[4] | info: contiguous
        STEP_HIP(hipEventRecord(pl->ev_status, side));
        if (s_late) {           // [S - I | m'] beside the forward solve (see prologue_s above); in front of it the fills of `preclear`
            if (pl->precleared) STEP_CALL(clear_tail_targets());     // (ordered by the wait for ev_s at the end of this function; [tril(G) ; b^T]
            ctx->stream = side;                                      //  is needed before that: its launcher keeps its own fill)
            const int rc = prologue_s(s_late == 1);
            ctx->stream = main;
            if (rc) return rc;
            STEP_HIP(hipEventRecord(pl->ev_s, side));
        }
    } else {
        STEP_HIP(hipMemcpyAsync(pl->host_status, hyp, 5 * sizeof(float), hipMemcpyDeviceToHost, main));
        STEP_HIP(hipEventRecord(pl->ev_status, main));
        STEP_CALL(prologue(false));
    }
    // ---- A = L^-1 K_ZX (fp64 product with the explicit inverse, fp32 result), mu = A^T m + c, residuals (DGVS.py:181-188)
    if (pipe) {
        STEP_CALL(solve_rows(r2, Mp, 0));
        STEP_HIP(hipStreamWaitEvent(main, pl->ev_pipe1, 0));      // (the side stream's row ranges: joined behind the main stream's own)
    } else {
        STEP_TIME(0);
        STEP_CALL(dsvgp_trsm(ctx, L, Mp, Mp, 0, Kzx, Bp, 0, Bp, nullptr, 0, A32, Bp, nb, trsm_ws, 1));
        STEP_TIME(1);
    }
    STEP_CALL(launch_stats_residual(ctx, A32, Bp, Mp, Bp, p, io->m, io->constant, hyp, io->mu, var0, stats_ws, io->y, rows, mu_bar, sums));
    // ---- [tril(G) ; b^T] = tril([A ; mu_bar^T] A^T), split-K over the minibatch axis; G mirrored
    STEP_TIME(6);
    if (flags & 32) {
        // opt-in: six bf16 products per fp32 product on the bf16 matrix pipe (gemm3b.hip); the planes of [A ; mu_bar^T] serve both operands
        if (!io->split_ws || ((uintptr_t)io->split_ws & 255) || io->split_ws_bytes < split_offsets(Mp, Bp, nullptr, nullptr)) {
            ctx->stream = main;
            return DSVGP_EINVAL;
        }
        STEP_CALL(dsvgp_split3_bf16(ctx, A32e, Bp, Mp + 1, Bp, 0, io->split_ws));
        STEP_CALL(dsvgp_gemm3b(ctx, DSVGP_GEMM_OUT_LOWER, Mp + 1, Mp, Bp, 1.f, io->split_ws, Mp + 1, io->split_ws, Mp + 1, Ge, Mp));
    } else {
        ZeroedScope zs(ctx, pl->precleared_ge);
        STEP_CALL(dsvgp_gemm(ctx, 0, DSVGP_GEMM_TRANS_B | DSVGP_GEMM_OUT_LOWER, Mp + 1, Mp, Bp, 1.0, A32e, Bp, A32, Bp, 0.0, nullptr, 0, Ge,
                             Mp, nullptr, 0, nullptr));
    }
    STEP_TIME(7);
    if (s_late) STEP_HIP(hipStreamWaitEvent(main, pl->ev_s, 0));       // ([S - I | m'] is final: every later reader is behind this point)
    return 0;
}

extern "C" int dsvgp_elbo_step_f32(dsvgp_ctx* ctx, dsvgp_step_plan* pl, const dsvgp_elbo_step_io* io, void* workspace,
                                   size_t workspace_bytes, int flags) {
    if (int rc = step_validate(ctx, pl, io, workspace, workspace_bytes)) return rc;
    if (pl->world != 1) return DSVGP_EINVAL;          // (a data-parallel rank's plan: dsvgp_elbo_step_dp_f32)
    STEP_LOCALS
    const bool overlap = (flags & 1) && !ctx->det_slab;           // (deterministic mode: the scratch serves one stream)
    const bool include_kl = flags & 2, timed = flags & 4;
    const hipStream_t main = ctx->stream, side = pl->side;
    struct PrezeroGuard {               // (the flag must not outlive the call, whatever path returns)
        dsvgp_ctx* c; bool prev;
        PrezeroGuard(dsvgp_ctx* c_, bool on) : c(c_), prev(c_->prezeroed) { c->prezeroed = on; }
        ~PrezeroGuard() { c->prezeroed = prev; }
    } prezero_guard(ctx, pl->arena_bytes <= ((size_t)48 << 20) && !ctx->det_slab);
    STEP_CALL(step_front(ctx, pl, io, workspace, flags, true));
    float* A32 = A32e;
    STEP_CALL(dsvgp_mirror_lower_f32(ctx, Ge, Mp, Mp));
    // ---- variational block (needs only G): L_S-bar = 2 vbar tril(G L_S) + KL gradient, m-bar = b + KL gradient, trace terms, scalars
    auto variational = [&]() -> int {
        // (tril(G L_S) stays on the K-trimmed register-staged kernel of gemm.hip: on the LDS-DMA kernel -- against a tril-clean copy of
        //  L_S, twice the multiply-adds at five times the rate -- the step got SLOWER, C4 12.91 -> 13.02 ms, C3 7.20 -> 7.34: this
        //  product runs beside the [Q' | a] solve, and the DMA kernel's two-per-CU workgroups crowd that solve out of the CUs)
        int rc = dsvgp_gemm(ctx, 0, DSVGP_GEMM_B_LOWER | DSVGP_GEMM_OUT_LOWER, Mp, Mp, Mp, 1.0, Ge, Mp, io->LS, io->ldls, 0.0, nullptr, 0,
                            io->dLS, io->lddls, nullptr, 0, nullptr);
        if (rc) return rc;
        // one pass + one reduction launch: m-bar = b + KL gradient (b = row M' of [G ; b^T], copied on the way), trace terms, scaling,
        // KL, and the scalar assembly of dsvgp_elbo_fast_finalize in the reduction's last thread
        return launch_variational_terms(ctx->stream, io->m, io->LS, io->ldls, Mp, io->num_data, 1 | (include_kl ? 2 : 0), hyp, rows, Ge, Mp,
                                        1.f, kl_buf, sums, Ge + (size_t)Mp * Mp, io->dm, io->dLS, io->lddls, B, p, scal);
    };
    // ---- [Q' | a / (2 vbar)] = L^-T [S - I | m / (2 vbar)] (fp64), K_ZX-bar = [Q' | a] [A ; mu_bar^T] (fp32, unscaled)
    const bool var_late = overlap && !(flags & 16) && Mp >= STEP_VAR_LATE_MP && (STEP_VAR_LATE == 2 || (STEP_VAR_LATE == 1 && Bp >= STEP_Q_CLASSIC_BP));
    auto solve_q = [&]() -> int {
        // (only the fp32 copy of [Q' | a] is read afterwards.  A large solve writes it directly; a small one -- fewer than 1024
        //  output tiles -- splits K onto an fp64 target: its own, Qe64, not the scratch the forward solve has used already)
        const bool small = (int64_t)((Mp + 63) / 64) * ((Mp + 1 + 63) / 64) < 1024;
        // (beside the side stream's G L_S and in front of a LARGE dense product -- B' >= STEP_Q_CLASSIC_BP, the C4 geometry -- the solve
        //  keeps the register-staged lean kernel: see dsvgp_ctx::lean_classic.  Measured: C4 12.31 against 12.42 ms; C3 and the
        //  8-rank share, whose dense products are small, gain 0.2 % / 1 % on the pipelined kernel.)
        const bool prev = ctx->lean_classic;
        ctx->lean_classic = overlap && !var_late && Bp >= STEP_Q_CLASSIC_BP;
        const int rc = dsvgp_trsm(ctx, L, Mp, Mp, 1, S32e, ldS, 0, Mp + 1, small ? Qe64 : nullptr, ldQ64, Qe32, ldQ32, nb, trsm_ws, 1);
        ctx->lean_classic = prev;
        return rc;
    };
    auto dense_product = [&]() -> int {
        if (flags & 32) {
            size_t o_pat, o_pq;
            split_offsets(Mp, Bp, &o_pat, &o_pq);
            char* sw = (char*)io->split_ws;
            int rc = dsvgp_split3_bf16(ctx, A32e, Bp, Mp + 1, Bp, 1, sw + o_pat);           // planes of [A ; mu_bar^T]^T: k-contiguous B operand
            if (rc) return rc;
            rc = dsvgp_split3_bf16(ctx, Qe32, ldQ32, Mp, Mp + 1, 0, sw + o_pq);
            if (rc) return rc;
            return dsvgp_gemm3b(ctx, 0, Mp, Bp, Mp + 1, 1.f, sw + o_pq, Mp, sw + o_pat, Bp, Kb32, Bp);
        }
        return dsvgp_gemm(ctx, 0, DSVGP_GEMM_K_PADDED, Mp, Bp, Mp + 1, 1.0, Qe32, ldQ32, A32e, Bp, 0.0, nullptr, 0, Kb32, Bp, nullptr, 0,
                          nullptr);
    };
    auto dense = [&]() -> int {
        if (timed && hipEventRecord(pl->tm[8], ctx->stream) != hipSuccess) return 1000 + (int)hipGetLastError();
        const int rc = dense_product();
        if (rc) return rc;
        if (timed && hipEventRecord(pl->tm[9], ctx->stream) != hipSuccess) return 1000 + (int)hipGetLastError();
        return 0;
    };
    // ---- Cholesky backward (DGVS.py:72-75 differentiated): K_ZZ-bar = 1/2 L^-T (Phi(L^T L-bar) + Phi(.)^T) L^-1 through the explicit
    // inverse, lower halves + mirrors.  Phi reads the lower triangle of L^T L-bar only, and there the tril() in
    // L-bar = -tril(K_ZX-bar A^T) = -tril([Q' | a][G ; b^T]) does not matter (row i of the upper-triangular L^T meets rows k >= i of
    // L-bar, column j <= i: entries on or below the diagonal), so with L^T Q' = S - I and L^T a = m / (2 vbar)
    //     tril(L^T L-bar) = -tril([S - I | m / (2 vbar)] [G ; b^T])
    // -- ONE product of two matrices the step already holds (fp64 accumulation, fp64 copy of the left operand from the prologue)
    // instead of L-bar (M'^3) followed by L^T L-bar (M'^3 / 3); neither L-bar nor the fp64 [Q' | a] is formed.  (Round 4.)
    const double* Linv = (const double*)trsm_ws;
    // Both operands of that product are fp32 DATA (S = L_S L_S^T and G = A A^T come out of fp32 MFMA products; the fp64 copy of
    // the left one is a widening), and G carries 2e-6 of its largest entry per element -- sqrt(M') times more than what fp32
    // accumulation of this product adds.  So the product itself runs on the fp32 LDS-DMA kernel (gemm32.hip, 2x the fp64 rate;
    // result into the [Q' | a] scratch, free once the dense product has read it) and only its RESULT is widened for the fp64
    // Cholesky backward.  STEP_PHI64 / flag 64 / shapes gemm32 does not take: the fp64-accumulated form.
    auto phi_arg = [&](bool qe32_free) -> int {
#if !STEP_PHI64
        const bool own = (ctx->prezeroed || pl->precleared) && pl->phi32_own;          // (its own, already cleared place: step_layout / preclear)
        if ((qe32_free || own) && !(flags & 64)) {
            GemmArgs g{};
            float* P32 = own ? (float*)(w + pl->o_phi32) : Qe32;
            g.M = Mp; g.N = Mp; g.K = Mp + 1; g.A = S32e; g.lda = ldS; g.B = Ge; g.ldb = Mp; g.C = P32; g.ldc = ldQ32;
            g.alpha = -1.0; g.beta = 0.0; g.batch = 1; g.splitk = 1;
            g.flags = DSVGP_GEMM_OUT_LOWER | DSVGP_GEMM_K_PADDED | DSVGP_GEMM_UPPER_UNDEF |     // (the widening below reads the lower triangle only: no zero fill of the rest)
                      (own ? DSVGP_GEMM_C_ZEROED : 0);
            g.slab = ctx->det_slab; g.slab_bytes = ctx->det_bytes;
            const int rc = launch_gemm32(ctx->stream, g);          // (clears the output itself when it is the [Q' | a] scratch: not part of the prezeroed arena)
            if (rc > 1) return rc;
            if (rc == 1) return launch_widen_sym_f32_f64(ctx->stream, P32, ldQ32, G1, Mp, Mp);   // widening + Phi(.) + Phi(.)^T in one pass
        }
#endif
        int rc = dsvgp_gemm(ctx, 1, DSVGP_GEMM_TRANS_A | DSVGP_GEMM_OUT_LOWER | DSVGP_GEMM_B_IS_FLOAT, Mp, Mp, Mp + 1, -1.0, S64e, ldST, Ge, Mp,
                            0.0, nullptr, 0, G1, Mp, nullptr, 0, nullptr);
        if (rc) return rc;
        return dsvgp_phi_symmetrize(ctx, G1, Mp, Mp);
    };
    auto chol_tail = [&](bool qe32_free) -> int {
        int rc = phi_arg(qe32_free);                                // S = Phi(P) + Phi(P)^T of P = tril(L^T L-bar), fp64, full
        if (rc) return rc;
        {
            ZeroedScope zs(ctx, pl->precleared);
            rc = dsvgp_gemm(ctx, 1, DSVGP_GEMM_TRANS_A | DSVGP_GEMM_B_LOWER | DSVGP_GEMM_OUT_LOWER, Mp, Mp, Mp, 1.0, G1, Mp, Linv, Mp, 0.0,
                            nullptr, 0, Yt, Mp, nullptr, 0, nullptr);
            if (rc) return rc;
            rc = dsvgp_gemm(ctx, 1, DSVGP_GEMM_TRANS_A | DSVGP_GEMM_A_UPPER | DSVGP_GEMM_OUT_LOWER, Mp, Mp, Mp, 0.5, Linv, Mp, Yt, Mp, 0.0,
                            nullptr, 0, Kbar, Mp, nullptr, 0, nullptr);
            if (rc) return rc;
        }
        return dsvgp_phi_symmetrize(ctx, Kbar, Mp, Mp);
    };
    // tail_side: the whole M'^3 tail (L-bar, Cholesky backward) follows the variational block on the side stream, under the dense
    // K_ZX-bar product on the main stream (it depends on [Q' | a] and G only); joined before K_ZZ-bar's kernel backward
    const bool tail_side = overlap && (flags & 16);
    if (overlap && var_late) {
        // the variational block (G L_S, the trace / KL pass) BEHIND the dense product, beside the Cholesky backward's few-tile fp64 products,
        // instead of beside the [Q' | a] solve: that solve then runs alone, on the pipelined lean kernel (STEP_VAR_LATE)
        STEP_CALL(solve_q());
        STEP_CALL(dense());
        STEP_HIP(hipEventRecord(pl->ev_fork2, main));
        STEP_HIP(hipStreamWaitEvent(side, pl->ev_fork2, 0));
        ctx->stream = side;
        STEP_CALL(variational());
        STEP_HIP(hipEventRecord(pl->ev_var, side));
        ctx->stream = main;
    } else if (overlap) {
        STEP_HIP(hipEventRecord(pl->ev_fork2, main));
        STEP_HIP(hipStreamWaitEvent(side, pl->ev_fork2, 0));
        ctx->stream = side;
        STEP_CALL(variational());
        STEP_HIP(hipEventRecord(pl->ev_var, side));
        ctx->stream = main;
        STEP_CALL(solve_q());
        if (tail_side) {
            STEP_HIP(hipEventRecord(pl->ev_dense, main));           // ([Q' | a] is final)
            STEP_HIP(hipStreamWaitEvent(side, pl->ev_dense, 0));
            ctx->stream = side;
            STEP_CALL(chol_tail(false));                            // ([Q' | a] is being read by the dense product beside it)
            STEP_HIP(hipEventRecord(pl->ev_zx, side));
            ctx->stream = main;
        }
        STEP_CALL(dense());
    } else {
        STEP_CALL(variational());
        STEP_CALL(solve_q());
        STEP_CALL(dense());
    }
    // ---- K_ZX-bar's kernel backward: beside the fp64 products that follow when the batch is small against M' (HBM-bound read)
#ifndef STEP_ZX_SIDE_MAX
#define STEP_ZX_SIDE_MAX 2          // K_ZX-bar's kernel backward goes to the side stream when B' <= STEP_ZX_SIDE_MAX * M'
#endif
    static const int zx_side_max = getenv("DSVGP_ZX_SIDE_MAX") ? atoi(getenv("DSVGP_ZX_SIDE_MAX")) : STEP_ZX_SIDE_MAX;
    const bool zx_side = overlap && !tail_side && (int64_t)Bp <= (int64_t)zx_side_max * Mp;
    // (round 6) the two kernel backwards leave their slabs (K_ZX-bar's in the second workspace, whichever stream it runs on) and ONE
    // kernel_bwd_points launch adds both, scales by 2 vbar and does the scalar tail: kernel_bwd_points x 2 + scale_epilogue before
    struct DeferGuard {
        dsvgp_ctx* c;
        DeferGuard(dsvgp_ctx* c_) : c(c_) { c->defer_points = true; c->n_deferred = 0; }
        ~DeferGuard() { c->defer_points = false; c->n_deferred = 0; }
    } defer_guard(ctx);
    if (zx_side) {
        STEP_HIP(hipEventRecord(pl->ev_dense, main));
        STEP_HIP(hipStreamWaitEvent(side, pl->ev_dense, 0));
        ctx->stream = side;
        STEP_TIME(4);
        STEP_CALL(zx_bwd(ctx, io, Kb32, Bp, PZ, sZ, vZ, M, PX, sX, B, d, p, hyp, kbwd_ws2));
        STEP_TIME(5);
        STEP_HIP(hipEventRecord(pl->ev_zx, side));
        ctx->stream = main;
    }
    if (!tail_side) {
        STEP_CALL(chol_tail(true));                                 // (its first product does not need the variational block)
        if (overlap) STEP_HIP(hipStreamWaitEvent(main, pl->ev_var, 0));
    }
    if (!zx_side) {
        STEP_TIME(4);
        STEP_CALL(zx_bwd(ctx, io, Kb32, Bp, PZ, sZ, vZ, M, PX, sX, B, d, p, hyp, kbwd_ws2));
        STEP_TIME(5);
    }
    if (zx_side || tail_side) STEP_HIP(hipStreamWaitEvent(main, pl->ev_zx, 0));
    STEP_CALL(zz_bwd(ctx, io, Kbar, Mp, PZ, sZ, vZ, M, PZ, sZ, M, d, p, hyp, kbwd_ws));
    // ---- both slab sets -> Z-bar, V-bar, the kernel hyper-parameter gradients; 2 vbar = 1 / (noise rows) on them (the products above ran
    // unscaled) and the scalar tail, in the same launch
    STEP_CALL(kernel_bwd_points_flush(ctx, PZ, vZ, M, d, p, hyp, io->dZ, io->dV, io->d_hyp, scal, kl_buf, rows, io->num_data,
                                      io->raw_lengthscale, io->raw_outputscale, io->raw_noise, io->d_raw_lengthscale, io->d_raw_outputscale,
                                      io->d_raw_noise, io->d_constant, io->loss));
    return 0;
}

// -------------------------------------------------------------------------------------------------------------------------
// The PER-OUTPUT step from one host call: the objective reads every output's own predictive variance -- mll_type = "PLL"
// (PredictiveLogLikelihood, reference directional_vi.py:218-219, what tests/test_grad_svgp.py:19-36 trains with) or the ELBO with
// per-output variances returned -- so the Gram formulation of dsvgp_elbo_step_f32 does not apply:
//   A = L^-1 K_ZX (fp64 + fp32 copy), W = L_S^T A, mu / var per output (DGVS.py:181-205), likelihood terms (mu-bar, var-bar),
//   U = L_S W, A-bar = m mu-bar^T + 2 (U - A) diag(var-bar), L_S-bar = tril(2 A diag(var-bar) W^T) + KL, m-bar = A mu-bar + KL,
//   K_ZX-bar = L^-T A-bar (fp64 + fp32 copy), L-bar = -tril(K_ZX-bar A^T) (fp64), the Cholesky backward and both kernel backwards.
// The same library calls as the Python-orchestrated path (`_step.ElboEngine._loss_and_grads`, ~100 ctypes calls: host-bound at the
// reference's own test sizes), queued here from C.  flags: bit 0 overlap (x pack + K_ZX assembly on the side stream under the
// Cholesky chain), bit 1 include the KL term, bit 8 (256) PLL objective (clear: ELBO).  varn [B(p+1)]: var + noise per output.
// -------------------------------------------------------------------------------------------------------------------------
extern "C" int dsvgp_elbo_step_po_f32(dsvgp_ctx* ctx, dsvgp_step_plan* pl, const dsvgp_elbo_step_io* io, float* varn, void* workspace,
                                      size_t workspace_bytes, int flags) {
    if (int rc = step_validate(ctx, pl, io, workspace, workspace_bytes)) return rc;
    if (pl->world != 1 || !pl->per_output || !varn) return DSVGP_EINVAL;
    STEP_LOCALS
    const bool overlap = (flags & 1) && !ctx->det_slab;
    const bool include_kl = flags & 2;
    const int mll_type = (flags & 256) ? 1 : 0;
    const hipStream_t main = ctx->stream, side = pl->side;
    double* A64 = (double*)(w + pl->o_A64);
    float* U32 = (float*)(w + pl->o_U32);
    float* Abar = (float*)(w + pl->o_Abar);
    double* Kb64 = (double*)(w + pl->o_Kb64);
    float* var = (float*)(w + pl->o_var);
    float* var_bar = (float*)(w + pl->o_varbar);
    float* W32 = Kzx;                                   // (K_ZX is dead once A has been formed)
    float* A32 = A32e;
    float* mu_bar = A32e + (size_t)Mp * Bp;
    double* Lbar = Qe64;                                // [M', M'] fp64 (the fast path's [Q' | a] scratch)
    pl->timed = false;
    struct PrezeroGuard {
        dsvgp_ctx* c; bool prev;
        PrezeroGuard(dsvgp_ctx* c_) : c(c_), prev(c_->prezeroed) { c->prezeroed = false; }
        ~PrezeroGuard() { c->prezeroed = prev; }
    } prezero_guard(ctx);
    STEP_HIP(hipMemsetAsync(io->flat, 0, io->flat_floats * sizeof(float), main));
    STEP_HIP(hipMemsetAsync(w + pl->o_zero, 0, pl->zero_bytes, main));
    STEP_CALL(launch_column_mean_hyp(main, io->Z, M, d, center, io->raw_lengthscale, io->raw_outputscale, io->raw_noise, hyp));
    STEP_CALL(dsvgp_pack_points(ctx, io->Z, io->V, M, d, p, hyp, center, PZ, sZ, vZ));
    auto prologue = [&]() -> int {
        int rc = dsvgp_pack_points(ctx, io->x, io->D, B, d, p, hyp, center, PX, sX, vX);
        if (rc) return rc;
        return zx_fwd(ctx, io, PZ, sZ, M, PX, sX, B, d, p, hyp, Kzx, Bp);
    };
    if (overlap) {
        STEP_HIP(hipEventRecord(pl->ev_fork, main));
        STEP_HIP(hipStreamWaitEvent(side, pl->ev_fork, 0));
        ctx->stream = side;
        STEP_CALL(prologue());
        STEP_HIP(hipEventRecord(pl->ev_side, side));
        ctx->stream = main;
    }
    STEP_CALL(zz_fwd(ctx, io, PZ, sZ, M, d, p, hyp, L, Mp));
    STEP_CALL(dsvgp_potrf_inverse(ctx, L, Mp, Mp, info, potrf_ws, nb, trsm_ws));
    STEP_HIP(hipMemcpyAsync(pl->host_status, hyp, 5 * sizeof(float), hipMemcpyDeviceToHost, main));
    STEP_HIP(hipEventRecord(pl->ev_status, main));
    if (overlap) STEP_HIP(hipStreamWaitEvent(main, pl->ev_side, 0));
    else STEP_CALL(prologue());
    // ---- forward (DGVS.py:181-205)
    STEP_CALL(dsvgp_trsm(ctx, L, Mp, Mp, 0, Kzx, Bp, 0, Bp, A64, Bp, A32, Bp, nb, trsm_ws, 1));
    STEP_CALL(dsvgp_gemm(ctx, 0, DSVGP_GEMM_TRANS_A | DSVGP_GEMM_A_UPPER, Mp, Bp, Mp, 1.0, io->LS, io->ldls, A32, Bp, 0.0, nullptr, 0, W32,
                         Bp, nullptr, 0, nullptr));                                                          // W = tril(L_S)^T A
    STEP_CALL(dsvgp_predictive_stats(ctx, A32, Bp, W32, Bp, Mp, Bp, p, io->m, io->constant, hyp, io->mu, var, stats_ws));
    STEP_CALL(dsvgp_likelihood_terms(ctx, io->mu, var, io->y, Bp, p, hyp, mll_type, rows, mu_bar, var_bar, varn, scal));
    // ---- variational parameters
    STEP_CALL(dsvgp_gemm(ctx, 0, DSVGP_GEMM_A_LOWER, Mp, Bp, Mp, 1.0, io->LS, io->ldls, W32, Bp, 0.0, nullptr, 0, U32, Bp, nullptr, 0,
                         nullptr));                                                                          // U = L_S W
    STEP_CALL(dsvgp_gemm(ctx, 0, DSVGP_GEMM_TRANS_B | DSVGP_GEMM_OUT_LOWER, Mp, Mp, Bp, 2.0, A32, Bp, W32, Bp, 0.0, nullptr, 0, io->dLS,
                         io->lddls, nullptr, 0, var_bar));                                                   // tril(2 A diag(var-bar) W^T)
    STEP_CALL(dsvgp_abar(ctx, A32, Bp, U32, Bp, Mp, Bp, io->m, mu_bar, var_bar, Abar, Bp));
    STEP_CALL(dsvgp_rowdot(ctx, A32, Bp, Mp, Bp, mu_bar, io->dm));                                            // m-bar += A mu-bar
    if (include_kl) STEP_CALL(dsvgp_kl_terms(ctx, io->m, io->LS, io->ldls, Mp, io->num_data, kl_buf, io->dm, io->dLS, io->lddls));
    // ---- through the triangular solve and the Cholesky factor (fp64)
    STEP_CALL(dsvgp_trsm(ctx, L, Mp, Mp, 1, Abar, Bp, 0, Bp, Kb64, Bp, Kb32, Bp, nb, trsm_ws, 1));            // K_ZX-bar = L^-T A-bar
    STEP_CALL(dsvgp_gemm(ctx, 1, DSVGP_GEMM_TRANS_B | DSVGP_GEMM_OUT_LOWER, Mp, Mp, Bp, -1.0, Kb64, Bp, A64, Bp, 0.0, nullptr, 0, Lbar, Mp,
                         nullptr, 0, nullptr));                                                              // L-bar = -tril(K_ZX-bar A^T)
    STEP_CALL(zx_bwd(ctx, io, Kb32, Bp, PZ, sZ, vZ, M, PX, sX, B, d, p, hyp, kbwd_ws));
    const double* Linv = (const double*)trsm_ws;
    STEP_CALL(dsvgp_gemm(ctx, 1, DSVGP_GEMM_TRANS_A | DSVGP_GEMM_A_UPPER | DSVGP_GEMM_B_LOWER | DSVGP_GEMM_OUT_LOWER, Mp, Mp, Mp, 1.0, L, Mp,
                         Lbar, Mp, 0.0, nullptr, 0, G1, Mp, nullptr, 0, nullptr));                            // tril(L^T L-bar)
    STEP_CALL(dsvgp_phi_symmetrize(ctx, G1, Mp, Mp));
    STEP_CALL(dsvgp_gemm(ctx, 1, DSVGP_GEMM_TRANS_A | DSVGP_GEMM_B_LOWER | DSVGP_GEMM_OUT_LOWER, Mp, Mp, Mp, 1.0, G1, Mp, Linv, Mp, 0.0, nullptr,
                         0, Yt, Mp, nullptr, 0, nullptr));
    STEP_CALL(dsvgp_gemm(ctx, 1, DSVGP_GEMM_TRANS_A | DSVGP_GEMM_A_UPPER | DSVGP_GEMM_OUT_LOWER, Mp, Mp, Mp, 0.5, Linv, Mp, Yt, Mp, 0.0, nullptr,
                         0, Kbar, Mp, nullptr, 0, nullptr));
    STEP_CALL(dsvgp_phi_symmetrize(ctx, Kbar, Mp, Mp));
    STEP_CALL(zz_bwd(ctx, io, Kbar, Mp, PZ, sZ, vZ, M, PZ, sZ, M, d, p, hyp, kbwd_ws));
    return dsvgp_step_epilogue(ctx, scal, kl_buf, rows, io->num_data, io->raw_lengthscale, io->raw_outputscale, io->raw_noise, io->d_hyp,
                               io->d_raw_lengthscale, io->d_raw_outputscale, io->d_raw_noise, io->d_constant, io->loss);
}

// -------------------------------------------------------------------------------------------------------------------------
// One RANK of a data-parallel job (global-Gram schedule with the replicated M'^3 stage
// sharded), queued in five pieces with the caller's collectives between them -- the same schedule `_step.ElboEngine` issues
// through ~120 ctypes calls:
//   phase 0  everything up to the Gram product of this rank's rows; [tril(G) | b] packed into dp->wire
//            -> caller: all-reduce(sum) of dp->wire                                                   (18 MB at M' = 3000)
//   phase 1  this rank's COLUMNS of [Q' | a] = L^-T [S - I | m / (2 vbar)] (fp64 product, fp32 copy in dp->q_local)
//                                                                                                     (needs L^-1 and S only)
//            -> caller: all-gather dp->q_local -> dp->q_all                                            (36 MB)
//   phase 2  (after the all-reduce) G global: mirror; L_S-bar, m-bar, KL, trace terms on the side stream (identical on every rank:
//            the G L_S product adds its K slices in a fixed order); this rank's rows of tril(L^T L-bar) =
//            -tril([S - I | m / (2 vbar)][G ; b^T]) (fp64 accumulation: the arithmetic of the one-GPU step), fp32 copy in dp->lbar_local
//            -> caller: all-gather dp->lbar_local -> dp->lbar_all                                      (36 MB)
//   phase 3  (after the first all-gather) [Q' | a] row-major; dense K_ZX-bar of this rank's rows; its kernel backward
//   phase 4  (after the second all-gather) Phi of the gathered matrix; this rank's column block of K_ZZ-bar = 1/2 L^-T Phi(.) L^-1 and its kernel
//            backward (row-side gradient doubled by symmetry), scaling by 2 vbar, scalar tail
//            -> caller: all-reduce(sum) of [Z-bar, V-bar, hyper-parameter gradients, loss]              (0.24 MB)
// flags as dsvgp_elbo_step_f32 (bit 1: this is the rank that counts the KL value and the trace terms of the global G).
// -------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_blocks_kernel(const float* __restrict__ src, float* __restrict__ dst, int rowsM, int wq,
                                                            int world) {
    // dst[m][g * wq + c] = src[g][m][c]: 16-byte pieces (wq is a multiple of 4)
    const int per_row = world * (wq / 4);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)rowsM * per_row; e += (int64_t)gridDim.x * 256) {
        const int m = (int)(e / per_row), r = (int)(e - (int64_t)m * per_row), g = r / (wq / 4), c4 = r - g * (wq / 4);
        reinterpret_cast<float4*>(dst)[e] = reinterpret_cast<const float4*>(src)[((int64_t)g * rowsM + m) * (wq / 4) + c4];
    }
}

extern "C" int dsvgp_elbo_step_dp_f32(dsvgp_ctx* ctx, dsvgp_step_plan* pl, const dsvgp_elbo_step_io* io, const dsvgp_elbo_step_dp* dp,
                                      void* workspace, size_t workspace_bytes, int flags, int phase) {
    if (int rc = step_validate(ctx, pl, io, workspace, workspace_bytes)) return rc;
    if (!dp || pl->world < 2 || dp->world != pl->world || dp->rank < 0 || dp->rank >= dp->world || phase < 0 || phase > 4 ||
        !dp->wire || !dp->q_local || !dp->q_all || !dp->lbar_local || !dp->lbar_all)
        return DSVGP_EINVAL;
    STEP_LOCALS
    const int world = pl->world, rank = dp->rank, wq = pl->wq, wr = pl->wr, q1 = p + 1;
    const int64_t wire_used = (int64_t)Mp * (Mp + 1) / 2 + Mp;
    if ((int64_t)dp->wire_floats < wire_used) return DSVGP_EINVAL;
    const bool overlap = (flags & 1) && !ctx->det_slab;
    const bool include_kl = flags & 2, timed = pl->timed;
    const hipStream_t main = ctx->stream, side = pl->side;
    float* Qfull = (float*)(w + pl->o_Qfull);
    double* qcol64 = (double*)(w + pl->o_qcol64);
    double* lrow64 = (double*)(w + pl->o_lrow64);
    double* cbT = (double*)(w + pl->o_cbT);
    double* cbK = (double*)(w + pl->o_cbK);
    const double* Linv = (const double*)trsm_ws;
    const int r0 = rank * wr < Mp ? rank * wr : Mp, r1 = r0 + wr < Mp ? r0 + wr : Mp, nrow = r1 - r0;
    struct PrezeroGuard {
        dsvgp_ctx* c; bool prev;
        PrezeroGuard(dsvgp_ctx* c_, bool on) : c(c_), prev(c_->prezeroed) { c->prezeroed = on; }
        ~PrezeroGuard() { c->prezeroed = prev; }
    } prezero_guard(ctx, false);             // (the launchers clear their own split-K targets: the arena is not cleared per phase)
    if (phase == 0) {
        // (front with flag 8 semantics kept; the arena-wide clear of small problems is off for data-parallel plans)
        const size_t keep = pl->arena_bytes;
        pl->arena_bytes = (size_t)1 << 60;
        const int rc = step_front(ctx, pl, io, workspace, flags);
        pl->arena_bytes = keep;
        if (rc) return rc;
        return dsvgp_tril_pack_f32(ctx, Ge, Mp, Mp, Ge + (size_t)Mp * Mp, Mp, dp->wire);
    }
    if (phase == 1) {
        const int c0 = rank * wq < Mp + 1 ? rank * wq : Mp + 1, c1 = c0 + wq < Mp + 1 ? c0 + wq : Mp + 1;
        if (c1 > c0)
            STEP_CALL(dsvgp_trsm(ctx, L, Mp, Mp, 1, S32e + c0, ldS, 0, c1 - c0, qcol64, wq, dp->q_local, wq, nb, trsm_ws, 1));
        return 0;
    }
    if (phase == 2) {
        STEP_CALL(dsvgp_tril_unpack_f32(ctx, dp->wire, Mp, Ge, Mp, Ge + (size_t)Mp * Mp, Mp));
        STEP_CALL(dsvgp_mirror_lower_f32(ctx, Ge, Mp, Mp));
        auto variational = [&]() -> int {
            // fixed-order sums for THIS product on every rank: the slab lives in the plan's workspace and serves this stream only
            void* const keep_slab = ctx->det_slab; const size_t keep_bytes = ctx->det_bytes;
            ctx->det_slab = w + pl->o_slab; ctx->det_bytes = pl->slab_bytes;
            int rc = dsvgp_gemm(ctx, 0, DSVGP_GEMM_B_LOWER | DSVGP_GEMM_OUT_LOWER, Mp, Mp, Mp, 1.0, Ge, Mp, io->LS, io->ldls, 0.0, nullptr,
                                0, io->dLS, io->lddls, nullptr, 0, nullptr);
            ctx->det_slab = keep_slab; ctx->det_bytes = keep_bytes;
            if (rc) return rc;
            // every rank adds the KL gradient (m-bar / L_S-bar are not reduced again); one rank counts the KL value + trace terms
            return launch_variational_terms(ctx->stream, io->m, io->LS, io->ldls, Mp, io->num_data, 1 | 2 | (include_kl ? 0 : 8), hyp,
                                            rows, Ge, Mp, 1.f, kl_buf, sums, Ge + (size_t)Mp * Mp, io->dm, io->dLS, io->lddls, B, p, scal);
        };
        if (overlap) {
            STEP_HIP(hipEventRecord(pl->ev_fork2, main));
            STEP_HIP(hipStreamWaitEvent(side, pl->ev_fork2, 0));
            ctx->stream = side;
            STEP_CALL(variational());
            STEP_HIP(hipEventRecord(pl->ev_var, side));
            ctx->stream = main;
        } else {
            STEP_CALL(variational());
        }
        if (nrow > 0)     // rows [r0, r1) of tril(L^T L-bar) = -tril([S - I | m / (2 vbar)][G ; b^T]) (see chol_tail of the one-GPU step),
            // columns [0, r1): fp64 accumulation, fp32 copy for the all-gather (the rest of dp->lbar_local stays zero)
            STEP_CALL(dsvgp_gemm(ctx, 1, DSVGP_GEMM_TRANS_A | DSVGP_GEMM_B_IS_FLOAT, nrow, r1, Mp + 1, -1.0, S64e + r0, ldST, Ge, Mp, 0.0,
                                 nullptr, 0, lrow64, Mp, dp->lbar_local, Mp, nullptr));
        return 0;
    }
    if (phase == 3) {
        {
            const int64_t n4 = (int64_t)Mp * world * (wq / 4);
            const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
            hipLaunchKernelGGL(gather_blocks_kernel, dim3(blocks), dim3(256), 0, main, dp->q_all, Qfull, Mp, wq, world);
            STEP_HIP(hipGetLastError());
        }
        STEP_TIME(8);
        if (flags & 32) {
            size_t o_pat, o_pq;
            split_offsets(Mp, Bp, &o_pat, &o_pq);
            char* sw = (char*)io->split_ws;
            STEP_CALL(dsvgp_split3_bf16(ctx, A32e, Bp, Mp + 1, Bp, 1, sw + o_pat));
            STEP_CALL(dsvgp_split3_bf16(ctx, Qfull, (int64_t)world * wq, Mp, Mp + 1, 0, sw + o_pq));
            STEP_CALL(dsvgp_gemm3b(ctx, 0, Mp, Bp, Mp + 1, 1.f, sw + o_pq, Mp, sw + o_pat, Bp, Kb32, Bp));
        } else
        STEP_CALL(dsvgp_gemm(ctx, 0, DSVGP_GEMM_K_PADDED, Mp, Bp, Mp + 1, 1.0, Qfull, (int64_t)world * wq, A32e, Bp, 0.0, nullptr, 0, Kb32,
                             Bp, nullptr, 0, nullptr));
        STEP_TIME(9);
        STEP_TIME(4);
        STEP_CALL(zx_bwd(ctx, io, Kb32, Bp, PZ, sZ, vZ, M, PX, sX, B, d, p, hyp, kbwd_ws));
        STEP_TIME(5);
        return 0;
    }
    // ---- phase 4: this rank's inducing points [m0, m1) = columns [m0 q, m1 q) of the symmetric K_ZZ-bar
    const int base = M / world, rem = M % world;
    const int m0 = rank * base + (rank < rem ? rank : rem), m1 = m0 + base + (rank < rem ? 1 : 0);
    const int c0 = m0 * q1, wcol = (m1 - m0) * q1;
    launch_widen_f32_f64(main, dp->lbar_all, Mp, G1, Mp, Mp, Mp);                                                       // tril(L^T L-bar), gathered
    STEP_HIP(hipGetLastError());
    STEP_CALL(dsvgp_phi_symmetrize(ctx, G1, Mp, Mp));                                                                   // S = Phi(.) + Phi(.)^T
    if (wcol > 0) {
        // S L^-1[:, c0 : c0 + wcol]: rows < c0 of that column block of the lower-triangular inverse are zero, the rest is lower
        // triangular in its own coordinates (S symmetric: read through its transpose)
        STEP_CALL(dsvgp_gemm(ctx, 1, DSVGP_GEMM_TRANS_A | DSVGP_GEMM_B_LOWER, Mp, wcol, Mp - c0, 1.0, G1 + (size_t)c0 * Mp, Mp,
                             Linv + (size_t)c0 * Mp + c0, Mp, 0.0, nullptr, 0, cbT, wcol, nullptr, 0, nullptr));
        STEP_CALL(dsvgp_gemm(ctx, 1, DSVGP_GEMM_TRANS_A | DSVGP_GEMM_A_UPPER, Mp, wcol, Mp, 0.5, Linv, Mp, cbT, wcol, 0.0, nullptr, 0, cbK,
                             wcol, nullptr, 0, nullptr));
    }
    if (overlap) STEP_HIP(hipStreamWaitEvent(main, pl->ev_var, 0));
    if (wcol > 0)
        STEP_CALL(zz_bwd(ctx, io, cbK, wcol, PZ, sZ, vZ, M, PZ + (size_t)c0 * pl->DP, sZ + c0, m1 - m0, d, p, hyp, kbwd_ws));
    STEP_CALL(launch_scale_epilogue(ctx, io->dZ, (int64_t)M * d, p > 0 ? io->dV : nullptr, p > 0 ? (int64_t)M * p * d : 0, hyp, rows, scal,
                                    kl_buf, io->num_data, io->raw_lengthscale, io->raw_outputscale, io->raw_noise, io->d_hyp,
                                    io->d_raw_lengthscale, io->d_raw_outputscale, io->d_raw_noise, io->d_constant, io->loss));
    return 0;
}
