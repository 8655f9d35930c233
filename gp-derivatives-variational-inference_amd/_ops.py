"""Tensor-level wrappers over the C ABI (include/dsvgp.h).

torch is used only as the device allocator / stream provider: every function here checks device,
dtype and contiguity, then hands raw device pointers to libdsvgp_hip.so.  There is no CPU path.
"""
import ctypes as C

import os
import torch

from . import _lib
from ._lib import check, lib

f32, f64 = torch.float32, torch.float64


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _req(t, dtype, name, dims=None):
    if not t.is_cuda:
        raise _lib.DsvgpError("%s must live on the GPU: the DSVGP hot path has no CPU fallback" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if dims is not None and t.dim() != dims:
        raise ValueError("%s must be %d-D" % (name, dims))
    if t.dim() == 2:
        if t.stride(1) != 1 and t.shape[1] > 1:
            raise ValueError("%s must be row-major with unit column stride" % name)
    elif not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def _ld(t):
    return int(t.stride(0)) if t.shape[0] > 1 else int(max(t.shape[1], t.stride(0)))


class Context:
    """One dsvgp_ctx per device; binds the library to torch's current HIP stream."""
    _cache = {}

    def __init__(self, device):
        self.device = torch.device(device)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.dsvgp_create(C.byref(h)), "dsvgp_create")
        self.h = h
        self._stream = None

    @classmethod
    def get(cls, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.DsvgpError("DSVGP HIP path needs a GPU device, got %s" % device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if idx not in cls._cache:
            cls._cache[idx] = Context(torch.device("cuda", idx))
        ctx = cls._cache[idx]
        ctx.bind()
        return ctx

    def set_deterministic(self, scratch):
        """scratch: a uint8 device tensor (split-K slabs / partial sums go there, no floating-point atomics) or None (atomics)"""
        if scratch is None:
            check(lib.dsvgp_set_deterministic(self.h, None, 0), "dsvgp_set_deterministic")
        else:
            check(lib.dsvgp_set_deterministic(self.h, _ptr(scratch), scratch.numel() * scratch.element_size()), "dsvgp_set_deterministic")

    def bind(self):
        # the library launches on this context's stream without a device guard of its own: make its device current
        # (one process per GPU is the intended use; torch.cuda.set_device(local_rank) has normally done this already)
        if torch.cuda.current_device() != self.device.index:
            torch.cuda.set_device(self.device)
        s = torch.cuda.current_stream(self.device).cuda_stream
        if s != self._stream:
            check(lib.dsvgp_set_stream(self.h, C.c_void_p(s)), "dsvgp_set_stream")
            self._stream = s


class StepPlan:
    """``dsvgp_step_plan`` of one (M, d, p, B): host-side object of the one-call ELBO step (csrc/step.hip)"""

    def __init__(self, ctx, M, d, p, B, world=1, per_output=False):
        h = C.c_void_p()
        self.per_output = bool(per_output)
        if per_output:      # dsvgp_elbo_step_po_f32: PLL objective / per-output variances (one rank)
            if world != 1:
                raise ValueError("the per-output step plan is a one-rank plan")
            check(lib.dsvgp_elbo_step_po_plan_create(ctx.h, int(M), int(d), int(p), int(B), C.byref(h)), "dsvgp_elbo_step_po_plan_create")
        elif world > 1:     # one rank of a data-parallel job (B = this rank's rows)
            check(lib.dsvgp_elbo_step_dp_plan_create(ctx.h, int(M), int(d), int(p), int(B), int(world), C.byref(h)),
                  "dsvgp_elbo_step_dp_plan_create")
            self.dp = _lib.ElboStepDP()
        else:
            check(lib.dsvgp_elbo_step_plan_create(ctx.h, int(M), int(d), int(p), int(B), C.byref(h)), "dsvgp_elbo_step_plan_create")
        self.world = int(world)
        self.h = h
        self.bytes = int(lib.dsvgp_elbo_step_plan_bytes(h))
        self.io = _lib.ElboStepIO()
        self._hyp = (C.c_float * 4)()
        self._info = C.c_int(0)
        self._ms = (C.c_float * 5)()

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            lib.dsvgp_elbo_step_plan_destroy(h)

    def run(self, ctx, workspace, flags):
        check(lib.dsvgp_elbo_step_f32(ctx.h, self.h, C.byref(self.io), _ptr(workspace), workspace.numel(), int(flags)),
              "dsvgp_elbo_step_f32")

    def run_po(self, ctx, workspace, flags, varn):
        check(lib.dsvgp_elbo_step_po_f32(ctx.h, self.h, C.byref(self.io), _ptr(varn), _ptr(workspace), workspace.numel(), int(flags)),
              "dsvgp_elbo_step_po_f32")

    def run_dp(self, ctx, workspace, flags, phase):
        check(lib.dsvgp_elbo_step_dp_f32(ctx.h, self.h, C.byref(self.io), C.byref(self.dp), _ptr(workspace), workspace.numel(),
                                         int(flags), int(phase)), "dsvgp_elbo_step_dp_f32 (phase %d)" % phase)

    def status(self):
        """(potrf status word, [lengthscale, outputscale, noise]) of the step queued last; waits for its factorisation only"""
        check(lib.dsvgp_elbo_step_status(self.h, self._hyp, C.byref(self._info)), "dsvgp_elbo_step_status")
        return int(self._info.value), [float(v) for v in self._hyp]

    def locate(self, workspace, which):
        """torch view of an intermediate of the step queued last inside ``workspace`` (dsvgp_elbo_step_locate)"""
        off, rows, cols, ld = C.c_size_t(0), C.c_int(0), C.c_int(0), C.c_int64(0)
        check(lib.dsvgp_elbo_step_locate(self.h, int(which), C.byref(off), C.byref(rows), C.byref(cols), C.byref(ld)),
              "dsvgp_elbo_step_locate")
        dt = torch.int32 if which == 5 else f32 if which in (0, 1, 4) else torch.float64
        esz = 4 if dt in (f32, torch.int32) else 8
        flat = workspace[off.value:off.value + rows.value * ld.value * esz].view(dt)
        return flat.view(rows.value, ld.value)[:, :cols.value]

    def timed_count(self):
        return int(lib.dsvgp_elbo_step_timed_count(self.h))

    def timings(self, back=0):
        """[solve_fwd, assemble_fwd, assemble_bwd, gram, dense] HIP-event durations (ms) of the timed step ``back`` steps before the last"""
        check(lib.dsvgp_elbo_step_timings5(self.h, int(back), self._ms), "dsvgp_elbo_step_timings5")
        return [float(v) for v in self._ms]


def step_supported(M, d, p, B, world=1, per_output=False):
    if per_output:
        return world == 1 and int(lib.dsvgp_elbo_step_po_workspace_bytes(int(M), int(d), int(p), int(B))) > 0
    if world > 1:
        return int(lib.dsvgp_elbo_step_dp_workspace_bytes(int(M), int(d), int(p), int(B), int(world))) > 0
    return int(lib.dsvgp_elbo_step_workspace_bytes(int(M), int(d), int(p), int(B))) > 0


def packed_width(d):
    return int(lib.dsvgp_packed_width(int(d)))


def hyp_forward(ctx, raw_l, raw_s, raw_n):
    hyp = torch.empty(4, dtype=f32, device=raw_l.device)
    check(lib.dsvgp_hyp_forward(ctx.h, _ptr(_req(raw_l.reshape(-1), f32, "raw_lengthscale")),
                                _ptr(_req(raw_s.reshape(-1), f32, "raw_outputscale")),
                                _ptr(_req(raw_n.reshape(-1), f32, "raw_noise")), _ptr(hyp)), "dsvgp_hyp_forward")
    return hyp


def hyp_backward(ctx, raw_l, raw_s, raw_n, d_hyp, d_raw_l, d_raw_s, d_raw_n):
    check(lib.dsvgp_hyp_backward(ctx.h, _ptr(raw_l), _ptr(raw_s), _ptr(raw_n), _ptr(d_hyp), _ptr(d_raw_l),
                                 _ptr(d_raw_s), _ptr(d_raw_n)), "dsvgp_hyp_backward")


def column_mean(ctx, x):
    """Column means of x[n, d]: the common shift ``center`` of the two point sets of one kernel call."""
    _req(x, f32, "x", 2)
    if not x.is_contiguous():
        raise ValueError("x must be contiguous")
    out = torch.empty(x.shape[1], dtype=f32, device=x.device)
    check(lib.dsvgp_column_mean(ctx.h, _ptr(x), x.shape[0], x.shape[1], _ptr(out)), "dsvgp_column_mean")
    return out


def step_epilogue(ctx, scal, kl0, rows, num_data, raw_l, raw_s, raw_n, d_hyp, d_raw_l, d_raw_s, d_raw_n, d_const, loss):
    check(lib.dsvgp_step_epilogue(ctx.h, _ptr(scal), _ptr(kl0), float(rows), float(num_data), _ptr(raw_l), _ptr(raw_s),
                                  _ptr(raw_n), _ptr(d_hyp), _ptr(d_raw_l), _ptr(d_raw_s), _ptr(d_raw_n), _ptr(d_const),
                                  _ptr(loss)), "dsvgp_step_epilogue")


def pack_points(ctx, x, v, p, hyp, center=None):
    """-> (P[n(p+1), DP], self[n(p+1)], vnorm[n p]).  ``center`` [d]: shift subtracted from x (same for both
    operands of a kernel call)."""
    _req(x, f32, "x", 2)
    n, d = x.shape
    if center is not None:
        _req(center, f32, "center", 1)
        if center.shape != (d,):
            raise ValueError("center must have shape [%d]" % d)
    if p > 0:
        _req(v, f32, "v", 2)
        if v.shape != (n * p, d):
            raise ValueError("directions must be [n*p, d] = [%d, %d], got %s" % (n * p, d, tuple(v.shape)))
    if not x.is_contiguous() or (p > 0 and not v.is_contiguous()):
        raise ValueError("x and v must be contiguous")
    DP = packed_width(d)
    P = torch.empty(n * (p + 1), DP, dtype=f32, device=x.device)
    sf = torch.empty(n * (p + 1), dtype=f32, device=x.device)
    vn = torch.empty(max(n * p, 1), dtype=f32, device=x.device)
    check(lib.dsvgp_pack_points(ctx.h, _ptr(x), _ptr(v if p > 0 else None), n, d, p, _ptr(hyp), _ptr(center), _ptr(P),
                                _ptr(sf), _ptr(vn)), "dsvgp_pack_points")
    return P, sf, vn


def kernel_fwd(ctx, pack1, n1, pack2, n2, d, p, hyp, jitter=0.0, out=None, dtype=f32):
    P1, s1 = pack1[0], pack1[1]
    P2, s2 = pack2[0], pack2[1]
    q = p + 1
    if out is None:
        out = torch.empty(n1 * q, n2 * q, dtype=dtype, device=P1.device)
    _req(out, dtype, "out", 2)
    if out.shape != (n1 * q, n2 * q):
        raise ValueError("out has shape %s, expected %s" % (tuple(out.shape), (n1 * q, n2 * q)))
    check(lib.dsvgp_kernel_fwd(ctx.h, _ptr(P1), _ptr(s1), n1, _ptr(P2), _ptr(s2), n2, d, p, _ptr(hyp), float(jitter),
                               _ptr(out), _ld(out), 1 if dtype == f64 else 0), "dsvgp_kernel_fwd")
    return out


def canon_supported(d, p):
    """geometries the canonical-direction assembly kernels take (csrc/assemble.hip): p + 1 in {3, 6}, packed width <= 32"""
    return p >= 1 and bool(lib.dsvgp_kernel_canon_supported(int(d), int(p)))


def state_directions(D, idx, base=0):
    """The caller's statement that the direction matrix ``D`` [B p, d] is one-hot and shared by all points: row j p + b = e_{idx[b] - base}
    for every j (what the reference's training step and evaluation build: directional_vi.py:81-88, 238, 292-294).  ``idx``: int32
    tensor [p] on D's device.  The statement travels with the tensor (``model(x, derivative_directions=D)`` is unchanged); the step
    then assembles K_ZX and its backward on the canonical-direction kernels where they take the geometry.  Returns D."""
    if D is not None and idx is not None and not _NO_CANON:
        D._dsvgp_dir_idx = (idx, int(base))
    return D


_RANGES = {}
_NO_CANON = os.environ.get("DSVGP_NO_CANON") == "1"      # tools: ignore the statements (the general assembly kernels everywhere: A/B runs)


def index_range(device, n):
    """the cached int32 tensor [0, 1, .., n - 1] on ``device`` (ONE object per device and length: statements that name all coordinates in
    order -- eye(d)[:p] tiled -- can be recognised as equal by identity)"""
    key = (str(device), int(n))
    t = _RANGES.get(key)
    if t is None:
        t = _RANGES[key] = torch.arange(int(n), dtype=torch.int32, device=device)
    return t


def same_statement(st_a, st_b):
    """whether two (idx, base) statements name the same list without reading device memory: the same tensor object / storage and base"""
    return (st_a is not None and st_b is not None and st_a[1] == st_b[1] and st_a[0].data_ptr() == st_b[0].data_ptr()
            and st_a[0].numel() == st_b[0].numel())


def detach_keep(t):
    """``t.detach()`` with the caller's statement about a direction matrix (state_directions) carried over to the new tensor object"""
    out = t.detach()
    st = getattr(t, "_dsvgp_dir_idx", None)
    if st is not None:
        out._dsvgp_dir_idx = st
    return out


def stated_directions(D, d, p, supported=None):
    """(idx, base) when ``D`` carries a usable statement (state_directions) for a geometry the canonical kernels take, else None.
    DSVGP_CHECK_DIRS=1: verify the statement against D (a device read: tests / debugging)."""
    st = getattr(D, "_dsvgp_dir_idx", None) if D is not None else None
    if st is None or p < 1 or _NO_CANON or not (supported or canon_supported)(d, p):
        return None
    idx, base = st
    if not (torch.is_tensor(idx) and idx.dtype == torch.int32 and idx.is_cuda and idx.numel() == p and idx.is_contiguous()
            and idx.device == D.device):
        return None
    if os.environ.get("DSVGP_CHECK_DIRS") == "1":
        E = torch.eye(d, device=D.device, dtype=D.dtype)[(idx.long() - base)]
        if not torch.equal(D.reshape(-1, p, d), E.expand(D.shape[0] // p, p, d)):
            raise _lib.DsvgpError("state_directions: D is not the one-hot matrix the index list states")
    return idx, int(base)


def canon2_supported(d, p):
    """geometries the both-sides one-hot assembly kernels take (csrc/assemble.hip: p + 1 = 11, d <= 12 -- the full-gradient SVGP at d = 10)"""
    return p >= 1 and bool(lib.dsvgp_kernel_canon2_supported(int(d), int(p)))


def _req_idx(dir_idx, p):
    if dir_idx.dtype != torch.int32 or not dir_idx.is_cuda or dir_idx.numel() != p or not dir_idx.is_contiguous():
        raise ValueError("dir_idx must be a contiguous int32 GPU tensor with p entries")


def kernel_fwd_canon2(ctx, pack1, n1, pack2, n2, d, p, dir_idx, idx_base, hyp, jitter=0.0, out=None, dtype=None):
    """K(x1, x2; E, E), E = the unit vectors e_{dir_idx - idx_base}, on both sides (GradVariationalStrategy.py:89-99 for dir_idx = 0..d-1)"""
    q = p + 1
    if out is None:
        out = torch.empty(n1 * q, n2 * q, dtype=dtype or f32, device=pack1[0].device)
    if out.dtype not in (f32, torch.float64) or out.dim() != 2 or out.stride(1) != 1 or not out.is_cuda:
        raise ValueError("out must be a float32 / float64 GPU matrix with unit inner stride")
    _req_idx(dir_idx, p)
    check(lib.dsvgp_kernel_fwd_canon2(ctx.h, _ptr(pack1[0]), n1, _ptr(pack2[0]), n2, d, p, _ptr(dir_idx), int(idx_base), _ptr(hyp),
                                      float(jitter), _ptr(out), _ld(out), 1 if out.dtype == torch.float64 else 0), "dsvgp_kernel_fwd_canon2")
    return out


def kernel_bwd_canon2(ctx, G, pack1, n1, pack2, n2, d, p, dir_idx, idx_base, hyp, symmetric, d_x1, d_v1, d_hyp, workspace=None):
    """backward of kernel_fwd_canon2: += d_x1, d_hyp[0..1] (d_v1 untouched: the directions are fixed)"""
    isd = G.dtype == torch.float64
    if G.dtype not in (f32, torch.float64) or G.dim() != 2 or G.stride(1) != 1 or not G.is_cuda:
        raise ValueError("G must be a float32 / float64 GPU matrix with unit inner stride")
    _req_idx(dir_idx, p)
    if workspace is None:
        workspace = torch.empty(int(lib.dsvgp_kernel_bwd_workspace_bytes(n1, n2, d, p)), dtype=torch.uint8, device=G.device)
    check(lib.dsvgp_kernel_bwd_canon2(ctx.h, _ptr(G), _ld(G), 1 if isd else 0, _ptr(pack1[0]), _ptr(pack1[2]), n1, _ptr(pack2[0]), n2, d, p,
                                      _ptr(dir_idx), int(idx_base), _ptr(hyp), 1 if symmetric else 0, _ptr(d_x1), _ptr(d_v1), _ptr(d_hyp),
                                      _ptr(workspace)), "dsvgp_kernel_bwd_canon2")


def kernel_fwd_canon(ctx, pack1, n1, pack2, n2, d, p, dir_idx, idx_base, hyp, out=None):
    """K(x1, x2; v1, E[dir_idx - idx_base]) with canonical (one-hot) directions on side 2 shared by all its points (the reference's K_ZX:
    directional_vi.py:81-88, 238, 292-294).  dir_idx: int32 device tensor with p entries."""
    q = p + 1
    if out is None:
        out = torch.empty(n1 * q, n2 * q, dtype=f32, device=pack1[0].device)
    _req(out, f32, "out", 2)
    if dir_idx.dtype != torch.int32 or not dir_idx.is_cuda or dir_idx.numel() != p or not dir_idx.is_contiguous():
        raise ValueError("dir_idx must be a contiguous int32 GPU tensor with p entries")
    check(lib.dsvgp_kernel_fwd_canon(ctx.h, _ptr(pack1[0]), _ptr(pack1[1]), n1, _ptr(pack2[0]), _ptr(pack2[1]), n2, d, p, _ptr(dir_idx),
                                     int(idx_base), _ptr(hyp), _ptr(out), _ld(out)), "dsvgp_kernel_fwd_canon")
    return out


def kernel_bwd_canon(ctx, G, pack1, n1, pack2, n2, d, p, dir_idx, idx_base, hyp, d_x1, d_v1, d_hyp, workspace=None):
    P1, s1, vn1 = pack1
    isd = G.dtype == f64
    _req(G, f64 if isd else f32, "G", 2)
    nbytes = int(lib.dsvgp_kernel_bwd_workspace_bytes(n1, n2, d, p))
    if workspace is None or workspace.numel() < nbytes:
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=G.device)
    check(lib.dsvgp_kernel_bwd_canon(ctx.h, _ptr(G), _ld(G), 1 if isd else 0, _ptr(P1), _ptr(s1), _ptr(vn1), n1, _ptr(pack2[0]),
                                     _ptr(pack2[1]), n2, d, p, _ptr(dir_idx), int(idx_base), _ptr(hyp), _ptr(d_x1), _ptr(d_v1), _ptr(d_hyp),
                                     _ptr(workspace)), "dsvgp_kernel_bwd_canon")
    return workspace


# ---- fp64 model mode (csrc/assemble64.hip) ---------------------------------------------------------------------
def pack_points_f64(ctx, x, v, p, hyp, center=None):
    """-> (P[n(p+1), DP], self[n(p+1)], vnorm[n p]) in double precision"""
    _req(x, f64, "x", 2)
    n, d = x.shape
    if p > 0:
        _req(v, f64, "v", 2)
        if v.shape != (n * p, d):
            raise ValueError("directions must be [n*p, d] = [%d, %d], got %s" % (n * p, d, tuple(v.shape)))
    if not x.is_contiguous() or (p > 0 and not v.is_contiguous()):
        raise ValueError("x and v must be contiguous")
    DP = packed_width(d)
    P = torch.empty(n * (p + 1), DP, dtype=f64, device=x.device)
    sf = torch.empty(n * (p + 1), dtype=f64, device=x.device)
    vn = torch.empty(max(n * p, 1), dtype=f64, device=x.device)
    check(lib.dsvgp_pack_points_f64(ctx.h, _ptr(x), _ptr(v if p > 0 else None), n, d, p, _ptr(_req(hyp, f64, "hyp", 1)),
                                    _ptr(center), _ptr(P), _ptr(sf), _ptr(vn)), "dsvgp_pack_points_f64")
    return P, sf, vn


def kernel_fwd_f64(ctx, pack1, n1, pack2, n2, d, p, hyp, jitter=0.0, out=None):
    """outputscale * K(x1, x2; v1, v2) [+ jitter I] in fp64: T = P1 P2^T on the fp64 MFMA GEMM, micro-block transform in place"""
    q = p + 1
    if out is None:
        out = torch.empty(n1 * q, n2 * q, dtype=f64, device=pack1[0].device)
    _req(out, f64, "out", 2)
    K4 = (d + 3) // 4 * 4
    gemm(ctx, _lib.TRANS_B, pack1[0], pack2[0], out, M=n1 * q, N=n2 * q, K=K4)
    check(lib.dsvgp_kernel_transform_f64(ctx.h, _ptr(out), _ld(out), _ptr(pack1[1]), n1, _ptr(pack2[1]), n2, p, _ptr(hyp),
                                         float(jitter)), "dsvgp_kernel_transform_f64")
    return out


def kernel_bwd_f64(ctx, G, pack1, n1, pack2, n2, d, p, hyp, symmetric, d_x1, d_v1, d_hyp, scratch=None):
    """backward of kernel_fwd_f64 w.r.t. (x1, v1, lengthscale, outputscale); accumulates into d_x1, d_v1, d_hyp[0..1]"""
    q = p + 1
    _req(G, f64, "G", 2)
    K4 = (d + 3) // 4 * 4
    DP = pack1[0].shape[1]
    T = scratch if scratch is not None else torch.empty(n1 * q, n2 * q, dtype=f64, device=G.device)
    gemm(ctx, _lib.TRANS_B, pack1[0], pack2[0], T, M=n1 * q, N=n2 * q, K=K4)
    check(lib.dsvgp_kernel_bwd_transform_f64(ctx.h, _ptr(G), _ld(G), _ptr(T), _ld(T), _ptr(pack1[1]), n1, _ptr(pack2[1]), n2, p,
                                             _ptr(hyp), _ptr(_req(d_hyp, f64, "d_hyp", 1))), "dsvgp_kernel_bwd_transform_f64")
    dP = torch.empty(n1 * q, DP, dtype=f64, device=G.device)
    gemm(ctx, 0, T, pack2[0], dP)                                   # Tbar [P2 | indicator]
    check(lib.dsvgp_kernel_bwd_points_f64(ctx.h, _ptr(dP), _ptr(pack1[0]), _ptr(pack1[2]), n1, d, p, _ptr(hyp),
                                          1 if symmetric else 0, _ptr(_req(d_x1, f64, "d_x1", 2)),
                                          _ptr(d_v1 if p > 0 else None)), "dsvgp_kernel_bwd_points_f64")


def colstats_f64(ctx, A, W, m):
    """(mu0 = A^T m, cs = colsum(W^2 - A^2)) in fp64; W None -> cs None"""
    _req(A, f64, "A", 2); _req(m, f64, "m", 1)
    Mp, Bp = A.shape
    mu = torch.empty(Bp, dtype=f64, device=A.device)
    cs = torch.empty(Bp, dtype=f64, device=A.device) if W is not None else None
    check(lib.dsvgp_colstats_f64(ctx.h, _ptr(A), _ld(A), _ptr(W), _ld(W) if W is not None else 0, _ptr(m), Mp, Bp, _ptr(mu),
                                 _ptr(cs)), "dsvgp_colstats_f64")
    return mu, cs


def abar_f64(ctx, A, U, m, mu_bar, var_bar, Abar, Av=None):
    _req(A, f64, "A", 2); _req(Abar, f64, "Abar", 2)
    Mp, Bp = A.shape
    check(lib.dsvgp_abar_f64(ctx.h, _ptr(A), _ld(A), _ptr(U), _ld(U) if U is not None else 0, _ptr(_req(m, f64, "m", 1)),
                             _ptr(_req(mu_bar, f64, "mu_bar", 1)), _ptr(_req(var_bar, f64, "var_bar", 1)), Mp, Bp, _ptr(Abar),
                             _ld(Abar), _ptr(Av), _ld(Av) if Av is not None else 0), "dsvgp_abar_f64")


def likelihood_terms_f64(ctx, mu0, cs, y, constant, p, hyp, mll_type, rows):
    """(mu, varn, mu_bar, var_bar, scal[8]) of the fp64 model's likelihood + objective in one launch (see dsvgp.h)"""
    n = mu0.shape[0]
    dev = mu0.device
    mu, varn, mu_bar, var_bar = (torch.empty(n, dtype=f64, device=dev) for _ in range(4))
    scal = torch.empty(8, dtype=f64, device=dev)
    check(lib.dsvgp_likelihood_terms_f64(ctx.h, _ptr(_req(mu0, f64, "mu0", 1)), _ptr(_req(cs, f64, "cs", 1)),
                                         _ptr(_req(y, f64, "y", 1)), _ptr(_req(constant, f64, "constant", 1)), n, int(p), _ptr(hyp),
                                         int(mll_type), float(rows), _ptr(mu), _ptr(varn), _ptr(mu_bar), _ptr(var_bar), _ptr(scal)),
          "dsvgp_likelihood_terms_f64")
    return mu, varn, mu_bar, var_bar, scal


def elbo_fast_tail_f64(ctx, mu0, y, constant, npts, pd, hyp, tvar, rows):
    """(mu, mu_bar, scal[8]) of the fp64 ELBO fast path's scalar tail in two launches (see dsvgp.h): scal = {sum ll, d/d noise,
    d/d constant, d/d outputscale, d/d lengthscale, vbar, sum r^2, sum r}"""
    n = mu0.shape[0]
    dev = mu0.device
    mu, mu_bar = torch.empty(n, dtype=f64, device=dev), torch.empty(n, dtype=f64, device=dev)
    scal = torch.empty(8, dtype=f64, device=dev)
    check(lib.dsvgp_elbo_fast_tail_f64(ctx.h, _ptr(_req(mu0, f64, "mu0", 1)), _ptr(_req(y, f64, "y", 1)),
                                       _ptr(_req(constant, f64, "constant", 1)), n, int(npts), int(pd), _ptr(_req(hyp, f64, "hyp", 1)),
                                       _ptr(_req(tvar.reshape(1), f64, "tvar", 1)), float(rows), _ptr(mu), _ptr(mu_bar), _ptr(scal)),
          "dsvgp_elbo_fast_tail_f64")
    return mu, mu_bar, scal


def kernel_diag(ctx, n, p, hyp):
    out = torch.empty(n * (p + 1), dtype=f32, device=hyp.device)
    check(lib.dsvgp_kernel_diag(ctx.h, n, p, _ptr(hyp), _ptr(out)), "dsvgp_kernel_diag")
    return out


def kernel_bwd(ctx, G, pack1, n1, pack2, n2, d, p, hyp, symmetric, d_x1, d_v1, d_hyp, workspace=None):
    P1, s1, vn1 = pack1
    P2, s2 = pack2[0], pack2[1]
    isd = G.dtype == f64
    _req(G, f64 if isd else f32, "G", 2)
    nbytes = int(lib.dsvgp_kernel_bwd_workspace_bytes(n1, n2, d, p))
    if workspace is None or workspace.numel() < nbytes:
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=G.device)
    check(lib.dsvgp_kernel_bwd(ctx.h, _ptr(G), _ld(G), 1 if isd else 0, _ptr(P1), _ptr(s1), _ptr(vn1), n1, _ptr(P2),
                               _ptr(s2), n2, d, p, _ptr(hyp), 1 if symmetric else 0, _ptr(d_x1),
                               _ptr(d_v1 if p > 0 else None), _ptr(d_hyp), _ptr(workspace)), "dsvgp_kernel_bwd")
    return workspace


def potrf_workspace(n, device):
    """Scratch of the blocked MFMA Cholesky (inverted 64 x 64 diagonal blocks, W_k tiles).  Owned by the caller, ONE per
    factor: ``trtri_blocks`` later seeds its recursion from the blocks the factorisation left in it."""
    return torch.empty(int(lib.dsvgp_potrf_workspace_bytes(int(n), 1)), dtype=torch.uint8, device=device)


def _potrf_scratch(A, ws):
    need = int(lib.dsvgp_potrf_workspace_bytes(A.shape[0], 1))
    if ws is None:
        return potrf_workspace(A.shape[0], A.device)
    if not ws.is_cuda or ws.numel() * ws.element_size() < need:
        raise ValueError("potrf workspace too small: %d < %d" % (ws.numel() * ws.element_size(), need))
    return ws


def potrf_(ctx, A, info, algo=1, ws=None):
    """In-place lower Cholesky.  algo 0 = rocSOLVER dpotrf, 1 = blocked MFMA Cholesky (default).
    ``ws``: ``potrf_workspace(n)`` owned by the caller for THIS factor (a fresh one is allocated when omitted);
    returned so that ``trtri_blocks(..., potrf_ws=)`` can reuse the inverted diagonal blocks."""
    _req(A, f64, "A", 2)
    n = A.shape[0]
    if algo == 1:
        ws = _potrf_scratch(A, ws)
    else:
        ws = None
    check(lib.dsvgp_potrf(ctx.h, _ptr(A), n, _ld(A), _ptr(info), int(algo), _ptr(ws)), "dsvgp_potrf")
    return ws


def potrf_inverse_(ctx, A, info, nb, workspace, ws=None):
    """In-place lower Cholesky (blocked MFMA algorithm) AND the explicit inverse into the trsm workspace, in the same
    launches.  Only for nb >= n; later ``trsm(..., reuse_inverse=True)`` calls use the inverse.  ``ws`` as in ``potrf_``."""
    _req(A, f64, "A", 2)
    n = A.shape[0]
    ws = _potrf_scratch(A, ws)
    need = int(lib.dsvgp_trsm_workspace_bytes(n, n, int(nb)))       # the inverse and its transposed copy live there
    if workspace.numel() < need:
        raise ValueError("potrf_inverse_: trsm workspace too small: %d < %d" % (workspace.numel(), need))
    check(lib.dsvgp_potrf_inverse(ctx.h, _ptr(A), n, _ld(A), _ptr(info), _ptr(ws), int(nb), _ptr(workspace)),
          "dsvgp_potrf_inverse")
    return ws


def add_diag_(ctx, A, delta):
    check(lib.dsvgp_add_diag(ctx.h, _ptr(A), A.shape[0], _ld(A), float(delta)), "dsvgp_add_diag")


def trsm_workspace(n, nrhs, nb, device):
    return torch.empty(int(lib.dsvgp_trsm_workspace_bytes(n, nrhs, nb)), dtype=torch.uint8, device=device)


def trsm(ctx, L, B, trans, X64, X32, nb, workspace, reuse_inverse=False):
    _req(L, f64, "L", 2)
    isd = B.dtype == f64
    _req(B, f64 if isd else f32, "B", 2)
    if X64 is not None:
        _req(X64, f64, "X64", 2)
    n, nrhs = B.shape
    if L.shape != (n, n) or (X64 is not None and X64.shape != (n, nrhs)) or (X32 is not None and X32.shape != (n, nrhs)):
        raise ValueError("trsm shape mismatch")
    need = int(lib.dsvgp_trsm_workspace_bytes(n, nrhs, nb))
    if workspace.numel() < need:
        raise ValueError("trsm workspace too small: %d < %d" % (workspace.numel(), need))
    check(lib.dsvgp_trsm(ctx.h, _ptr(L), _ld(L), n, 1 if trans else 0, _ptr(B), _ld(B), 1 if isd else 0, nrhs,
                         _ptr(X64), _ld(X64) if X64 is not None else 0, _ptr(X32), _ld(X32) if X32 is not None else 0, nb,
                         _ptr(workspace),
                         1 if reuse_inverse else 0), "dsvgp_trsm")


def trtri_blocks(ctx, L, nrhs_max, nb, workspace, potrf_ws=None):
    """Invert the nb x nb diagonal blocks of L into the trsm workspace (the first phase of dsvgp_trsm).
    ``potrf_ws``: what ``potrf_(algo=1)`` returned for THIS L (its inverted 64 x 64 diagonal blocks are reused)."""
    _req(L, f64, "L", 2)
    n = L.shape[0]
    need = int(lib.dsvgp_trsm_workspace_bytes(n, nrhs_max, nb))
    if workspace.numel() < need:
        raise ValueError("trsm workspace too small: %d < %d" % (workspace.numel(), need))
    check(lib.dsvgp_trtri(ctx.h, _ptr(L), _ld(L), n, nb, _ptr(potrf_ws), _ptr(workspace)), "dsvgp_trtri")


def gemm(ctx, flags, A, B, C_out, alpha=1.0, beta=0.0, Cin=None, C32=None, kscale=None, M=None, N=None, K=None):
    """C_out = alpha*op(A)op(B) + beta*Cin on the MFMA GEMM; compute dtype = C_out.dtype."""
    isd = C_out.dtype == f64
    if M is None:
        M = A.shape[1] if flags & _lib.TRANS_A else A.shape[0]
    if K is None:
        K = A.shape[0] if flags & _lib.TRANS_A else A.shape[1]
    if N is None:
        N = B.shape[0] if flags & _lib.TRANS_B else B.shape[1]
    kb = B.shape[1] if flags & _lib.TRANS_B else B.shape[0]
    if kb < K or C_out.shape[0] < M or C_out.shape[1] < N:
        raise ValueError("gemm shape mismatch")
    _req(A, f64 if isd else f32, "A", 2)
    if isd and B.dtype == f32:
        flags |= _lib.B_IS_FLOAT
    else:
        _req(B, f64 if isd else f32, "B", 2)
    if Cin is not None and isd and Cin.dtype == f32:
        flags |= _lib.CIN_IS_FLOAT
    check(lib.dsvgp_gemm(ctx.h, 1 if isd else 0, flags, M, N, K, float(alpha), _ptr(A), _ld(A), _ptr(B), _ld(B),
                         float(beta), _ptr(Cin), _ld(Cin) if Cin is not None else 0, _ptr(C_out), _ld(C_out),
                         _ptr(C32), _ld(C32) if C32 is not None else 0, _ptr(kscale)), "dsvgp_gemm")
    return C_out


def split3_bf16(ctx, src, transpose=False, out=None):
    """three bf16 planes of the float32 matrix ``src`` [R, C] (or of its transpose): uint8 buffer holding 3 x rows_out x kpad(K) bf16"""
    _req(src, f32, "src", 2)
    R, Cc = src.shape
    rows_out, K = (Cc, R) if transpose else (R, Cc)
    nbytes = int(lib.dsvgp_split3_bytes(rows_out, K))
    if out is None or out.numel() < nbytes:
        out = torch.empty(nbytes, dtype=torch.uint8, device=src.device)
    check(lib.dsvgp_split3_bf16(ctx.h, _ptr(src), _ld(src), R, Cc, 1 if transpose else 0, _ptr(out)), "dsvgp_split3_bf16")
    return out


def gemm3b(ctx, flags, M, N, K, Aplanes, a_rows, Bplanes, b_rows, C, alpha=1.0):
    """C[M, N] = alpha A B^T on the bf16 matrix pipe from plane triples (csrc/gemm3b.hip)"""
    _req(C, f32, "C", 2)
    check(lib.dsvgp_gemm3b(ctx.h, int(flags), int(M), int(N), int(K), float(alpha), _ptr(Aplanes), int(a_rows), _ptr(Bplanes), int(b_rows),
                           _ptr(C), _ld(C)), "dsvgp_gemm3b")
    return C


def widen_f32_f64(ctx, src, dst):
    """dst (float64) = src (float32), both 2-D with unit inner stride"""
    M, N = src.shape
    check(lib.dsvgp_widen_f32_f64(ctx.h, _ptr(_req(src, f32, "src", 2)), _ld(src), _ptr(_req(dst, f64, "dst", 2)), _ld(dst), M, N),
          "dsvgp_widen_f32_f64")
    return dst


def predictive_stats(ctx, A, W, p, m, constant, hyp, mu, var, workspace=None):
    Mp, nc = A.shape
    nbytes = int(lib.dsvgp_stats_workspace_bytes(Mp, nc))
    if workspace is None or workspace.numel() < nbytes:
        workspace = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=A.device)
    check(lib.dsvgp_predictive_stats(ctx.h, _ptr(_req(A, f32, "A", 2)), _ld(A), _ptr(_req(W, f32, "W", 2)), _ld(W), Mp,
                                     nc, p, _ptr(m), _ptr(constant), _ptr(hyp), _ptr(mu), _ptr(var), _ptr(workspace)),
          "dsvgp_predictive_stats")
    return workspace


def likelihood_terms(ctx, mu, var, y, p, hyp, mll_type, global_rows, mu_bar, var_bar, varn, scalars):
    check(lib.dsvgp_likelihood_terms(ctx.h, _ptr(mu), _ptr(var), _ptr(_req(y, f32, "y", 1)), mu.shape[0], p, _ptr(hyp),
                                     int(mll_type), float(global_rows), _ptr(mu_bar), _ptr(var_bar), _ptr(varn),
                                     _ptr(scalars)), "dsvgp_likelihood_terms")


def abar(ctx, A, U, m, mu_bar, var_bar, out):
    Mp, nc = A.shape
    check(lib.dsvgp_abar(ctx.h, _ptr(A), _ld(A), _ptr(U), _ld(U), Mp, nc, _ptr(m), _ptr(mu_bar), _ptr(var_bar),
                         _ptr(out), _ld(out)), "dsvgp_abar")


def rowdot_accum(ctx, A, vec, out):
    Mp, nc = A.shape
    check(lib.dsvgp_rowdot(ctx.h, _ptr(A), _ld(A), Mp, nc, _ptr(vec), _ptr(out)), "dsvgp_rowdot")


def kl_terms(ctx, m, LS, num_data, kl_buf, d_m, d_LS):
    Mp = m.shape[0]
    check(lib.dsvgp_kl_terms(ctx.h, _ptr(m), _ptr(_req(LS, f32, "L_S", 2)), _ld(LS), Mp, float(num_data), _ptr(kl_buf),
                             _ptr(d_m), _ptr(d_LS), _ld(d_LS)), "dsvgp_kl_terms")


def kl_terms_scaled(ctx, m, LS, num_data, add_kl, hyp, global_rows, kl_buf, d_m, d_LS):
    """d_LS(lower) <- d_LS / (noise rows) [+ KL gradient], d_m [+= m / num_data], kl_buf[0] = KL or 0 -- noise read on the device"""
    Mp = m.shape[0]
    check(lib.dsvgp_kl_terms_scaled(ctx.h, _ptr(m), _ptr(_req(LS, f32, "L_S", 2)), _ld(LS), Mp, float(num_data),
                                    1 if add_kl else 0, _ptr(hyp), float(global_rows), _ptr(kl_buf), _ptr(d_m), _ptr(d_LS),
                                    _ld(d_LS)), "dsvgp_kl_terms_scaled")


def scale_by_vbar_(ctx, tensors, hyp, global_rows):
    """x *= 1 / (noise * global_rows) for up to three contiguous float32 tensors, noise read on the device"""
    ts = [t for t in tensors if t is not None and t.numel()]
    if len(ts) > 3:
        raise ValueError("at most three tensors")
    for t in ts:
        if t.dtype != f32 or not t.is_cuda or not t.is_contiguous():
            raise ValueError("scale_by_vbar: contiguous float32 GPU tensors only")
    args = []
    for k in range(3):
        if k < len(ts):
            args += [_ptr(ts[k]), ts[k].numel()]
        else:
            args += [C.c_void_p(0), 0]
    check(lib.dsvgp_scale_by_vbar(ctx.h, *args, _ptr(hyp), float(global_rows)), "dsvgp_scale_by_vbar")


def adam_step_multi_dev_(ctx, params, grads, exp_avgs, exp_avg_sqs, lr_step_dev, beta1, beta2, eps, guard=None):
    """``adam_step_multi_`` with (lr, step) read from the device float[2] ``lr_step_dev``; skipped while ``guard[0] != 0``"""
    import ctypes
    n = len(params)
    if n > ADAM_MAX_TENSORS:
        raise ValueError("at most %d tensors per launch" % ADAM_MAX_TENSORS)
    for group in (params, grads, exp_avgs, exp_avg_sqs):
        for t in group:
            if t.dtype != f32 or not t.is_cuda or not t.is_contiguous():
                raise ValueError("adam: tensors must be contiguous float32 GPU tensors")
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    sizes = (ctypes.c_int64 * n)(*[t.numel() for t in params])
    check(lib.dsvgp_adam_step_multi_dev(ctx.h, n, arr(params), arr(grads), arr(exp_avgs), arr(exp_avg_sqs), sizes,
                                        _ptr(_req(lr_step_dev, f32, "lr_step_dev", 1)), float(beta1), float(beta2), float(eps),
                                        _ptr(guard)), "dsvgp_adam_step_multi_dev")


def phi_symmetrize_(ctx, G):
    check(lib.dsvgp_phi_symmetrize(ctx.h, _ptr(G), G.shape[0], _ld(G)), "dsvgp_phi_symmetrize")


def gemv_f64(ctx, A, x, y, trans=False):
    """y = A x (trans False: A [M, N], x [N], y [M]) or y = A^T x (x [M], y [N]) in fp64"""
    _req(A, f64, "A", 2)
    M, N = A.shape
    _req(x, f64, "x", 1)
    _req(y, f64, "y", 1)
    if x.numel() != (M if trans else N) or y.numel() != (N if trans else M):
        raise ValueError("gemv_f64 shape mismatch")
    check(lib.dsvgp_gemv_f64(ctx.h, 1 if trans else 0, _ptr(A), _ld(A), M, N, _ptr(x), _ptr(y)), "dsvgp_gemv_f64")
    return y


def transpose_f64(ctx, src, dst):
    check(lib.dsvgp_transpose_f64(ctx.h, _ptr(src), _ld(src), src.shape[0], src.shape[1], _ptr(dst), _ld(dst)),
          "dsvgp_transpose_f64")


def residual_terms(ctx, mu, y, hyp, global_rows, mu_bar, sums):
    check(lib.dsvgp_residual_terms(ctx.h, _ptr(mu), _ptr(_req(y, f32, "y", 1)), mu.shape[0], _ptr(hyp),
                                   float(global_rows), _ptr(mu_bar), _ptr(sums)), "dsvgp_residual_terms")


def variational_terms(ctx, m, LS, num_data, scale, add_kl, hyp, global_rows, G, t1_scale, kl_buf, sums, d_m, d_LS):
    """trace terms + [scaling by 1 / (noise rows)] + [KL value and gradient] in one pass over (L_S, d_LS = tril(G L_S)); see dsvgp.h"""
    Mp = m.shape[0]
    if kl_buf.numel() < 1 + 2 * Mp:
        raise ValueError("kl_buf needs 1 + 2 M' floats")
    check(lib.dsvgp_variational_terms(ctx.h, _ptr(m), _ptr(_req(LS, f32, "L_S", 2)), _ld(LS), Mp, float(num_data),
                                      (1 if scale else 0) | (2 if add_kl else 0), _ptr(hyp), float(global_rows), _ptr(G), _ld(G),
                                      float(t1_scale), _ptr(kl_buf), _ptr(sums), _ptr(d_m), _ptr(d_LS), _ld(d_LS)),
          "dsvgp_variational_terms")


def trace_terms(ctx, LS, T1, G, n, sums, t1_scale=1.0):
    check(lib.dsvgp_trace_terms(ctx.h, _ptr(LS), _ld(LS), _ptr(T1), _ld(T1), _ptr(G), _ld(G), n, float(t1_scale),
                                _ptr(sums)), "dsvgp_trace_terms")


def elbo_fast_finalize(ctx, sums, hyp, npts, p, global_rows, scalars):
    check(lib.dsvgp_elbo_fast_finalize(ctx.h, _ptr(sums), _ptr(hyp), npts, p, float(global_rows), _ptr(scalars)),
          "dsvgp_elbo_fast_finalize")


def mirror_lower_f32_(ctx, G, n):
    check(lib.dsvgp_mirror_lower_f32(ctx.h, _ptr(_req(G, f32, "G", 2)), n, _ld(G)), "dsvgp_mirror_lower_f32")


def sminus_i_col_(ctx, A, n, m, hyp, rows):
    """A[n, n+1] -> [A[:, :n] - I | m * noise * rows] in place (one launch)"""
    if A.shape != (n, n + 1) or A.dtype != f32 or A.stride(1) != 1:
        raise ValueError("sminus_i_col_ wants a float32 [n, n + 1] view with unit column stride")
    check(lib.dsvgp_sminus_i_col(ctx.h, _ptr(A), n, int(A.stride(0)), _ptr(_req(m, f32, "m", 1)), _ptr(hyp), float(rows)),
          "dsvgp_sminus_i_col")


def add_diag_f32_(ctx, A, n, delta):
    check(lib.dsvgp_add_diag_f32(ctx.h, _ptr(_req(A, f32, "A", 2)), n, _ld(A), float(delta)), "dsvgp_add_diag_f32")


def transpose_f32(ctx, src, dst):
    check(lib.dsvgp_transpose_f32(ctx.h, _ptr(src), _ld(src), src.shape[0], src.shape[1], _ptr(dst), _ld(dst)),
          "dsvgp_transpose_f32")


def tril_packed_numel(n, nextra=0):
    return n * (n + 1) // 2 + nextra


def tril_pack_f32(ctx, src, extra, dst):
    """dst = [packed lower triangle of src[n, n] | extra]: the half-volume data-parallel all-reduce operand."""
    n = src.shape[0]
    _req(src, f32, "src", 2); _req(extra, f32, "extra", 1); _req(dst, f32, "dst", 1)
    if src.shape[1] < n or dst.numel() < tril_packed_numel(n, extra.numel()):
        raise ValueError("tril_pack shape mismatch")
    check(lib.dsvgp_tril_pack_f32(ctx.h, _ptr(src), _ld(src), n, _ptr(extra), extra.numel(), _ptr(dst)),
          "dsvgp_tril_pack_f32")


def tril_unpack_f32(ctx, src, dst, extra):
    n = dst.shape[0]
    _req(src, f32, "src", 1); _req(extra, f32, "extra", 1); _req(dst, f32, "dst", 2)
    if dst.shape[1] < n or src.numel() < tril_packed_numel(n, extra.numel()):
        raise ValueError("tril_unpack shape mismatch")
    check(lib.dsvgp_tril_unpack_f32(ctx.h, _ptr(src), n, _ptr(dst), _ld(dst), _ptr(extra), extra.numel()),
          "dsvgp_tril_unpack_f32")


def gather_batch(ctx, X, Y, idx, cols, p, xb, yb, E=None, Db=None):
    """minibatch rows of X, the selected columns of Y and (with the [d, d] direction table E) the batch's derivative
    directions Db[nb * p, d] in one launch"""
    if E is not None and (E.shape != (X.shape[1], X.shape[1]) or not E.is_contiguous() or E.dtype != f32 or Db is None or
                          Db.shape != (idx.shape[0] * p, X.shape[1]) or not Db.is_contiguous() or Db.dtype != f32):
        raise ValueError("gather_batch: E must be [d, d] and Db [nb * p, d] float32 contiguous")
    check(lib.dsvgp_gather_batch(ctx.h, _ptr(_req(X, f32, "X", 2)), _ptr(_req(Y, f32, "Y", 2)),
                                 _ptr(_req(idx, torch.int64, "idx", 1)), idx.shape[0], X.shape[1], Y.shape[1],
                                 _ptr(_req(cols, torch.int32, "cols", 1)), p, _ptr(xb), _ptr(yb), _ptr(E), _ptr(Db)),
          "dsvgp_gather_batch")


ADAM_MAX_TENSORS = 16


def adam_step_multi_(ctx, params, grads, exp_avgs, exp_avg_sqs, lr, beta1, beta2, eps, step, tril=None, guard=None):
    """torch.optim.Adam update of several tensors that share (lr, betas, eps, step) in one launch.  ``tril[k]`` true: params[k] is a square
    matrix of which only the lower triangle is a parameter (its gradient and moments are zero above the diagonal): the strict upper
    triangle is left alone (float32 only)."""
    import ctypes
    n = len(params)
    if n > ADAM_MAX_TENSORS:
        raise ValueError("at most %d tensors per launch" % ADAM_MAX_TENSORS)
    dt = params[0].dtype if n else f32
    for group in (params, grads, exp_avgs, exp_avg_sqs):
        for t in group:
            if t.dtype != dt or dt not in (f32, f64) or not t.is_cuda or not t.is_contiguous():
                raise ValueError("adam: tensors must be contiguous GPU tensors of one dtype (float32 or float64)")
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    sz = [t.numel() for t in params]
    if tril is not None and dt == f32:
        sz = [-t.shape[0] if (tr and t.dim() == 2 and t.shape[0] == t.shape[1] and t.shape[0] > 1) else k for t, k, tr in zip(params, sz, tril)]
    sizes = (ctypes.c_int64 * n)(*sz)
    if guard is not None:       # (``guard``: int32 device tensor, the update is skipped on the device while guard[0] != 0; float32 only)
        if dt != f32 or guard.dtype != torch.int32 or not guard.is_cuda:
            raise ValueError("adam: a guarded update takes float32 tensors and an int32 GPU guard word")
        check(lib.dsvgp_adam_step_multi_guarded(ctx.h, n, arr(params), arr(grads), arr(exp_avgs), arr(exp_avg_sqs), sizes, float(lr), float(beta1),
                                                float(beta2), float(eps), int(step), _ptr(guard)), "dsvgp_adam_step_multi_guarded")
        return
    fn, name = (lib.dsvgp_adam_step_multi_f64, "dsvgp_adam_step_multi_f64") if dt == f64 else (lib.dsvgp_adam_step_multi, "dsvgp_adam_step_multi")
    check(fn(ctx.h, n, arr(params), arr(grads), arr(exp_avgs), arr(exp_avg_sqs), sizes, float(lr), float(beta1), float(beta2),
             float(eps), int(step)), name)


def adam_step_(ctx, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
    for t, nm in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        if not t.is_contiguous() or t.dtype != f32 or not t.is_cuda:
            raise ValueError("adam: %s must be a contiguous float32 GPU tensor" % nm)
    check(lib.dsvgp_adam_step(ctx.h, _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(),
                              float(lr), float(beta1), float(beta2), float(eps), int(step)), "dsvgp_adam_step")


def gemm_lib_f32(ctx, flags, A, B, C_out, alpha=1.0, beta=0.0):
    """C_out = alpha op(A) op(B) + beta C_out, plain dense fp32, through rocBLAS (no structure flags)."""
    _req(A, f32, "A", 2); _req(B, f32, "B", 2); _req(C_out, f32, "C", 2)
    M = A.shape[1] if flags & _lib.TRANS_A else A.shape[0]
    K = A.shape[0] if flags & _lib.TRANS_A else A.shape[1]
    N = B.shape[0] if flags & _lib.TRANS_B else B.shape[1]
    kb = B.shape[1] if flags & _lib.TRANS_B else B.shape[0]
    if kb != K or C_out.shape != (M, N):
        raise ValueError("gemm_lib_f32 shape mismatch")
    check(lib.dsvgp_gemm_lib_f32(ctx.h, int(flags), M, N, K, float(alpha), _ptr(A), _ld(A), _ptr(B), _ld(B), float(beta),
                                 _ptr(C_out), _ld(C_out)), "dsvgp_gemm_lib_f32")


# ---- contour-integral-quadrature whitening (csrc/ciq.hip) ------------------------------------------------------
# Every wrapper serves both scalar types: float32 tensors go to dsvgp_ciq_*, float64 tensors (the fp64 model mode) to
# dsvgp_ciq_*_f64; all tensors of one call must share the type of its first matrix argument.
def _ciq_fn(name, dt):
    if dt == f32:
        return getattr(lib, name)
    if dt == f64:
        return getattr(lib, (name[:-4] if name.endswith("_f32") else name) + "_f64")
    raise TypeError("CIQ kernels are built for float32 and float64 tensors, got %s" % dt)


def ciq_workspace_bytes(Q, t, n, cap, dtype=f32):
    return int(_ciq_fn("dsvgp_ciq_workspace_bytes", dtype)(int(Q), int(t), int(n), int(cap)))


def ciq_lanczos(ctx, K, v0, iters):
    """alpha[iters], beta[iters] of `iters` Lanczos steps with the symmetric K started at v0."""
    dt = K.dtype
    _req(K, dt, "K", 2)
    n = K.shape[0]
    alpha = torch.zeros(iters, dtype=dt, device=K.device)
    beta = torch.zeros(iters, dtype=dt, device=K.device)
    ws = torch.empty(3 * n + 8, dtype=dt, device=K.device)
    check(_ciq_fn("dsvgp_ciq_lanczos", dt)(ctx.h, _ptr(K), _ld(K), _ptr(_req(v0, dt, "v0", 1)), n, int(iters), _ptr(alpha),
                                          _ptr(beta), _ptr(ws)), "dsvgp_ciq_lanczos")
    return alpha, beta


def ciq_qp(Q):
    """Shift / output counts of the CIQ coefficient tables are padded to multiples of 4 (csrc/ciq.hip)."""
    return 4 * ((int(Q) + 3) // 4)


def ciq_solve(ctx, K, R, sigma, omega, basis, ycoef, rnorm, out, workspace, tol=1e-4, max_iter=1000, check_every=10):
    """out[t,n] = sum_q omega_q (K + sigma_q)^-1 R rows by basis-resident msMINRES: basis[cap+1,t,n] receives the Lanczos
    rows, ycoef[t,cap,QP] the per-shift coefficients (x_q = rnorm * mix(basis, ycoef)), rnorm[t] the row norms of R.
    Returns the iteration count, or None when `cap` iterations were not enough (call again with a larger basis)."""
    dt = K.dtype
    _req(K, dt, "K", 2)
    _req(R, dt, "R", 2)
    t, n = R.shape
    Q = sigma.shape[0]
    cap = basis.shape[0] - 1
    if (K.shape != (n, n) or basis.shape != (cap + 1, t, n) or cap < 1 or ycoef.shape != (t, cap, ciq_qp(Q)) or
            rnorm.shape != (t,) or out.shape != (t, n) or not basis.is_contiguous() or not ycoef.is_contiguous()):
        raise ValueError("ciq_solve shape mismatch")
    for a, name in ((basis, "basis"), (ycoef, "ycoef"), (rnorm, "rnorm"), (out, "out")):
        if a.dtype != dt or not a.is_cuda:
            raise TypeError("%s must be a %s GPU tensor" % (name, dt))
    need = ciq_workspace_bytes(Q, t, n, cap, dt)
    if workspace.numel() * workspace.element_size() < need:
        raise ValueError("ciq workspace too small")
    its = C.c_int(0)
    rc = _ciq_fn("dsvgp_ciq_solve", dt)(ctx.h, _ptr(K), _ld(K), _ptr(R), _ld(R), t, n, _ptr(_req(sigma, dt, "sigma", 1)),
                                        _ptr(_req(omega, dt, "omega", 1)), Q, float(tol), int(max_iter), int(check_every),
                                        _ptr(basis), cap, _ptr(ycoef), _ptr(rnorm), _ptr(out), _ld(out), _ptr(workspace),
                                        C.byref(its))
    if rc == _lib.ENOSPACE:
        return None
    check(rc, "dsvgp_ciq_solve")
    return its.value


def ciq_mix(ctx, basis, J, coef, Kout, rowscale, out):
    """out[k,row,:] = rowscale[row] * sum_{j<J} coef[row,j,k] basis[j,row,:] for k < Kout (coef[t,ldj,KP], KP % 4 == 0)."""
    _, t, n = basis.shape
    ldj, KP = coef.shape[1], coef.shape[2]
    dt = basis.dtype
    if (coef.shape[0] != t or J > ldj or J > basis.shape[0] or out.shape != (Kout, t, n) or not out.is_contiguous() or
            not coef.is_contiguous() or not basis.is_contiguous() or coef.dtype != dt or out.dtype != dt or
            (rowscale is not None and rowscale.dtype != dt)):
        raise ValueError("ciq_mix shape mismatch")
    check(_ciq_fn("dsvgp_ciq_mix", dt)(ctx.h, _ptr(basis), int(J), t, n, _ptr(coef), ldj, KP, int(Kout), _ptr(rowscale), _ptr(out),
                                      n), "dsvgp_ciq_mix")
    return out


def ciq_cross(ctx, ya, Ja, yb, Jb, omega, rn_a, rn_b):
    """C[t,Jb,KPa] = rn_a rn_b sum_q omega_q ya[:,ia,q] yb[:,jb,q]: coefficients that turn sum_q omega_q A_q^T B_q into
    stack(basisA[:Ja])^T stack(mix(basisB, C)) (A_q = rn_a mix(basisA, ya)_q, B_q likewise)."""
    t = ya.shape[0]
    Q = omega.shape[0]
    dt = ya.dtype
    if yb.shape[0] != t or ya.shape[2] != ciq_qp(Q) or yb.shape[2] != ciq_qp(Q) or Ja > ya.shape[1] or Jb > yb.shape[1]:
        raise ValueError("ciq_cross shape mismatch")
    if any(a.dtype != dt for a in (yb, omega, rn_a, rn_b)):
        raise TypeError("ciq_cross: every tensor must be %s" % dt)
    out = torch.empty(t, Jb, ciq_qp(Ja), dtype=dt, device=ya.device)
    check(_ciq_fn("dsvgp_ciq_cross", dt)(ctx.h, _ptr(ya), int(Ja), ya.shape[1], _ptr(yb), int(Jb), yb.shape[1], _ptr(omega), Q, t,
                                        _ptr(rn_a), _ptr(rn_b), _ptr(out)), "dsvgp_ciq_cross")
    return out


def ciq_rowstats(ctx, T, ST, p, m, constant, hyp, kxx_jitter=0.0):
    t, n = T.shape
    dev = T.device
    dt = T.dtype
    if any(a.dtype != dt for a in (ST, m, constant, hyp)):
        raise TypeError("ciq_rowstats: every tensor must be %s" % dt)
    imean, mu, var, live = (torch.empty(t, dtype=dt, device=dev) for _ in range(4))
    check(_ciq_fn("dsvgp_ciq_rowstats", dt)(ctx.h, _ptr(T), _ptr(ST), t, n, p, _ptr(m), _ptr(constant), _ptr(hyp), float(kxx_jitter),
                                           _ptr(imean), _ptr(mu), _ptr(var), _ptr(live)), "dsvgp_ciq_rowstats")
    return imean, mu, var, live


def ciq_tbar(ctx, T, ST, m, mu_bar, var_bar, live, imean, Tbar, VT):
    t, n = T.shape
    dt = T.dtype
    if any(a.dtype != dt for a in (ST, m, mu_bar, var_bar, live, imean, Tbar, VT)):
        raise TypeError("ciq_tbar: every tensor must be %s" % dt)
    cvec = torch.empty(t, dtype=dt, device=T.device)
    check(_ciq_fn("dsvgp_ciq_tbar", dt)(ctx.h, _ptr(T), _ptr(ST), t, n, _ptr(m), _ptr(mu_bar), _ptr(var_bar), _ptr(live),
                                       _ptr(imean), _ptr(Tbar), _ptr(VT), _ptr(cvec)), "dsvgp_ciq_tbar")
    return cvec


def sym_average_f32(ctx, A, out):
    check(lib.dsvgp_sym_average_f32(ctx.h, _ptr(_req(A, f32, "A", 2)), A.shape[0], _ld(A), _ptr(out), _ld(out)),
          "dsvgp_sym_average_f32")


def sym_average_f64(ctx, A, out):
    check(lib.dsvgp_sym_average_f64(ctx.h, _ptr(_req(A, f64, "A", 2)), A.shape[0], _ld(A), _ptr(_req(out, f64, "out", 2)),
                                    _ld(out)), "dsvgp_sym_average_f64")


def mfma_rate(ctx, is_double=True, millis=40):
    """TFLOP/s this card sustains on back-to-back MFMA instructions from registers (fp64 16x16x4 or fp32 32x32x2): the roof a
    GEMM kernel can reach under the card's power management.  Measurement aid for bench.py."""
    scratch = torch.empty(2 << 20, dtype=torch.uint8, device=ctx.device)
    out = C.c_double(0.0)
    check(lib.dsvgp_mfma_rate(ctx.h, 1 if is_double else 0, int(millis), _ptr(scratch), C.byref(out)), "dsvgp_mfma_rate")
    return out.value


def mfma_rate2(ctx, is_double=True, one_wave_per_simd=False, millis=40):
    """(sustained TFLOP/s, TFLOP/s of the first ~2 ms launch, in-kernel clock in GHz of the last launch) of the same probe with four
    waves per SIMD or one (the form the hardware guide's 155 TF fp32 figure was measured in)."""
    scratch = torch.empty(2 << 20, dtype=torch.uint8, device=ctx.device)
    out, burst, clk = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
    check(lib.dsvgp_mfma_rate2(ctx.h, (1 if is_double else 0) | (2 if one_wave_per_simd else 0), int(millis), _ptr(scratch),
                               C.byref(out), C.byref(burst), C.byref(clk)), "dsvgp_mfma_rate2")
    return out.value, burst.value, clk.value
