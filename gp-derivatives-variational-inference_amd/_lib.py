"""ctypes binding of libdsvgp_hip.so (the C ABI declared in include/dsvgp.h).

The product path has NO fallback: if the shared library is missing or fails to load, importing
this module raises.  Build it with ``python gp-derivatives-variational-inference_amd/build_ext.py``
(or ``__graft_entry__.build()``).
"""
import ctypes as C
import os

import torch  # noqa: F401  (loads torch's bundled HIP/rocBLAS/rocSOLVER first so one runtime is shared)

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSVGP_LIB_PATH lets tools/ time experimental builds of the SAME library (kernel ablations); it is not a fallback
LIB_PATH = os.environ.get("DSVGP_LIB_PATH") or os.path.join(_HERE, "libdsvgp_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libdsvgp_hip.so not found at %s -- the HIP extension is required (no CPU fallback). "
        "Run `python gp-derivatives-variational-inference_amd/build_ext.py`." % LIB_PATH)

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

_p = C.c_void_p
_i = C.c_int
_l = C.c_int64
_f = C.c_float
_d = C.c_double
_z = C.c_size_t

class ElboStepIO(C.Structure):
    """``dsvgp_elbo_step_io`` of include/dsvgp.h (device pointers of one ELBO step)"""
    _fields_ = [("Z", _p), ("V", _p), ("m", _p), ("LS", _p), ("ldls", _l),
                ("constant", _p), ("raw_lengthscale", _p), ("raw_outputscale", _p), ("raw_noise", _p),
                ("x", _p), ("y", _p), ("D", _p),
                ("flat", _p), ("flat_floats", _z),
                ("dZ", _p), ("dV", _p), ("dm", _p), ("dLS", _p), ("lddls", _l),
                ("d_hyp", _p), ("d_constant", _p), ("d_raw_lengthscale", _p), ("d_raw_outputscale", _p), ("d_raw_noise", _p),
                ("loss", _p), ("mu", _p), ("num_data", _d), ("global_rows", _d), ("kzz_jitter", _f),
                ("split_ws", _p), ("split_ws_bytes", _z), ("dir_idx", _p), ("dir_idx_base", _i), ("v_one_hot", _i)]


class ElboStepDP(C.Structure):
    """``dsvgp_elbo_step_dp`` of include/dsvgp.h (rank, world and the collective operands of one data-parallel rank)"""
    _fields_ = [("rank", _i), ("world", _i), ("wire", _p), ("wire_floats", _z), ("q_local", _p), ("q_all", _p),
                ("lbar_local", _p), ("lbar_all", _p)]


# name -> (restype, argtypes); mirrors include/dsvgp.h one to one
SIGNATURES = {
    "dsvgp_create": (_i, [C.POINTER(_p)]),
    "dsvgp_destroy": (_i, [_p]),
    "dsvgp_set_stream": (_i, [_p, _p]),
    "dsvgp_set_deterministic": (_i, [_p, _p, _z]),
    "dsvgp_elbo_step_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "dsvgp_elbo_step_plan_create": (_i, [_p, _i, _i, _i, _i, C.POINTER(_p)]),
    "dsvgp_elbo_step_plan_destroy": (_i, [_p]),
    "dsvgp_elbo_step_plan_bytes": (_z, [_p]),
    "dsvgp_elbo_step_f32": (_i, [_p, _p, _p, _p, _z, _i]),
    "dsvgp_elbo_step_po_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "dsvgp_elbo_step_po_plan_create": (_i, [_p, _i, _i, _i, _i, C.POINTER(_p)]),
    "dsvgp_elbo_step_po_f32": (_i, [_p, _p, _p, _p, _p, _z, _i]),
    "dsvgp_elbo_step_split_bytes": (_z, [_i, _i, _i, _i]),
    "dsvgp_elbo_step_dp_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "dsvgp_elbo_step_dp_plan_create": (_i, [_p, _i, _i, _i, _i, _i, C.POINTER(_p)]),
    "dsvgp_elbo_step_dp_f32": (_i, [_p, _p, _p, _p, _p, _z, _i, _i]),
    "dsvgp_elbo_step_status": (_i, [_p, _p, _p]),
    "dsvgp_elbo_step_timings": (_i, [_p, _i, _p]),
    "dsvgp_elbo_step_timings5": (_i, [_p, _i, _p]),
    "dsvgp_elbo_step_timed_count": (C.c_long, [_p]),
    "dsvgp_elbo_step_locate": (_i, [_p, _i, _p, _p, _p, _p]),
    "dsvgp_version": (C.c_char_p, []),
    "dsvgp_hyp_forward": (_i, [_p, _p, _p, _p, _p]),
    "dsvgp_hyp_backward": (_i, [_p, _p, _p, _p, _p, _p, _p, _p]),
    "dsvgp_step_epilogue": (_i, [_p, _p, _p, _d, _d, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "dsvgp_gemm_lib_f32": (_i, [_p, _i, _i, _i, _i, _f, _p, _l, _p, _l, _f, _p, _l]),
    "dsvgp_ciq_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "dsvgp_ciq_lanczos": (_i, [_p, _p, _l, _p, _i, _i, _p, _p, _p]),
    "dsvgp_ciq_solve": (_i, [_p, _p, _l, _p, _l, _i, _i, _p, _p, _i, _f, _i, _i, _p, _i, _p, _p, _p, _l, _p, _p]),
    "dsvgp_mfma_rate": (_i, [_p, _i, _i, _p, _p]),
    "dsvgp_mfma_rate2": (_i, [_p, _i, _i, _p, _p, _p, _p]),
    "dsvgp_ciq_mix": (_i, [_p, _p, _i, _i, _i, _p, _i, _i, _i, _p, _p, _l]),
    "dsvgp_ciq_cross": (_i, [_p, _p, _i, _i, _p, _i, _i, _p, _i, _i, _p, _p, _p]),
    "dsvgp_ciq_rowstats": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p, _f, _p, _p, _p, _p]),
    "dsvgp_ciq_tbar": (_i, [_p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "dsvgp_sym_average_f32": (_i, [_p, _p, _i, _l, _p, _l]),
    "dsvgp_ciq_workspace_bytes_f64": (_z, [_i, _i, _i, _i]),
    "dsvgp_ciq_lanczos_f64": (_i, [_p, _p, _l, _p, _i, _i, _p, _p, _p]),
    "dsvgp_ciq_solve_f64": (_i, [_p, _p, _l, _p, _l, _i, _i, _p, _p, _i, _d, _i, _i, _p, _i, _p, _p, _p, _l, _p, _p]),
    "dsvgp_ciq_mix_f64": (_i, [_p, _p, _i, _i, _i, _p, _i, _i, _i, _p, _p, _l]),
    "dsvgp_ciq_cross_f64": (_i, [_p, _p, _i, _i, _p, _i, _i, _p, _i, _i, _p, _p, _p]),
    "dsvgp_ciq_rowstats_f64": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p, _d, _p, _p, _p, _p]),
    "dsvgp_ciq_tbar_f64": (_i, [_p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "dsvgp_sym_average_f64": (_i, [_p, _p, _i, _l, _p, _l]),
    "dsvgp_packed_width": (_i, [_i]),
    "dsvgp_column_mean": (_i, [_p, _p, _i, _i, _p]),
    "dsvgp_pack_points": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p]),
    "dsvgp_kernel_fwd": (_i, [_p, _p, _p, _i, _p, _p, _i, _i, _i, _p, _f, _p, _l, _i]),
    "dsvgp_kernel_fwd_canon": (_i, [_p, _p, _p, _i, _p, _p, _i, _i, _i, _p, _i, _p, _p, _l]),
    "dsvgp_kernel_diag": (_i, [_p, _i, _i, _p, _p]),
    "dsvgp_kernel_bwd_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "dsvgp_kernel_bwd": (_i, [_p, _p, _l, _i, _p, _p, _p, _i, _p, _p, _i, _i, _i, _p, _i, _p, _p, _p, _p]),
    "dsvgp_kernel_canon_supported": (_i, [_i, _i]),
    "dsvgp_kernel_canon2_supported": (_i, [_i, _i]),
    "dsvgp_kernel_fwd_canon2": (_i, [_p, _p, _i, _p, _i, _i, _i, _p, _i, _p, _f, _p, _l, _i]),
    "dsvgp_kernel_bwd_canon2": (_i, [_p, _p, _l, _i, _p, _p, _i, _p, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p]),
    "dsvgp_kernel_bwd_canon": (_i, [_p, _p, _l, _i, _p, _p, _p, _i, _p, _p, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p]),
    "dsvgp_pack_points_f64": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p]),
    "dsvgp_kernel_transform_f64": (_i, [_p, _p, _l, _p, _i, _p, _i, _i, _p, _d]),
    "dsvgp_kernel_bwd_transform_f64": (_i, [_p, _p, _l, _p, _l, _p, _i, _p, _i, _i, _p, _p]),
    "dsvgp_kernel_bwd_points_f64": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _i, _p, _p]),
    "dsvgp_colstats_f64": (_i, [_p, _p, _l, _p, _l, _p, _i, _i, _p, _p]),
    "dsvgp_abar_f64": (_i, [_p, _p, _l, _p, _l, _p, _p, _p, _i, _i, _p, _l, _p, _l]),
    "dsvgp_likelihood_terms_f64": (_i, [_p, _p, _p, _p, _p, _i, _i, _p, _i, _d, _p, _p, _p, _p, _p]),
    "dsvgp_elbo_fast_tail_f64": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p, _d, _p, _p, _p]),
    "dsvgp_potrf_workspace_bytes": (_z, [_i, _i]),
    "dsvgp_potrf": (_i, [_p, _p, _i, _l, _p, _i, _p]),
    "dsvgp_add_diag": (_i, [_p, _p, _i, _l, _d]),
    "dsvgp_trsm_workspace_bytes": (_z, [_i, _i, _i]),
    "dsvgp_trsm": (_i, [_p, _p, _l, _i, _i, _p, _l, _i, _i, _p, _l, _p, _l, _i, _p, _i]),
    "dsvgp_trtri": (_i, [_p, _p, _l, _i, _i, _p, _p]),
    "dsvgp_potrf_inverse": (_i, [_p, _p, _i, _l, _p, _p, _i, _p]),
    "dsvgp_split3_kpad": (_i, [_i]),
    "dsvgp_split3_bytes": (_z, [_i, _i]),
    "dsvgp_split3_bf16": (_i, [_p, _p, _l, _i, _i, _i, _p]),
    "dsvgp_gemm3b": (_i, [_p, _i, _i, _i, _i, _f, _p, _i, _p, _i, _p, _l]),
    "dsvgp_widen_f32_f64": (_i, [_p, _p, _l, _p, _l, _i, _i]),
    "dsvgp_gemm": (_i, [_p, _i, _i, _i, _i, _i, _d, _p, _l, _p, _l, _d, _p, _l, _p, _l, _p, _l, _p]),
    "dsvgp_stats_workspace_bytes": (_z, [_i, _i]),
    "dsvgp_predictive_stats": (_i, [_p, _p, _l, _p, _l, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "dsvgp_likelihood_terms": (_i, [_p, _p, _p, _p, _i, _i, _p, _i, _d, _p, _p, _p, _p]),
    "dsvgp_abar": (_i, [_p, _p, _l, _p, _l, _i, _i, _p, _p, _p, _p, _l]),
    "dsvgp_rowdot": (_i, [_p, _p, _l, _i, _i, _p, _p]),
    "dsvgp_kl_terms": (_i, [_p, _p, _p, _l, _i, _d, _p, _p, _p, _l]),
    "dsvgp_phi_symmetrize": (_i, [_p, _p, _i, _l]),
    "dsvgp_transpose_f64": (_i, [_p, _p, _l, _i, _i, _p, _l]),
    "dsvgp_gemv_f64": (_i, [_p, _i, _p, _l, _i, _i, _p, _p]),
    "dsvgp_transpose_f32": (_i, [_p, _p, _l, _i, _i, _p, _l]),
    "dsvgp_residual_terms": (_i, [_p, _p, _p, _i, _p, _d, _p, _p]),
    "dsvgp_trace_terms": (_i, [_p, _p, _l, _p, _l, _p, _l, _i, _f, _p]),
    "dsvgp_elbo_fast_finalize": (_i, [_p, _p, _p, _i, _i, _d, _p]),
    "dsvgp_mirror_lower_f32": (_i, [_p, _p, _i, _l]),
    "dsvgp_add_diag_f32": (_i, [_p, _p, _i, _l, _f]),
    "dsvgp_sminus_i_col": (_i, [_p, _p, _i, _l, _p, _p, _d]),
    "dsvgp_adam_step_multi_dev": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _f, _f, _f, _p]),
    "dsvgp_scale_by_vbar": (_i, [_p, _p, _l, _p, _l, _p, _l, _p, _d]),
    "dsvgp_kl_terms_scaled": (_i, [_p, _p, _p, _l, _i, _d, _i, _p, _d, _p, _p, _p, _l]),
    "dsvgp_variational_terms": (_i, [_p, _p, _p, _l, _i, _d, _i, _p, _d, _p, _l, _f, _p, _p, _p, _p, _l]),
    "dsvgp_tril_pack_f32": (_i, [_p, _p, _l, _i, _p, _i, _p]),
    "dsvgp_tril_unpack_f32": (_i, [_p, _p, _i, _p, _l, _p, _i]),
    "dsvgp_gather_batch": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _i, _p, _p, _p, _p]),
    "dsvgp_adam_step": (_i, [_p, _p, _p, _p, _p, _l, _f, _f, _f, _f, _i]),
    "dsvgp_adam_step_multi": (_i, [_p, _i, _p, _p, _p, _p, _p, _f, _f, _f, _f, _i]),
    "dsvgp_adam_step_multi_guarded": (_i, [_p, _i, _p, _p, _p, _p, _p, _f, _f, _f, _f, _i, _p]),
    "dsvgp_adam_step_multi_f64": (_i, [_p, _i, _p, _p, _p, _p, _p, _d, _d, _d, _d, _i]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)   # AttributeError here == header/library mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args

# GEMM flags (include/dsvgp.h)
TRANS_A, TRANS_B = 1, 2
A_LOWER, A_UPPER, B_LOWER, B_UPPER = 4, 8, 16, 32
OUT_LOWER, B_IS_FLOAT, CIN_IS_FLOAT, K_PADDED, BACKGROUND = 64, 128, 256, 512, 1024


class DsvgpError(RuntimeError):
    pass


ENOSPACE = -4


def check(rc, what):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "matrix not positive definite", -3: "misaligned",
                -4: "caller-sized buffer too small"}.get(rc)
        if kind is None:
            kind = "hipError %d" % (rc - 1000) if rc < 2000 else "rocblas_status %d" % (rc - 2000)
        raise DsvgpError("%s failed: %s (code %d)" % (what, kind, rc))
