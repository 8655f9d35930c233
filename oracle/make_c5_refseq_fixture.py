"""C5 ``init`` state, inducing-point / direction gradients: float64 truth and the reference's own fp32 op sequence.

At the state ``train_gp(use_ciq=True)`` constructs (lengthscale = 1/1024 in d = 50, reference directional_vi.py:58-60) the
kernel backward multiplies fp32 cancellation residue of r . v inner products by 1/lengthscale^2 = 1e6, so dZ / dV of ANY fp32
evaluation are far from the float64 values.  tests/golden/c5_init_refseq.npz makes the comparison numerical instead of a flat
tolerance:
    g64_inducing_points / g64_inducing_directions   float64 oracle (pair-wise difference form) on the float32 inputs
    err_refseq32_*   max-norm relative error of the SAME step evaluated in float32 with the reference's kernel op sequence
                     (``kernel_matrix_refseq``: RBFKernelDirectionalGrad.py:57-107 matmul / bmm projections, the value block
                     through gpytorch's centred quadratic expansion) -- what the reference itself delivers in its default dtype
    err_pairwise32_* the same for the oracle's pair-wise form in float32 (the form behind tests/golden/c5_step_init.npz)
tests/test_ciq.py holds the HIP step's dZ / dV error against the float64 values to "no worse than err_refseq32".
Usage: python oracle/make_c5_refseq_fixture.py      (three full-size CIQ oracle runs: ~40 min on 8 cores, ~35 GB)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dsvgp_oracle as O
from make_c5_fixture import Q, make_inputs

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "c5_init_refseq.npz")
KEYS = ("inducing_points", "inducing_directions")


def relmax(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max()).item()


def main():
    torch.set_num_threads(os.cpu_count())
    P, x, y, D, nd = make_inputs("init")
    t0 = time.time()
    P64 = {k: v.double() for k, v in P.items()}
    _, g64, _, _ = O.ciq_loss_and_grads(P64, x.double(), y.double(), D.double(), nd, Q=Q)
    print("float64 oracle: %.0f s" % (time.time() - t0), flush=True)
    out = {"g64_" + k: g64[k].numpy() for k in KEYS}
    del P64
    for tag, assembly in (("refseq32", O.kernel_matrix_refseq), ("pairwise32", None)):
        t0 = time.time()
        _, g, _, _ = O.ciq_loss_and_grads(P, x, y, D, nd, Q=Q, assembly=assembly)
        for k in KEYS:
            out["err_%s_%s" % (tag, k)] = np.float64(relmax(g[k], g64[k]))
        print("%s: %.0f s, %s" % (tag, time.time() - t0, {k: out["err_%s_%s" % (tag, k)] for k in KEYS}), flush=True)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
