#!/bin/bash
# tools only: build potrf.hip with -DPOTRF_DEBUG and print in-kernel cycles / clock of the 64x64 factor kernel
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; C=$R/gp-derivatives-variational-inference_amd/csrc; B=/tmp/potrf_dbg; mkdir -p $B
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-pass-failed -I$R/include -I$C"
for f in gemm elbo assemble api; do hipcc $FL -c $C/$f.hip -o $B/$f.o & done; wait
hipcc $FL -DPOTRF_DEBUG ${POTRF_DEFS} -c $C/potrf.hip -o $B/potrf.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libdsvgp_hip.so $B/potrf.o $B/assemble.o $B/gemm.o $B/elbo.o $B/api.o -L/opt/rocm/lib -lrocsolver -lrocblas
DSVGP_LIB_PATH=$B/libdsvgp_hip.so python - <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, dsvgp_amd
ops = dsvgp_amd._ops
dev = torch.device("cuda", 0); ctx = ops.Context.get(dev)
g = torch.Generator().manual_seed(0)
n = 3000
Q = torch.randn(n, n, generator=g, dtype=torch.float64)
K = (Q @ Q.t() / n + torch.eye(n, dtype=torch.float64)).to(dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
for rep in range(3):
    A = K.clone(); ops.potrf_(ctx, A, info, 1); torch.cuda.synchronize()
ws = ops._potrf_ws[(0, n)].view(torch.float64).view(-1, 64, 64)
cyc, rt = ws[:, 0, 62].cpu(), ws[:, 0, 63].cpu()
print("potf2_inv_kernel: cycles/block median %.0f, realtime %.1f us, clock %.2f GHz, cycles/column %.0f" % (
    cyc.median().item(), rt.median().item() / 100, (cyc / rt * 0.1).median().item(), cyc.median().item() / 64))
PY
