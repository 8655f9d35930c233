#!/bin/bash
# A/B of the Cholesky chain: fused launches (DSVGP_POTRF_HYBRID=0) against the hybrid chain (persistent diagonal workgroup), same box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for h in 0 1; do
  echo "hybrid=$h $(DSVGP_POTRF_HYBRID=$h timeout -k 10 120 python3 tools/potrf_inv_probe.py 2>&1 | tail -1)"
done; done
