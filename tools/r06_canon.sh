#!/bin/bash
# round 6: the assembly kernels with wave-scope LDS hand-offs (ASM_WAVE_SYNC) against the workgroup barrier: tests, then A/B on one box
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_canon; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "kernel or assemble or canon or pack" > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
B=$(mktemp -d /tmp/canon_XXXX)
tools/build_variant.sh $B/sync0 "assemble.hip:-DASM_WAVE_SYNC=0" 
for rep in 1 2; do
  for v in head sync0; do
    if [ $v = head ]; then L=$R/gp-derivatives-variational-inference_amd/libdsvgp_hip.so; else L=$B/$v/libdsvgp_hip.so; fi
    echo "== $v" | tee -a $O/ab.txt
    DSVGP_LIB_PATH=$L python tools/canon_probe.py 2>&1 | tee -a $O/ab.txt
    DSVGP_LIB_PATH=$L python tools/assemble_probe.py 2>&1 | tail -1 | tee -a $O/ab.txt
    DSVGP_LIB_PATH=$L python tools/assemble_bwd_probe.py 2>&1 | tail -2 | tee -a $O/ab.txt
  done
done
