#!/bin/bash
# tools only (round 6): first GPU pass of the round -- the new tests, the driver's command with the new bench keys, the end-to-end training runs
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r06_first; mkdir -p $O
python3 -m pytest tests/test_gpu_train.py tests/test_gpu_ops.py tests/test_gpu_parallel.py tests/test_gpu_step.py -m gpu -x -q -rP \
  -k "train or trajectory or row_ranges or not_16_byte or self_launch or kernel_bwd_matches or kernel_fwd_random" > $O/tests.txt 2>&1
echo "tests rc=$?"; tail -3 $O/tests.txt; grep -h "^\[train\]\|^\[dp\]" $O/tests.txt
python3 tools/train_quality.py --config c4 --epochs 1 --data welch --out $O/train_c4_welch.json > $O/train_c4_welch.log 2>&1; echo "train c4 welch rc=$?"; tail -c 600 $O/train_c4_welch.log
python3 tools/train_quality.py --config c4 --epochs 1 --data sin --out $O/train_c4_sin.json > $O/train_c4_sin.log 2>&1; echo "train c4 sin rc=$?"
python3 tools/train_quality.py --config c2 --epochs 2 --against-oracle 50 --n-test 2000 --out $O/train_c2.json > $O/train_c2.log 2>&1; echo "train c2 rc=$?"
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -1 $O/bench_default.json | cut -c1-300
