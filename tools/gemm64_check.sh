#!/bin/bash
# tools only: build a library that routes EVERY eligible fp64 product through gemm64.hip and run the GPU test suite against it
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; B=$(mktemp -d /tmp/g64_build_XXXX)
$R/tools/build_variant.sh $B "gemm.hip:-DGEMM64=1 -DGEMM64_MIN_TILES=1 -DGEMM64_MIN_K=1"
DSVGP_LIB_PATH=$B/libdsvgp_hip.so python -m pytest $R/tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15
