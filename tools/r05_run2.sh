mkdir -p gpurun_out/r05_potrf
tools/potrf_variants.sh "il1_minw2:" "il1_minw1:-DPOTRF_MINW=1" "il0:-DPOTRF_IL=0" "il1_minw1:-DPOTRF_MINW=1" "il0:-DPOTRF_IL=0" > gpurun_out/r05_potrf/il_ab2.txt 2>&1; cat gpurun_out/r05_potrf/il_ab2.txt
B=$(mktemp -d /tmp/pv_XXXX); tools/build_variant.sh $B "potrf.hip:-DPOTRF_MINW=1" > /dev/null 2>&1
DSVGP_LIB_PATH=$B/libdsvgp_hip.so tools/potrf_inv_trace.sh 3000 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_parallel.py -x -q -k "virtual" 2>&1 | tail -15
