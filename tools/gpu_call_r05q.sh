#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lch or rs or tower or gf" > gpurun_out/x3_tests.log 2>&1 || { tail -30 gpurun_out/x3_tests.log; exit 1; }
tail -2 gpurun_out/x3_tests.log
timeout -k 10 300 python tools/bench_lch.py 1024 20 5 > gpurun_out/bench_lch_x3.txt 2>&1 || { tail -5 gpurun_out/bench_lch_x3.txt; exit 1; }
tail -5 gpurun_out/bench_lch_x3.txt
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lch_x3 -o z -- python3 tools/bench_lch.py 1024 20 5 > /dev/null 2>&1 || true
f=$(find gpurun_out/prof_lch_x3 -name "z_kernel_stats.csv" | head -1); head -6 "$f" | cut -c1-120; cp "$f" gpurun_out/lch_x3_kernel_stats.csv; rm -rf gpurun_out/prof_lch_x3
