#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lch_x3 -o z -- python3 tools/bench_lch.py 1024 20 5 > gpurun_out/prof_lch_x3.log 2>&1 || { tail -5 gpurun_out/prof_lch_x3.log; exit 1; }
f=$(find gpurun_out/prof_lch_x3 -name "z_kernel_stats.csv" | head -1); head -5 "$f" | cut -c1-140; cp "$f" gpurun_out/lch_x3_kernel_stats.csv; rm -rf gpurun_out/prof_lch_x3
