#!/bin/bash
# where does the round-hand time of the resident sumcheck kernel go on the host side?
mkdir -p gpurun_out
LFGPU_VERBOSE=1 timeout -k 10 300 python tools/bench_zk.py 1 3 > gpurun_out/zk1_verbose.json 2> gpurun_out/zk1_verbose.err || exit 1
grep "sumcheck_layer" gpurun_out/zk1_verbose.err | tail -13
tail -c 600 gpurun_out/zk1_verbose.json
