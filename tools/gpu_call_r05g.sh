#!/bin/bash
mkdir -p gpurun_out
for wt in 0 1; do
  LFGPU_VERBOSE=1 LFGPU_P256_WAVE_TAIL=$wt timeout -k 10 300 python tools/bench_zk.py 1 2 --mdoc-sig > /dev/null 2> gpurun_out/sig_verbose_wt${wt}.err || exit 1
  echo "== wave_tail $wt"; grep "sumcheck_layer256 grid" gpurun_out/sig_verbose_wt${wt}.err | tail -21 | cut -c1-140
done
