#!/bin/bash
# in-kernel phase clocks of the resident sumcheck kernel by size bucket (profiling build of sumcheck.hip, made on the box)
mkdir -p gpurun_out
cp longfellow-zk_amd/liblfgpu.so /tmp/liblfgpu_orig.so
for B in "0 64" "64 1024" "1024 100000000"; do
  set -- $B
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -DLF_SC_PROF -DLF_SC_PROF_LO=$1 -DLF_SC_PROF_HI=$2 -c -o /tmp/sc_prof.o longfellow-zk_amd/csrc/sumcheck.hip || exit 1
  objs=$(ls longfellow-zk_amd/build/*.o | grep -v sumcheck.hip.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o longfellow-zk_amd/liblfgpu.so $objs /tmp/sc_prof.o || exit 1
  echo "== bucket ($1, $2]" >> gpurun_out/sc_phases.txt
  timeout -k 10 300 python tools/bench_zk.py 1 4 > /dev/null 2> /tmp/err.txt || { tail -5 /tmp/err.txt; exit 1; }
  grep "sc_grid phases" /tmp/err.txt | tail -2 >> gpurun_out/sc_phases.txt
done
cp /tmp/liblfgpu_orig.so longfellow-zk_amd/liblfgpu.so
cat gpurun_out/sc_phases.txt
