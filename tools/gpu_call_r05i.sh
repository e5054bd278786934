#!/bin/bash
# stress of the resident sumcheck kernels with the single-wave tails: GF2_128 (1 and 32 blocks), Fp128, Fp256Base
mkdir -p gpurun_out
: > gpurun_out/stress_wave_tail.jsonl
timeout -k 10 300 python tools/stress_zk.py flatsha_nb1 6000 >> gpurun_out/stress_wave_tail.jsonl 2> gpurun_out/stress1.err || { tail -5 gpurun_out/stress1.err; exit 1; }
timeout -k 10 300 python tools/stress_zk.py flatsha_nb32 3000 >> gpurun_out/stress_wave_tail.jsonl 2> gpurun_out/stress32.err || { tail -5 gpurun_out/stress32.err; exit 1; }
timeout -k 10 300 python tools/stress_zk.py flatsha_fp_nb1 4000 >> gpurun_out/stress_wave_tail.jsonl 2> gpurun_out/stressfp.err || { tail -5 gpurun_out/stressfp.err; exit 1; }
timeout -k 10 400 python tools/stress_zk256.py 1500 >> gpurun_out/stress_wave_tail.jsonl 2> gpurun_out/stress256.err || { tail -5 gpurun_out/stress256.err; exit 1; }
cat gpurun_out/stress_wave_tail.jsonl
