# PMC evidence for K1 (fp_fft_tile) and K2 (bs_*): separate --pmc passes, no trace domains, program directly after `--`
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02b
mkdir -p $O
K1="python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-secondary"
K2="python3 tools/bench_lch.py 1024 20 5"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/k1_p$i -o p -- $K1 > $O/k1_p$i.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/k2_p$i -o p -- $K2 > $O/k2_p$i.log 2>&1 || exit 1
done
python3 tools/pmc_summary.py $O/pmc_k1.json fp_fft_tile $O/k1_p1 $O/k1_p2 $O/k1_p3 $O/k1_p4 $O/k1_p5 > /dev/null
python3 tools/pmc_summary.py $O/pmc_k2.json bs_ $O/k2_p1 $O/k2_p2 $O/k2_p3 $O/k2_p4 $O/k2_p5 > /dev/null
cat $O/pmc_k1.json $O/pmc_k2.json
# production-mode sumcheck profile (grid kernel, ordinary launch): must exit cleanly now
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/zk1 -o z -- python3 tools/bench_zk.py 1 3 --no-cpu > $O/zk1.log 2>&1; echo "zk1 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/zk32 -o z -- python3 tools/bench_zk.py 32 3 --no-cpu > $O/zk32.log 2>&1; echo "zk32 rc=$?"
tail -2 $O/zk1.log | cut -c1-600
find $O -name "*.csv" -size +20M -delete
