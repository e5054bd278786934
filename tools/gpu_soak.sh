# GPU box: soak of the final tree (every proof verified / wire bytes compared).  Usage: gpurun --timeout 900 -- 'bash tools/gpu_soak.sh [outdir]'
set -e
O=${1:-gpurun_out/soak}
mkdir -p $O
{
echo "tools/stress_zk.py flatsha_nb1 3000:"; timeout -k 10 200 python tools/stress_zk.py flatsha_nb1 3000 2>&1 | tail -1
echo "tools/stress_zk.py flatsha_nb32 1500:"; timeout -k 10 200 python tools/stress_zk.py flatsha_nb32 1500 2>&1 | tail -1
echo "tools/stress_zk.py mdoc_hash 300:"; timeout -k 10 200 python tools/stress_zk.py mdoc_hash 300 2>&1 | tail -1
echo "tools/stress_zk256.py 300:"; timeout -k 10 200 python tools/stress_zk256.py 300 2>&1 | tail -1
echo "tools/zk_throughput.py --k 16 --seconds 20:"; timeout -k 10 300 python tools/zk_throughput.py --jobs flatsha32,mdoc --k 16 --seconds 20 2>&1 | grep -v amdgpu.ids | tail -3
echo "smoke():"; python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
} > $O/soak.txt 2>&1
cat $O/soak.txt
