set -u
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_pytest_gpu.log 2>&1; tail -3 gpurun_out/r06_pytest_gpu.log
{
echo "# output rows (M, h, Jc, pf) of the sweep body stored non-temporally (lib_nt, -DWBC_NT_STORES=1) against plain stores (lib); alternating, one MI355X"
for a in "--steps 200 --warmup 20 --batch 32768 --config 4" "--steps 40 --warmup 5 --batch 262144 --config 4" "--steps 40 --warmup 5 --batch 262144" "--steps 100 --warmup 10 --batch 32768" "--steps 100 --warmup 10 --batch 65536" "--steps 200 --warmup 20 --batch 16384"; do
  bash tools/ab_r06.sh "$a" lib lib_nt 2>&1
done
} > gpurun_out/r06l_ab_nt_stores.log
cat gpurun_out/r06l_ab_nt_stores.log | cut -c1-175
