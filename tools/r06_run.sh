set -u
export TMPDIR=/tmp
pick='import sys,json; d=json.loads(sys.stdin.read()); print("%-40s %9.1f M steps/s  %8.4f ms/step" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"]))'
{
echo "# one device, one process: the batch as ONE solver against k shards of wbc_multi_* on the SAME device (each shard its own stream, free-running from tick to tick: a shard's HBM-bound sweep runs under another shard's latency-bound QP kernels)"
for spec in "262144 1" "131072 2" "65536 4" "32768 8"; do set -- $spec
  if [ $2 = 1 ]; then python bench.py --steps 40 --warmup 5 --batch $1 --no-cpu --no-latency --large-batch 0 --no-closed-loop 2>/dev/null | python -c "$pick" "fp64 obs off, 1 x $1"
  else python bench.py --gpus $2 --single-process --batch $1 --steps 40 --warmup 5 2>/dev/null | python -c "$pick" "fp64 obs off, $2 x $1 (wbc_multi, one device)"; fi
done
for spec in "262144 1" "131072 2" "65536 4"; do set -- $spec
  if [ $2 = 1 ]; then python bench.py --config 3 --steps 40 --warmup 5 --batch $1 --no-cpu --no-latency --large-batch 0 --no-closed-loop 2>/dev/null | python -c "$pick" "fp64 obs on, 1 x $1"
  else python bench.py --config 3 --gpus $2 --single-process --batch $1 --steps 40 --warmup 5 2>/dev/null | python -c "$pick" "fp64 obs on, $2 x $1 (wbc_multi, one device)"; fi
done
for spec in "262144 1" "131072 2" "65536 4" "32768 8"; do set -- $spec
  if [ $2 = 1 ]; then python bench.py --config 4 --steps 40 --warmup 5 --batch $1 --no-cpu --no-latency --large-batch 0 --no-closed-loop 2>/dev/null | python -c "$pick" "fp32 obs on, 1 x $1"
  else python bench.py --config 4 --gpus $2 --single-process --batch $1 --steps 40 --warmup 5 2>/dev/null | python -c "$pick" "fp32 obs on, $2 x $1 (wbc_multi, one device)"; fi
done
} > gpurun_out/r06p_shards_on_one_device.log 2>&1
cat gpurun_out/r06p_shards_on_one_device.log
