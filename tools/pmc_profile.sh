#!/bin/bash
# PMC passes (one counter group per pass, as MI355X_MICROARCH.md prescribes) for the bench and the calibration kernels.
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
OUT="$R/gpurun_out/pmc"
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT" -o calib_$C -- "$R/tools/calib_copy.bin" > "$OUT/calib_$C.log" 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT" -o n4096_$C -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu --no-latency --large-batch 0 > "$OUT/n4096_$C.log" 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT" -o n262144_$C -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu --no-latency --large-batch 0 --batch 262144 > "$OUT/n262144_$C.log" 2>&1
  # the pair tick (fused_pair_kernel): 8 192 fp64 states, observer off -- its spill traffic is in these bytes
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT" -o n8192_$C -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu --no-latency --large-batch 0 --no-closed-loop --batch 8192 > "$OUT/n8192_$C.log" 2>&1
  # fp32 (configs[3] arithmetic): the per-GPU batch of the 8-GPU config and the whole batch; dynamics stage alone at 262 144
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT" -o n32768f32_$C -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu --no-latency --large-batch 0 --batch 32768 --config 4 > "$OUT/n32768f32_$C.log" 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT" -o n262144f32_$C -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu --no-latency --large-batch 0 --batch 262144 --config 4 > "$OUT/n262144f32_$C.log" 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT" -o n262144dyn_$C -- python3 "$R/tools/dyn_only.py" 262144 f64 > "$OUT/n262144dyn_$C.log" 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT" -o n262144dynf32_$C -- python3 "$R/tools/dyn_only.py" 262144 f32 > "$OUT/n262144dynf32_$C.log" 2>&1
done
ls -la "$OUT"
python3 "$R/tools/pmc_summarize.py" "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
