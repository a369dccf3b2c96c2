#!/bin/bash
# What the device runs at while the N = 262 144 tick is timed: samples rocm-smi (clocks, package power, temperatures) five times a
# second beside a long bench run.  The boxes of the pool differ by up to 15 % on this workload; this records what can be seen
# of the cause from user space (usage: tools/clock_probe.sh > gpurun_out/clock_probe.log).
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"
python3 "$R/bench.py" --no-cpu --no-latency --large-batch 0 --steps 6000 --warmup 20 --batch 262144 > /tmp/clock_probe_bench.json 2>/dev/null &
BP=$!
sleep 2
for i in $(seq 1 60); do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Package Power|junction|memory" | sed -E 's/GPU\[0\]\s*: //; s/Temperature \(Sensor (junction|memory)\) \(C\)/T_\1/; s/ clock level: [0-9S]+: \(([0-9]+)Mhz\)/ \1 MHz/; s/Current Socket Graphics Package Power \(W\)/power W/' | tr '\n' ';'; echo
  kill -0 $BP 2>/dev/null || break
  sleep 0.2
done
wait $BP
python3 -c "
import json; d=json.load(open('/tmp/clock_probe_bench.json')); k=d['kernels']
print('bench: %.1f M steps/s, %.4f ms/step, sweep %.1f us, per-lane QP %.1f us, list %.1f us' % (d['value']/1e6, d['ms_per_step'], k['dyn_sweep_us'], k['qp_lane_us'] or 0, k['qp_us']))"
