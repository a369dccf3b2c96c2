#!/bin/bash
# Instruction-cache counters of the fused tick and the persistent rollout kernel (one counter group per pass).
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/icache"; rm -rf "$O"; mkdir -p "$O"
cd /tmp
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O" -o tick_$T -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu --no-latency --large-batch 0 > /dev/null 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O" -o roll_$T -- python3 "$R/bench.py" --config 5 --steps 10 --warmup 2 > /dev/null 2>&1
  WBC_ROLLOUT_PERSISTENT=0 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O" -o tickroll_$T -- python3 "$R/bench.py" --config 5 --steps 10 --warmup 2 > /dev/null 2>&1
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections, os
for f in sorted(glob.glob(sys.argv[1] + "/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", os.path.basename(f))
    for k, d in agg.items():
        if "wbc" in k: print("  ", k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n", len(next(iter(d.values()))))
PY
