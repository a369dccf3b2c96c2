#!/bin/bash
# round 3, GPU run 1: all GPU tests; A/B explicit R^-1 dual step (lib) vs back-substitution (lib_norinv); packed fp32 sweep on / off
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03_run1"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -m pytest tests -q -m gpu -x > "$O/pytest_gpu_full.log" 2>&1; tail -15 "$O/pytest_gpu_full.log"
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-34s %8.1f M/s %8.4f ms/step fused %s sweep %s qp %s lane %s rnea %s it %.2f max %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("fused_tick_us")), f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us")), f(k.get("rnea_step_us")), (d.get("qp") or {}).get("iters_mean", 0) or 0, (d.get("qp") or {}).get("iters_max")))'
for rep in 1 2; do
for L in lib lib_norinv; do
  export WBC_LIB=$R/wbc_quadruped_dob_amd/$L/libwbc_hip.so
  $B --steps 300 --warmup 30 | python -c "$pick" "$L cfg2 n4096" >> "$O/ab_rinv.log"
  $B --steps 300 --warmup 30 --config 3 | python -c "$pick" "$L cfg3 n4096" >> "$O/ab_rinv.log"
  $B --steps 100 --warmup 10 --batch 32768 | python -c "$pick" "$L cfg2 n32768" >> "$O/ab_rinv.log"
  $B --steps 100 --warmup 10 --batch 16384 | python -c "$pick" "$L cfg2 n16384" >> "$O/ab_rinv.log"
  $B --steps 50 --warmup 5 --batch 262144 | python -c "$pick" "$L cfg2 n262144" >> "$O/ab_rinv.log"
  python bench.py --config 5 --steps 50 --warmup 5 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "cfg5 h20 n1024: %.1f M steps/s, %.2f us/tick" % (d["value"]/1e6, d["us_per_tick"]))' $L >> "$O/ab_rinv.log"
done
done
unset WBC_LIB
cat "$O/ab_rinv.log"
for rep in 1 2; do
for PK in 1 -1; do
  export WBC_F32_PACK2=$PK
  for n in 8192 16384 32768 65536 131072 262144; do
    st=$(( 3000000 / n + 20 ))
    $B --steps $st --warmup 10 --batch $n --config 4 | python -c "$pick" "pack=$PK cfg4 f32 n$n" >> "$O/ab_pack.log"
  done
  WBC_OBS_SPLIT_MIN=1 $B --steps 100 --warmup 10 --batch 32768 --config 4 | python -c "$pick" "pack=$PK cfg4 f32 n32768 obs-split" >> "$O/ab_pack.log"
  $B --steps 100 --warmup 10 --batch 32768 --config 2 --dtype f32 | python -c "$pick" "pack=$PK cfg2 f32 n32768 obs-off" >> "$O/ab_pack.log"
  $B --steps 50 --warmup 5 --batch 262144 --config 2 --dtype f32 | python -c "$pick" "pack=$PK cfg2 f32 n262144 obs-off" >> "$O/ab_pack.log"
done
done
unset WBC_F32_PACK2
cat "$O/ab_pack.log"
# dynamics stage alone, fp32, both forms (events of the dispatch)
python - <<'PY' > "$O/dyn_alone_f32.log" 2>&1
import numpy as np, torch, wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth
model = W.Model.from_urdf(W.SYNTHETIC_URDF)
for n in (32768, 262144):
    B = synth.make_batch(2, n, model.total_mass)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(torch.float32).cuda()
    q, v = dev(B["q"]), dev(B["v"])
    for pk in (1, -1):
        s = W.Solver(model, W.Params.from_dict(synth.default_params(dtype="f32"), "f32"), dtype="f32", device=0, max_batch=n, options={"f32_pack2": pk})
        out = s.dynamics(q, v, want=("M", "h", "Jc"))
        for _ in range(20): s.dynamics(q, v, want=("M", "h", "Jc"), out=out)
        torch.cuda.synchronize(); s.enable_timing(1)
        for _ in range(100): s.dynamics(q, v, want=("M", "h", "Jc"), out=out)
        torch.cuda.synchronize(); tm = s.collect_timing()
        t = tm["dyn_ms"] / tm["dyn_launches"] * 1e3
        print("dyn alone f32 n=%d pack=%d: %.1f us  %.2f TB/s of 443 words (frac %.3f)" % (n, pk, t, 443 * 4 * n / t / 1e6, 443 * 4 * n / t / 1e6 / 8))
PY
cat "$O/dyn_alone_f32.log"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_cfg4_n32768 -- python3 "$R/bench.py" --steps 100 --warmup 10 --no-cpu --no-latency --large-batch 0 --batch 32768 --config 4 > "$O/bench_under_rocprof_cfg4_n32768.json" 2> "$O/rocprof.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_cfg4_n262144 -- python3 "$R/bench.py" --steps 50 --warmup 5 --no-cpu --no-latency --large-batch 0 --batch 262144 --config 4 > "$O/bench_under_rocprof_cfg4_n262144.json" 2>> "$O/rocprof.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_n4096 -- python3 "$R/bench.py" --steps 300 --warmup 30 --no-cpu --no-latency --large-batch 0 > "$O/bench_under_rocprof_n4096.json" 2>> "$O/rocprof.err"
find "$O" -name "*kernel_trace.csv" -delete
head -8 "$O"/stats_cfg4_n32768_kernel_stats.csv "$O"/stats_cfg4_n262144_kernel_stats.csv "$O"/stats_n4096_kernel_stats.csv
cd "$R"
python bench.py --steps 20 --warmup 5 > "$O/bench_default_driver_flags.json" 2> "$O/bench_default.err"; tail -c 3000 "$O/bench_default_driver_flags.json"
