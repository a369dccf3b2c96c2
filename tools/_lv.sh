export TMPDIR=/tmp
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; print("%-30s %8.1f M steps/s  sweep %7.1f qp_lane %7.1f  qp(dense) %7.1f" % (sys.argv[1], d["value"]/1e6, k.get("dyn_sweep_us") or 0, k.get("qp_lane_us") or 0, k.get("qp_us") or 0))'
for rep in 1 2; do for v in ${VARIANTS:-0}; do
  lib=$PWD/wbc_quadruped_dob_amd/lib_$v/libwbc_hip.so; [ $v = 0 ] && lib=$PWD/wbc_quadruped_dob_amd/lib/libwbc_hip.so
  for n in ${NS:-262144}; do for cfg in ${CFGS:-2}; do
  WBC_LIB=$lib WBC_FUSED_MAX=0 $B --steps 40 --warmup 5 --batch $n --config $cfg | python -c "$pick" "variant $v cfg$cfg n=$n"
  done; done
done; done
