set -u
export TMPDIR=/tmp
export WBC_TILE_TICK=1 WBC_FUSED_MAX=0
{
echo "# fp64 observer on, tile tick beyond one round: 32-state workgroups of four wavefronts, two per CU (lib_x) against 64-state workgroups of eight, one per CU (lib); tile tick forced"
for a in "--steps 100 --warmup 10 --batch 32768 --config 3" "--steps 60 --warmup 10 --batch 65536 --config 3" "--steps 40 --warmup 5 --batch 131072 --config 3" "--steps 30 --warmup 5 --batch 262144 --config 3" "--steps 100 --warmup 10 --batch 24576 --config 3"; do
  bash tools/ab_r06.sh "$a" lib lib_x 2>&1
done
} > gpurun_out/r06s_ab_tile_tick_f64_obs_two_per_cu.log
cat gpurun_out/r06s_ab_tile_tick_f64_obs_two_per_cu.log
