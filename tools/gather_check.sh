#!/bin/bash
# the N > 1 legs on one rank with the hipGraph forms of the gather, and the single-process two-shard line (tools/gather_check.sh [pytest -k expression])
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/gather_check"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -m pytest tests/test_gpu_multi.py tests/test_gpu_bench_contract.py -q -x -k "${1:-multisolver or scale_legs}" > "$O/pytest.log" 2>&1; tail -3 "$O/pytest.log"
WBC_BENCH_GRAPH_GATHER=1 WBC_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-latency --large-batch 0 > "$O/bench_scale_legs_1rank.json" 2>> "$O/bench.err"
python - <<PY
import json
d = json.load(open("$O/bench_scale_legs_1rank.json"))
fmt = lambda g: "  ".join("%s %s" % (k, ("%.1f M" % (g[k]["value"]/1e6)) if "value" in g[k] else g[k].get("error")) for k in ("eager_serial", "eager_overlapped", "graph_ticks_only", "graph_serial", "graph_overlapped"))
g = d["with_tau_allgather"]; print("cfg2 n4096: value %.1f M  long blocks %.1f M  with gather %.1f M (%.3f of value, %s)\n    %s" % (d["value"]/1e6, d["value_long_blocks"]["value"]/1e6, g["value"]/1e6, g.get("frac_of_graph_ticks_only", 0), g["value_is"], fmt(g)))
c = d["scale_config3"]; g = c["with_tau_allgather"]; print("cfg4 f32 n32768: value %.1f M  with gather %.1f M (%.3f, %s)\n    %s" % (c["value"]/1e6, g["value"]/1e6, g.get("frac_of_graph_ticks_only", 0), g["value_is"], fmt(g)))
for k, v in d["scale_config5"].items(): print(k, "%.1f M steps/s %.2f us/tick" % (v["value"]/1e6, v["us_per_tick"]), "roofline frac", (v.get("roofline") or {}).get("frac"))
PY
python bench.py --gpus 2 --single-process --steps 100 --warmup 10 > "$O/bench_single_process_2shards.json" 2>> "$O/bench.err"
python -c "
import json; d=json.load(open('$O/bench_single_process_2shards.json')); g=d['with_tau_allgather']; print('single-process 2 shards: value %.1f M, gather async %.1f M, serial %.1f M' % (d['value']/1e6, g['value']/1e6, g['serial']['value']/1e6))"
tail -5 "$O/bench.err"
