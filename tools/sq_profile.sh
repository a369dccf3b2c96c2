#!/bin/bash
# SQ counter passes (<= 8 SQ counters per pass) for the bench at a given batch: tools/sq_profile.sh <batch> <tag> [passes]
# (kernel-selection switches travel through the WBC_* variables the Python binding turns into wbc_solver_options)
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; B=${1:-262144}; TAG=${2:-sq}; PASSES=${3:-"1 2 3 4"}
OUT="$R/gpurun_out/$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR"
P3="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P4="SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
for i in $PASSES; do
  eval P=\$P$i
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT" -o p$i -- python3 "$R/bench.py" --steps 6 --warmup 2 --no-cpu --no-latency --large-batch 0 --batch $B ${EXTRA:-} > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections,json,os
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out,'**','*counter_collection.csv'),recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
res={k:{c:sum(v)/len(v) for c,v in d.items()} for k,d in acc.items() if 'wbc' in k}
json.dump(res,open(os.path.join(out,'summary.json'),'w'),indent=1)
for k,d in res.items():
    print(k)
    for c in sorted(d): print('   %-28s %14.0f'%(c,d[c]))
PY
