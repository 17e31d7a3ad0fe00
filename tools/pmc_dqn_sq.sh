#!/bin/bash
# tools/pmc_dqn_sq.sh OUT.txt : SQ / MFMA counters of the shipped DeepQN kernels on the cfg 4 agent-step (1010 frames, 90 tasks,
# 60 nets) and the cfg 5 cohort launch, per-launch means (tools/pmc_kernel.sh: one rocprofv3 --pmc pass per counter set, no trace
# domain besides --kernel-trace).  Run on the GPU box through gpurun.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=${1:-$ROOT/gpurun_out/r05_pmc_dqn_sq.txt}
W=$ROOT/gpurun_out/pmc_dqn_sq; mkdir -p $W
SETS=("SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
{
echo "# tools/pmc_dqn_sq.sh: tools/pmc_kernel.sh (one rocprofv3 --pmc pass per counter set, --kernel-trace only) -- python3 tools/bench_dqn_shapes.py --shape <s> --fc1 <layout> --reps 5"
echo "# per-launch means; MI355X; library version $(grep -o 'COEVO_VERSION [0-9]*' $ROOT/include/coevo.h)"
for spec in "cfg4 tiled dqn_conv_kernel" "cfg4 tiled dqn_fc1_tiled_kernel" "cfg4 streamed dqn_fc1_kernel" "cfg5 streamed dqn_fc1_kernel" "eval tiled dqn_fc1_narrow_tiled_kernel"; do
  set -- $spec
  echo "## $3  (--shape $1 --fc1 $2)"
  if [ ! -d $W/$1_$2 ]; then
    bash $ROOT/tools/pmc_kernel.sh $W/$1_$2 "$3" "${SETS[@]}" -- python3 $ROOT/tools/bench_dqn_shapes.py --shape $1 --fc1 $2 --reps 5
  else   # the passes of this (shape, layout) exist: only re-aggregate for another kernel
    python3 - $W/$1_$2 "$3" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:36s} mean/launch {sum(v) / len(v):16.1f}   launches {len(v)}")
PY
  fi
done
} > $OUT 2>&1
echo "wrote $OUT"
