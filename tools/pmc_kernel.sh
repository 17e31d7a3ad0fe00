#!/bin/bash
# tools/pmc_kernel.sh OUTDIR KERNEL_SUBSTR "COUNTER_SET_1" "COUNTER_SET_2" ... -- python3 script args
# One rocprofv3 --pmc pass per counter set (no trace domains besides --kernel-trace), then the per-launch mean of every
# counter for kernels whose name contains KERNEL_SUBSTR.  Run on the GPU box (under gpurun).
out=$1; kern=$2; shift 2
sets=()
while [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
# the profiler starts the HIP runtime before python runs: the package's own setdefault would come too late
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8} DEBUG_HIP_DYNAMIC_QUEUES=${DEBUG_HIP_DYNAMIC_QUEUES:-0}
mkdir -p "$out"
i=0
for s in "${sets[@]}"; do
  rocprofv3 --kernel-trace --pmc $s --output-format csv -d "$out/pass$i" -- "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i ($s) failed"; tail -5 "$out/pass$i.log"; }
  i=$((i+1))
done
python3 - "$out" "$kern" <<'PY'
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
