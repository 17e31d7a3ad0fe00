#!/bin/bash
# rocprofv3 runs behind the numbers of DESIGN.md / bench.py (run on the GPU box through gpurun; results land in
# gpurun_out/prof_<tag>/ and tools/summarise_profiles.py turns them into the tracked summaries under profiles/).
#   tools/profile_round.sh <tag>         e.g. r02
# Separate passes: kernel trace + stats per workload; FETCH_SIZE and WRITE_SIZE (one counter per pass, never combined
# with a trace domain other than --kernel-trace) for the headline.
set -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the profiler starts the HIP runtime before python runs: the package's own setdefault would come too late
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
run() {  # name, bench args...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --no-extra --no-cpu-baseline "$@" \
        > $OUT/$name.log 2>&1 || echo "FAILED: $name"
    grep -h '^{' $OUT/$name.log | tail -1 > $OUT/$name.bench.json
    echo "done $name"
}
run headline --steps 20 --warmup 3
run cfg2_T200 --steps 10 --warmup 3 --max-cycles 67
run cfg3_es --workload es --steps 10 --warmup 3
run cfg3_es_ext --workload es --extension --steps 10 --warmup 3
run cfg4_dqn_ga --workload dqn-ga --steps 2 --warmup 1
run cfg5_dqn_es --workload dqn-es --steps 2 --warmup 1
for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_$ctr -- python3 $ROOT/bench.py --no-extra --no-cpu-baseline \
        --steps 5 --warmup 2 > $OUT/pmc_$ctr.log 2>&1 || echo "FAILED: pmc $ctr"
    echo "done pmc $ctr"
done
