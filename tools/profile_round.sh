#!/bin/bash
# rocprofv3 runs behind the numbers of DESIGN.md / bench.py (run on the GPU box through gpurun; results land in
# gpurun_out/prof_<tag>/ and tools/summarise_profiles.py turns them into the tracked summaries under profiles/).
#   tools/profile_round.sh <tag>         e.g. r03
# then: python tools/summarise_profiles.py ... and python tools/pmc_traffic_all.py gpurun_out/prof_<tag> profiles/<tag>_pmc_hbm_traffic.json
# Separate passes: kernel trace + stats per workload; FETCH_SIZE and WRITE_SIZE (one counter per pass, never combined
# with a trace domain other than --kernel-trace) for the headline.
set -o pipefail
TAG=${1:-r05}
PART=${2:-all}   # all | stats | pmc | pmc_shards (a gpurun call is limited to 20 minutes: the two halves fit one call each)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the profiler starts the HIP runtime before python runs: the package's own setdefault would come too late.  Static queues under the
# profiler: with DEBUG_HIP_DYNAMIC_QUEUES=1 (the package default) rocprofv3 runs read 10-40 % low (headline 502, host env 277, cfg 5
# 10.5 against 560 / 470 / 12.3 unprofiled); each workload is its own process here, so the static mapping has nothing to trip over
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8} DEBUG_HIP_DYNAMIC_QUEUES=${DEBUG_HIP_DYNAMIC_QUEUES:-0}
run() {  # name, bench args...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --no-extra --no-cpu-baseline "$@" \
        > $OUT/$name.log 2>&1 || echo "FAILED: $name"
    grep -h '^{' $OUT/$name.log | tail -1 > $OUT/$name.bench.json
    echo "done $name"
}
if [ "$PART" != "pmc" ] && [ "$PART" != "pmc_shards" ]; then
run headline --steps 20 --warmup 3
run cfg2_T200 --steps 10 --warmup 3 --max-cycles 67
run cfg3_es --workload es --steps 10 --warmup 3
run cfg3_es_ext --workload es --extension --steps 10 --warmup 3
run cfg4_dqn_ga --workload dqn-ga --steps 2 --warmup 1
run cfg5_dqn_es --workload dqn-es --steps 2 --warmup 1
run cfg4_dqn_ga_c6 --workload dqn-ga --channels 6 --steps 2 --warmup 1
run cfg5_dqn_es_c6 --workload dqn-es --channels 6 --steps 2 --warmup 1
run cfg2_host_env --env host --steps 40 --warmup 3
run cfg2_shard_1_of_4 --shard-of 4 --steps 20 --warmup 3
run cfg2_shard_1_of_8 --shard-of 8 --steps 20 --warmup 3
run cfg3_host_env --workload es --env host --steps 5 --warmup 2
run cfg4_host_frames --workload dqn-ga --frames host --steps 1 --warmup 1
fi
pmc() {  # workload name, bench args...: one pass per counter (never combined with a trace domain other than --kernel-trace)
    local name=$1; shift
    for ctr in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_${name}_$ctr -- python3 $ROOT/bench.py --no-extra \
            --no-cpu-baseline "$@" > $OUT/pmc_${name}_$ctr.log 2>&1 || echo "FAILED: pmc $name $ctr"
        grep -h '^{' $OUT/pmc_${name}_$ctr.log | tail -1 > $OUT/pmc_${name}_$ctr.bench.json
        echo "done pmc $name $ctr"
    done
}
if [ "$PART" != "stats" ] && [ "$PART" != "pmc_shards" ]; then
pmc headline --steps 5 --warmup 2
pmc cfg3_es --workload es --steps 3 --warmup 1
pmc cfg4_dqn_ga --workload dqn-ga --steps 1 --warmup 1
pmc cfg5_dqn_es --workload dqn-es --steps 1 --warmup 1
pmc cfg4_dqn_ga_c6 --workload dqn-ga --channels 6 --steps 1 --warmup 1
pmc cfg5_dqn_es_c6 --workload dqn-es --channels 6 --steps 1 --warmup 1
fi
if [ "$PART" = "pmc_shards" ] || [ "$PART" = "all" ]; then   # the persistent rollout launches of a rank of 4 / 8
pmc cfg2_shard_1_of_4 --shard-of 4 --steps 5 --warmup 2
pmc cfg2_shard_1_of_8 --shard-of 8 --steps 5 --warmup 2
fi
