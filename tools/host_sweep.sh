#!/bin/bash
# host-cores env mode (bench.py --env host, cfg 2): generations/s for "P T K" triples = COEVO_HOST_PREQUEUE, _THREADS, _COHORTS
#   tools/host_sweep.sh "1 2 4" "0 2 4" ...
for spec in "$@"; do
  set -- $spec
  COEVO_HOST_PREQUEUE=$1 COEVO_HOST_THREADS=$2 COEVO_HOST_COHORTS=$3 timeout -k 10 120 python bench.py --env host --no-extra \
    --no-cpu-baseline 2>/dev/null > /tmp/hs.json
  python - "$spec" <<'PY'
import json, sys
j = json.loads(open("/tmp/hs.json").read().strip().splitlines()[-1])
print("prequeue threads cohorts =", sys.argv[1], "->", round(j["gens_per_sec"], 1),
      {k: round(v, 1) for k, v in j["host_env"]["per_cohort_cycle_us"].items()})
PY
done
