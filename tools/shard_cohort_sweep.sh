#!/bin/bash
# generations/s of one rank of an N-GPU strong-scaling split on one GPU (bench.py --shard-of N) for 1..4 game cohorts
for N in 2 4 8; do for K in 1 2 3 4; do
  python bench.py --shard-of $N --cohorts $K --no-extra --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null > /tmp/sc.json
  python - $N $K <<'PY'
import json, sys
j = json.loads(open("/tmp/sc.json").read().strip().splitlines()[-1])
print("shard_of", sys.argv[1], "cohorts", sys.argv[2], round(j["gens_per_sec"], 1), j["roofline"]["avg_launch_ms"])
PY
done; done
