#!/bin/bash
# one rank of an N-GPU split (bench.py --shard-of N) under library variants:  tools/shard_variant_sweep.sh "" variants/libcoevo_X.so ...
for V in "$@"; do for N in 4 8; do
  COEVO_ALLOW_VARIANT=1 COEVO_LIB=$V python bench.py --shard-of $N --no-extra --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null > /tmp/sv.json
  python - "$V" $N <<'PY'
import json, sys
j = json.loads(open("/tmp/sv.json").read().strip().splitlines()[-1])
print(sys.argv[1] or "shipped", "shard_of", sys.argv[2], round(j["gens_per_sec"], 1), "launch_ms", j["roofline"]["avg_launch_ms"])
PY
done; done
