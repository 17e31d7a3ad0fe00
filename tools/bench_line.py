#!/usr/bin/env python3
"""reads one bench.py JSON line on stdin, prints the few numbers an A/B run needs"""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads([ln for ln in sys.stdin.read().splitlines() if ln.startswith('{')][-1])
r = d.get("roofline", {})
agg = (r.get("rollout_aggregate") or {}).get("frac")
print(tag, f"gens/s {d['gens_per_sec']:.2f}  ms/step {d['ms_per_step']:.3f}  launch_ms {r.get('avg_launch_ms')}  frac {r.get('frac')}  aggregate {agg}")
