#!/usr/bin/env python3
"""stdin: bench.py output -> one short line (label from argv)"""
import json
import sys

j = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = j.get("roofline", {})
print(" ".join(sys.argv[1:]), f"gens/s {j['gens_per_sec']:.2f} ms {j['ms_per_step']:.2f}", r.get("kernel", "")[:18], f"{1e3 * r.get('avg_launch_ms', 0):.1f} us frac {r.get('frac', 0):.3f}")
