#!/usr/bin/env python3
"""bench.py output (stdin, or the file named by the last argument if it exists) -> one short line; other argv = label"""
import json
import os
import sys

src = open(sys.argv.pop()) if len(sys.argv) > 1 and os.path.isfile(sys.argv[-1]) else sys.stdin
j = json.loads(src.read().strip().splitlines()[-1])
r = j.get("roofline", {})
print(" ".join(sys.argv[1:]), f"gens/s {j['gens_per_sec']:.2f} ms {j['ms_per_step']:.2f}", r.get("kernel", "")[:18], f"{1e3 * r.get('avg_launch_ms', 0):.1f} us frac {r.get('frac', 0):.3f}")
