#!/usr/bin/env python3
"""cProfile of bench.py --env host (the host-cores env mode): where a generation's host time goes.
    python tools/host_env_profile.py [steps]"""
import cProfile
import os
import pstats
import runpy
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
steps = sys.argv[1] if len(sys.argv) > 1 else "4"
sys.argv = ["bench.py", "--env", "host", "--steps", steps, "--warmup", "1", "--no-extra", "--no-cpu-baseline"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
