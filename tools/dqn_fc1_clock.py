#!/usr/bin/env python3
"""The clock the chip holds during the DeepQN fc1 launch: shader-clock (s_memtime) against 100 MHz (s_memrealtime) stamps of one
wave per task at the launch's start and end (diagnostic build: tools/build_variant.sh stamps deepqn.hip -DCOEVO_PHASE_STAMPS).
    COEVO_LIB=variants/libcoevo_stamps.so python tools/dqn_fc1_clock.py [--shape cfg4|cfg5|pop:NxR]"""
import os
import subprocess
import sys

os.environ.setdefault("COEVO_ALLOW_VARIANT", "1")
import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
# run the shapes tool in-process (it leaves its launches behind), then read the stamps of the LAST fc1 launch
sys.argv = [os.path.join(root, "tools", "bench_dqn_shapes.py")] + sys.argv[1:]
exec(compile(open(sys.argv[0]).read(), sys.argv[0], "exec"))
buf = (L.C.c_ulonglong * (1024 * 16))()
lib.coevo_debug_read_dqn_stamps.argtypes = [L.C.c_void_p, L.C.c_int]
assert lib.coevo_debug_read_dqn_stamps(buf, 1024 * 16) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.int64)[:len(layout)]
real = (st[:, 12] - st[:, 10]) / 100.0          # us
shader = (st[:, 13] - st[:, 11]).astype(np.float64)
ok = real > 1
print(f"fc1 waves (one per task, {ok.sum()} stamped): life {real[ok].mean():.1f} us (min {real[ok].min():.1f}, max {real[ok].max():.1f}); "
      f"shader clock {np.mean(shader[ok] / real[ok]) / 1e3:.2f} GHz (min {np.min(shader[ok] / real[ok]) / 1e3:.2f}, "
      f"max {np.max(shader[ok] / real[ok]) / 1e3:.2f}); launch span {(st[ok, 12].max() - st[ok, 10].min()) / 100.0:.1f} us")
