#!/usr/bin/env python3
"""The DeepQN agent-step launches of the cfg 4 / cfg 5 shards (their exact task tables, random weights and frames) timed
kernel by kernel with HIP events: conv stack, fc1, and the whole three-launch step.

    [COEVO_LIB=variants/libcoevo_X.so] python tools/bench_dqn_shapes.py [--shape cfg4|cfg5|eval|pop:<nets>x<rows>] [--C 4] [--reps 40]
"""
import argparse
import os
import sys

os.environ.setdefault("COEVO_ALLOW_VARIANT", "1")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coevonet_amd import lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="cfg4")
ap.add_argument("--C", type=int, default=4)
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--interleave", action="store_true", help="cfg4: HoF nets spread among the population nets in the task table")
ap.add_argument("--fc1", default="streamed", choices=("streamed", "tiled"), help="fc1 layout of the slab (COEVO_DQN_FC1_TILED)")
ap.add_argument("--task-rows", type=int, default=L.DQN_MAX_ROWS, help="rows per task of a net that acts in many games")
a = ap.parse_args()
dev = "cuda"
lib = L.load()


def _cut(rows):
    return [min(a.task_rows, rows - i) for i in range(0, rows, a.task_rows)]


if a.shape == "cfg4":      # pop 50 x 10 HoF games, 10 HoF nets x 50 games (+ 10 evaluation games for the newest)
    n_act = 6
    if a.interleave:   # five pop nets, one HoF net, ...: every contiguous eighth of the table (= one XCD's share in the
        layout = []    # fc1 launch) then holds about the same number of distinct nets
        for j in range(10):
            layout += [(5 * j + i, 10) for i in range(5)]
            layout += [(50 + j, r) for r in (_cut(60) if j == 0 else _cut(50))]
    else:
        layout = [(i, 10) for i in range(50)]
        for j in range(10):
            layout += [(50 + j, r) for r in (_cut(60) if j == 0 else _cut(50))]
    n_nets = 60
elif a.shape.startswith("pop:"):   # pop:<nets>x<rows>: that many nets with that many frames each
    n_act = 6
    n_nets, r = (int(v) for v in a.shape[4:].split("x"))
    layout = [(i, rr) for i in range(n_nets) for rr in _cut(r)]
elif a.shape == "eval":    # the evaluation games of a Co-ES generation: one base net x 10 games
    n_act = 18
    layout = [(0, 10)]
    n_nets = 1
else:                      # one cohort of cfg 5: 125 perturbed nets x 1 game, the base net x 125 games
    n_act = 18
    layout = []
    for j in range(125):
        layout += [(1 + j, 1)]
    layout += [(0, r) for r in _cut(125)]
    n_nets = 126
stride = int(lib.coevo_dqn_slab_stride(a.C, n_act))
P = int(lib.coevo_dqn_param_count(a.C, n_act))
slab = (torch.randn(n_nets, stride, device=dev) * 0.02).contiguous()
tasks = np.zeros(len(layout), dtype=L.DQN_TASK_DTYPE)
row = 0
for i, (net, r) in enumerate(layout):
    tasks[i] = (net * stride, row, r)
    row += r
d_tasks = L.tasks_to_device(tasks, dev)
frames = torch.randint(0, 256, (row, 84, 84, a.C), dtype=torch.uint8, device=dev)
actions = torch.zeros(row, dtype=torch.int32, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
ws = torch.zeros(int(lib.coevo_dqn_workspace_bytes(row)) // 4, dtype=torch.float32, device=dev)
tc = [lib.coevo_rollout_ctx_create(a.reps + 8) for _ in range(2)]
stream = torch.cuda.current_stream().cuda_stream
c_arg = a.C | (L.DQN_FC1_TILED if a.fc1 == "tiled" else 0)


def run(which=None):
    L._check(lib.coevo_dqn_forward_argmax_timed(L._p(slab), L._p(d_tasks), len(layout), a.task_rows, row, c_arg, n_act, L._p(frames),
                                                L._p(actions), None, L._p(status), L._p(ws),
                                                tc[which] if which is not None else None, which or 0, stream), "fwd")


for _ in range(5):
    run()
torch.cuda.synchronize()
for rep in range(2 * a.reps):
    run(rep & 1)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
for s, e in ev:
    s.record(); run(); e.record()
torch.cuda.synchronize()
step = float(np.median([s.elapsed_time(e) for s, e in ev])) * 1e3


def times(c):
    buf = (L.C.c_float * 100000)()
    n = lib.coevo_rollout_ctx_light_times(c, buf, 100000)
    return np.array([buf[i] for i in range(n)]) * 1e3


conv, fc1 = times(tc[0]), times(tc[1])
mac = 3276800 + 2654208 + 1806336 + (a.C - 4) * 64 * 32 * 400
tf = row * 2 * mac / (conv.mean() * 1e-6) / 1e12
fc1_bytes = n_nets * 512 * 3136 * 4 + row * (3136 + 512) * 4
print(f"{a.shape} C={a.C} fc1={a.fc1} lib={os.path.basename(L.LIB_PATH)} flags='{(lib.coevo_build_flags() or b'').decode()}': "
      f"{row} rows, {len(layout)} tasks, {n_nets} nets | conv {conv.mean():.1f} us (min {conv.min():.1f}) = {tf:.1f} TF/s "
      f"= {tf / 157.3:.3f} | fc1 {fc1.mean():.1f} us (min {fc1.min():.1f}) = {fc1_bytes / fc1.mean() / 1e6:.2f} TB/s "
      f"= {fc1_bytes / fc1.mean() / 1e6 / 8:.3f} | step (3 launches) {step:.1f} us")
