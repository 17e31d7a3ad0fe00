#!/usr/bin/env python3
"""K2 timing on a cfg4-like shard: n_nets DeepQN weight sets x rows frames each (synthetic uint8 frames)."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coevonet_amd import lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--nets", type=int, default=100)
ap.add_argument("--rows", type=int, default=10)
ap.add_argument("--C", type=int, default=4)
ap.add_argument("--actions", type=int, default=6)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
dev = "cuda"
P = int(L.load().coevo_dqn_param_count(a.C, a.actions)); stride = int(L.load().coevo_dqn_slab_stride(a.C, a.actions))
slab = (torch.randn(a.nets, stride, device=dev) * 0.02).contiguous()
rows = a.nets * a.rows
tasks = np.zeros(a.nets, dtype=L.DQN_TASK_DTYPE)
for i in range(a.nets):
    tasks[i] = (i * stride, i * a.rows, a.rows)
d_tasks = L.tasks_to_device(tasks, dev)
frames = torch.randint(0, 256, (rows, 84, 84, a.C), dtype=torch.uint8, device=dev)
actions = torch.zeros(rows, dtype=torch.int32, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
ws = torch.zeros(int(L.load().coevo_dqn_workspace_bytes(rows)) // 4, dtype=torch.float32, device=dev)

def run():
    L.call("coevo_dqn_forward_argmax", L._p(slab), L._p(d_tasks), a.nets, a.rows, rows, a.C, a.actions, L._p(frames),
           L._p(actions), None, L._p(status), L._p(ws))
for _ in range(2): run()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
for s, e in ev:
    s.record(); run(); e.record()
torch.cuda.synchronize()
ms = sorted(s.elapsed_time(e) for s, e in ev)[len(ev) // 2]
alg = a.nets * P * 4 + rows * 84 * 84 * a.C
flops = rows * 2 * (a.C * 64 * 32 * 400 + 512 * 64 * 81 + 576 * 64 * 49 + 3136 * 512 + 512 * a.actions)
print(f"DeepQN step: {a.nets} nets x {a.rows} frames: {ms:.3f} ms  -> {rows / ms * 1e3:.0f} frames/s, "
      f"{alg / ms / 1e6:.0f} GB/s algorithmic ({alg / 1e6:.0f} MB), {flops / ms / 1e9:.1f} TFLOP/s fp32")

# diagnostic build (COEVO_EXTRA_FLAGS=-DCOEVO_PHASE_STAMPS): where a frame's workgroup spends its time
import ctypes as C
dll = L.load()
if hasattr(dll, "coevo_debug_read_dqn_stamps"):
    n = min(rows, 2048)
    buf = (C.c_ulonglong * (n * 16))()
    dll.coevo_debug_read_dqn_stamps.argtypes = [C.c_void_p, C.c_int]
    assert dll.coevo_debug_read_dqn_stamps(buf, n * 16) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(n, 16).astype(np.int64)
    names = ["stage frame", "conv1", "bn1", "conv2", "bn2", "conv3", "bn3", "store act"]
    d = (st[:, 1:9] - st[:, 0:8]) / 100.0
    for i, nm in enumerate(names):
        print(f"  {nm:12s} mean {d[:, i].mean():7.2f} us  min {d[:, i].min():7.2f}  max {d[:, i].max():7.2f}")
    tot = (st[:, 8] - st[:, 0]) / 100.0
    print(f"  workgroup total mean {tot.mean():.2f} us; launch span {(st[:, 8].max() - st[:, 0].min()) / 100.0:.1f} us; "
          f"starts spread {(st[:, 0].max() - st[:, 0].min()) / 100.0:.1f} us")
