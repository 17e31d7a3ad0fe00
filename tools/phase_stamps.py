#!/usr/bin/env python3
"""Diagnostic: where does a workgroup of the streaming policy kernel spend its cycles?  Needs a build with
COEVO_EXTRA_FLAGS=-DCOEVO_PHASE_STAMPS (never the shipped one).  Prints mean cycles per phase for 1 WG/CU and for the
full 600-task launch."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_args
from coevonet_amd import lib as L
from coevonet_amd.game_logic import initialize_env
from coevonet_amd.genetic_algorithm import GATrainer

torch.manual_seed(0)
args = make_args(200, 5, 2, 200)
env = initialize_env(args)
tr = GATrainer(env, args, rng="device_philox", env_mode="device", collect=False)
tr.step()
ro, p = tr.eng.ro, tr.eng.plan
dll = L.load()
names = ["entry->obs staged", "fc1+LN1 (to h1 in LDS)", "fc2 stream", "LN2", "output chain", "argmax/store"]
print("=== MFMA (shared-opponent) kernel, fused mode not used here: state-read variant")
nh = len(p.heavy_np)
for rep in range(3):
    L.call("coevo_mpe_policy_cycle", L._p(ro.slab), L._p(p.heavy), nh, p.heavy_max, L._p(ro.state), p.n_games,
           L._p(p.row_game), L._p(p.row_slot), L._p(ro.actions), L._p(ro.status))
torch.cuda.synchronize()
buf = (C.c_ulonglong * (nh * 16))()
assert dll.coevo_debug_read_phase_stamps(buf, nh * 16) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(nh, 16).astype(np.int64)
d = np.diff(st[:, :7], axis=1)
for i, nm in enumerate(["entry->obs staged", "fc1 (MFMA) + LN1 + h1 image", "fc2 (MFMA) stream", "LN2", "output chain", "argmax/store"]):
    print(f"  {nm:28s} {d[:, i].mean():9.0f}  (min {d[:, i].min():7d} max {d[:, i].max():7d})")
print(f"  {'whole workgroup':28s} {(st[:, 6] - st[:, 0]).mean():9.0f}")
print("=== streaming (per-individual) kernel")
for n in (128, 256, 600):
    for rep in range(3):
        L.call("coevo_mpe_policy_cycle", L._p(ro.slab), L._p(p.light), n, p.light_max, L._p(ro.state), p.n_games,
               L._p(p.row_game), L._p(p.row_slot), L._p(ro.actions), L._p(ro.status))
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (n * 16))()
    assert dll.coevo_debug_read_phase_stamps(buf, n * 16) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(n, 16).astype(np.int64)
    d = np.diff(st[:, :7], axis=1)
    print(f"--- {n} workgroups: mean cycles per phase (100 MHz realtime? no: shader clock via s_memtime)")
    for i, nm in enumerate(names):
        print(f"  {nm:28s} {d[:, i].mean():9.0f}  (min {d[:, i].min():7d} max {d[:, i].max():7d})")
    tot = st[:, 6] - st[:, 0]
    print(f"  {'whole workgroup':28s} {tot.mean():9.0f}   launch span {st[:, 6].max() - st[:, 0].min()} cycles")
