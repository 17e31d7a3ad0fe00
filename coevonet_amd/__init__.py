"""MI355X-native accelerator for CoEvoNet's population-evaluation hot path."""
import os as _os
import sys as _sys

# GPU_MAX_HW_QUEUES: HIP maps streams onto this many hardware queues round robin (default 4).  With several engines in
# one process a later engine's cohort stream then shares the caller's queue and its launches serialise (cfg 5 after the
# headline engine: 8.3 instead of 10.1 generations/s), so the package asks for 8 - which only counts when the variable is
# set BEFORE the HIP runtime starts.  Under rocprofv3 the profiler's library starts the runtime first: export it in the
# shell (tools/profile_round.sh, tools/pmc_kernel.sh do).  bench.py records the value it ran with.
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _t = _sys.modules.get("torch")
    if _t is not None and _t.cuda.is_initialized():
        import warnings as _w
        _w.warn("coevonet_amd: the HIP runtime started before this import, so GPU_MAX_HW_QUEUES=8 cannot take effect; "
                "export it before the process starts (cohort streams otherwise share hardware queues)")
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"
# DEBUG_HIP_DYNAMIC_QUEUES (NOT set here).  With the runtime's default (static) mapping a stream keeps the hardware queue it was
# dealt when it was created; the first hipGraph launch of a process creates the graph executor's own parallel streams, and
# streams created after that share hardware queues - every operation on them queues behind its neighbours' (~9 us per
# operation: the host-cores env mode after the device-resident loop ran 233-303 instead of 440-455 generations/s; bisected to
# ONE replayed graph in round 4, mechanism found in round 5: profiles/r05_experiments.md section 1).  DEBUG_HIP_DYNAMIC_QUEUES=1
# (a stream takes a free hardware queue when it has work) removes the cliff - both orders then run at a fresh process's rate -
# but it is a debug switch of the runtime: a long process that builds and tears down many engines (the DeepQN GPU tests, 34
# in one process) died with a segmentation fault inside hipDeviceSynchronize under it, twice out of two, and never without.
# So: export it yourself for a process that runs the host-cores mode after a graph-replaying mode and is short-lived, or do
# what bench.py does - run the host-cores mode in a process of its own.
