"""Builds libcoevo.so (all HIP kernels + the C ABI) for gfx950, in-tree, with hipcc.

    python -m coevonet_amd.build [--force]

Every source is compiled to its own object (in parallel, only when it or a header changed), then linked.  The .so is
git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libcoevo.so")
SOURCES = ["fc_forward.hip", "mpe_env.hip", "offspring.hip", "select.hip", "rollout_api.hip", "deepqn.hip",
           "dqn_engine.hip", "host_rollout.hip", "host_placement.hip"]
# -ffp-contract=off: only explicit fmaf fuses (the canonical arithmetic contract with the oracle)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall",
         "-Wno-unused-function"]


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return hs + [os.path.join(HERE, "..", "include", "coevo.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _extra_flags():
    return os.environ.get("COEVO_EXTRA_FLAGS", "").split()  # tuning experiments only, e.g. -DCOEVO_LIGHT_U=32


def _flag_signature():
    return " ".join(FLAGS + _extra_flags())


def needs_build():
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    flag_file = os.path.join(OBJ, "flags.txt")
    if not os.path.exists(flag_file) or open(flag_file).read() != _flag_signature():
        return True   # same sources, other switches: the library on disk is another variant
    return _stale(LIB, srcs + _headers())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = _extra_flags()
    if extra:   # reported by coevo_build_flags(); lib.load() refuses such a library outside tools/
        extra = extra + ['-DCOEVO_TU_FLAGS="' + " ".join(extra).replace('"', "'") + '"']
    os.makedirs(OBJ, exist_ok=True)
    flag_file = os.path.join(OBJ, "flags.txt")
    flag_sig = _flag_signature()
    if not os.path.exists(flag_file) or open(flag_file).read() != flag_sig:
        force = True
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + _headers()):
            jobs.append([hipcc] + FLAGS + extra + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(len(jobs), 6) or 1) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in srcs]
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    with open(flag_file, "w") as f:
        f.write(flag_sig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
