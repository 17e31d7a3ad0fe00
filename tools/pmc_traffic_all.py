"""Aggregate the rocprofv3 --pmc passes of tools/profile_round.sh (one counter per pass and workload: directories
pmc_<workload>_FETCH_SIZE / pmc_<workload>_WRITE_SIZE) into profiles/<tag>_pmc_hbm_traffic.json, which bench.py reads for
`roofline.traffic`.

usage: python tools/pmc_traffic_all.py gpurun_out/prof_<tag> profiles/<tag>_pmc_hbm_traffic.json [--merge]
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced 16 B/lane stream
(MI355X_MICROARCH.md, HBM section): bytes = 2 * 1024 * FETCH_SIZE; WRITE_SIZE * 1024 is exact."""
import collections
import csv
import glob
import json
import os
import sys


def load(d, counter):
    per = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return per


def main():
    root, out_path = sys.argv[1], sys.argv[2]
    merge = len(sys.argv) > 3 and sys.argv[3] == "--merge"   # keep the workloads already in out_path that root does not hold
    old = json.load(open(out_path)) if merge and os.path.exists(out_path) else None
    out = {"source": "rocprofv3 --kernel-trace --pmc <counter> (one counter per pass) -- python3 bench.py --workload ... "
                     "--no-extra --no-cpu-baseline (tools/profile_round.sh), MI355X",
           "correction": "FETCH_SIZE is in KiB and on gfx950 counts half of a wide coalesced 16 B/lane stream "
                         "(MI355X_MICROARCH.md, HBM): bytes = 2*1024*FETCH_SIZE; WRITE_SIZE*1024 is exact",
           "workloads": {}}
    for d in sorted(glob.glob(os.path.join(root, "pmc_*_FETCH_SIZE"))):
        wl = os.path.basename(d)[len("pmc_"):-len("_FETCH_SIZE")]
        fetch, write = load(d, "FETCH_SIZE"), load(d.replace("_FETCH_SIZE", "_WRITE_SIZE"), "WRITE_SIZE")
        ks = {}
        for k in sorted(set(fetch) | set(write)):
            e = {}
            if k in fetch:
                e["FETCH_SIZE_KiB_mean"] = sum(fetch[k]) / len(fetch[k])
                e["launches"] = len(fetch[k])
            if k in write:
                e["WRITE_SIZE_KiB_mean"] = sum(write[k]) / len(write[k])
            e["hbm_bytes_per_launch"] = 2 * 1024 * e.get("FETCH_SIZE_KiB_mean", 0.0) + 1024 * e.get("WRITE_SIZE_KiB_mean", 0.0)
            if e["hbm_bytes_per_launch"] >= 1e6:   # (the bookkeeping kernels are noise here)
                ks[k] = e
        out["workloads"][wl] = {"kernels": ks}
        bj = os.path.join(root, f"pmc_{wl}_FETCH_SIZE.bench.json")
        if os.path.exists(bj) and os.path.getsize(bj):
            out["workloads"][wl]["bench_line_of_the_profiled_run"] = json.load(open(bj))
    if old:
        for wl, e in old.get("workloads", {}).items():
            out["workloads"].setdefault(wl, e)
    json.dump(out, open(out_path, "w"), indent=1)
    for wl, e in out["workloads"].items():
        for k, v in e["kernels"].items():
            print(f"{wl:18s} {k[:60]:60s} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch")


if __name__ == "__main__":
    main()
