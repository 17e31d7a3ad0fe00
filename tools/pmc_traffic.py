"""Aggregate rocprofv3 --pmc passes (one counter per pass) into profiles/r01_pmc_hbm_traffic.json.

usage: python tools/pmc_traffic.py <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass> <out.json>
Each dir holds rocprofv3's *_counter_collection.csv.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
half of a wide coalesced 16 B/lane stream (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact."""
import csv, glob, json, sys, collections


def load(d, counter):
    per = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return per


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --kernel-trace --pmc <counter> (one counter per pass) -- python3 bench.py --steps 5 "
                     "--warmup 2 --no-cpu-baseline, MI355X, round 1",
           "correction": "FETCH_SIZE is in KiB and on gfx950 counts half of a wide coalesced 16 B/lane stream "
                         "(MI355X_MICROARCH.md, HBM): bytes = 2*1024*FETCH_SIZE; WRITE_SIZE*1024 is exact",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        e = {}
        if k in fetch:
            e["FETCH_SIZE_KiB_mean"] = sum(fetch[k]) / len(fetch[k]); e["FETCH_SIZE_launches"] = len(fetch[k])
        if k in write:
            e["WRITE_SIZE_KiB_mean"] = sum(write[k]) / len(write[k]); e["WRITE_SIZE_launches"] = len(write[k])
        out["kernels"][k] = e
    dom = max(fetch, key=lambda k: sum(fetch[k]))
    out["dominant_kernel"] = dom
    e = out["kernels"][dom]
    out["dominant_kernel_hbm_bytes_per_launch"] = 2 * 1024 * e["FETCH_SIZE_KiB_mean"] + 1024 * e.get("WRITE_SIZE_KiB_mean", 0.0)
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(dom, out["dominant_kernel_hbm_bytes_per_launch"])


if __name__ == "__main__":
    main()
