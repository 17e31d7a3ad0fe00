#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/profile_round.sh) -> tracked summaries under profiles/:
  <tag>_<workload>_kernel_stats.csv   rocprofv3 --stats per kernel (names shortened), + the bench line of the same run
  <tag>_headline_overlap.json         per-stream begin/end of the dominant kernel: how much of the rollout has two cohort
                                      launches in flight (what reconciles the summed kernel time with ms_per_step)
  <tag>_pmc_hbm_traffic.json          FETCH_SIZE / WRITE_SIZE per kernel (gfx950 correction noted inside)
usage: python tools/summarise_profiles.py <tag>"""
import collections
import csv
import glob
import json
import os
import re
import sys

csv.field_size_limit(1 << 30)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"\(.*", "", name)                      # drop the argument list
    name = name.replace("void ", "").replace("coevo::", "")
    return name[:90]


def stats(tag, wl, src, dst):
    # gpurun merges every call's output into the same scratch tree: take the newest run of this workload, and of it the
    # process that launched the kernels (rocprofv3 also writes an empty set for helper processes)
    files = sorted(glob.glob(f"{src}/{wl}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
    files = [f for f in files if os.path.getmtime(f) > os.path.getmtime(files[-1]) - 120 and os.path.getsize(f) > 200] if files else []
    if not files:
        return None
    rows = list(csv.DictReader(open(max(files, key=os.path.getsize))))
    out = f"{dst}/{tag}_{wl}_kernel_stats.csv"
    with open(out, "w") as f:
        bench = open(f"{src}/{wl}.bench.json").read().strip() if os.path.exists(f"{src}/{wl}.bench.json") else ""
        f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extra --no-cpu-baseline ... ({wl}); bench line of this profiled run:\n")
        if bench:
            b = json.loads(bench)
            keep = {k: b[k] for k in ("metric", "value", "unit", "gens_per_sec", "ms_per_step", "steps", "warmup") if k in b}
            keep["roofline"] = {k: v for k, v in b.get("roofline", {}).items()
                                if k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches_timed")}
            f.write("# " + json.dumps(keep) + "\n")
        wr = csv.writer(f)
        wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:14]:
            wr.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], f'{float(r["AverageNs"]):.0f}', r["Percentage"],
                         r["MinNs"], r["MaxNs"]])
    return out


def overlap(tag, src, dst, kernel="fc_cycle16_kernel"):
    files = sorted(glob.glob(f"{src}/headline/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
    files = [max([f for f in files if os.path.getmtime(f) > os.path.getmtime(files[-1]) - 120], key=os.path.getsize)] if files else []
    if not files:
        return None
    iv = []
    for r in csv.DictReader(open(files[0])):
        if kernel in r["Kernel_Name"]:
            iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    iv.sort()
    if not iv:
        return None
    # the rollouts are bursts of launches separated by the selection / breeding tail: analyse the second half (timed region)
    iv = iv[len(iv) // 2:]
    ev = []
    for s, e, _, _ in iv:
        ev += [(s, 1), (e, -1)]
    ev.sort()
    t_prev, depth, busy = ev[0][0], 0, collections.Counter()
    for t, d in ev:
        busy[depth] += t - t_prev
        depth += d
        t_prev = t
    span = iv[-1][1] - iv[0][0]
    per_stream = collections.Counter((q, st) for _, _, q, st in iv)
    durs = [e - s for s, e, _, _ in iv]
    # gaps between consecutive launches of one stream (dependent kernel boundaries)
    gaps = []
    last = {}
    for s, e, q, st in iv:
        if (q, st) in last and 0 <= s - last[(q, st)] < 50000:
            gaps.append(s - last[(q, st)])
        last[(q, st)] = e
    out = {"source": f"rocprofv3 --kernel-trace of the headline bench (second half of the {kernel} launches)",
           "kernel": kernel, "launches": len(iv), "launches_per_queue_stream": {f"{k[0]}/{k[1]}": v for k, v in per_stream.items()},
           "mean_duration_us": sum(durs) / len(durs) / 1e3,
           "sum_of_durations_us": sum(durs) / 1e3, "union_of_intervals_us": (busy[1] + sum(v for k, v in busy.items() if k >= 2)) / 1e3,
           "time_with_two_or_more_in_flight_us": sum(v for k, v in busy.items() if k >= 2) / 1e3,
           "time_with_exactly_one_us": busy[1] / 1e3, "time_with_none_us": busy[0] / 1e3, "span_us": span / 1e3,
           "fraction_of_kernel_busy_time_with_two_in_flight":
               sum(v for k, v in busy.items() if k >= 2) / max(1, busy[1] + sum(v for k, v in busy.items() if k >= 2)),
           "mean_same_stream_gap_us": (sum(gaps) / len(gaps) / 1e3) if gaps else None,
           "note": "sum_of_durations exceeds the span because two cohort launches run side by side; time_with_none is the "
                   "selection / breeding / reset tail between rollouts plus the dependent-launch gaps"}
    path = f"{dst}/{tag}_headline_overlap.json"
    json.dump(out, open(path, "w"), indent=1)
    return path


def pmc(tag, src, dst):
    def load(d, counter):
        per = collections.defaultdict(list)
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter:
                    per[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        return per
    fetch, write = load(f"{src}/pmc_FETCH_SIZE", "FETCH_SIZE"), load(f"{src}/pmc_WRITE_SIZE", "WRITE_SIZE")
    if not fetch:
        return None
    out = {"source": "rocprofv3 --kernel-trace --pmc <counter> (one counter per pass) -- python3 bench.py --no-extra "
                     f"--no-cpu-baseline --steps 5 --warmup 2, MI355X, {tag}",
           "correction": "FETCH_SIZE is in KiB and on gfx950 counts half of a wide coalesced 16 B/lane stream "
                         "(MI355X_MICROARCH.md, HBM): bytes = 2*1024*FETCH_SIZE; WRITE_SIZE*1024 is exact", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith(("fc_", "mpe_", "ga_", "dist_", "es_", "dqn_", "synth_", "counter", "stamps", "centered")):
            continue
        e = {}
        if k in fetch:
            e["FETCH_SIZE_KiB_mean"] = sum(fetch[k]) / len(fetch[k]); e["FETCH_SIZE_launches"] = len(fetch[k])
        if k in write:
            e["WRITE_SIZE_KiB_mean"] = sum(write[k]) / len(write[k]); e["WRITE_SIZE_launches"] = len(write[k])
        out["kernels"][k] = e
    dom = max(out["kernels"], key=lambda k: sum(fetch.get(k, [0])))
    e = out["kernels"][dom]
    out["dominant_kernel"] = dom
    out["dominant_kernel_hbm_bytes_per_launch"] = 2 * 1024 * e["FETCH_SIZE_KiB_mean"] + 1024 * e.get("WRITE_SIZE_KiB_mean", 0.0)
    path = f"{dst}/{tag}_pmc_hbm_traffic.json"
    json.dump(out, open(path, "w"), indent=1)
    return path


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
    src, dst = f"{REPO}/gpurun_out/prof_{tag}", f"{REPO}/profiles"
    for wl in ("headline", "cfg2_T200", "cfg2_host_env", "cfg3_es", "cfg3_es_ext", "cfg4_dqn_ga", "cfg5_dqn_es",
               "cfg4_dqn_ga_c6", "cfg5_dqn_es_c6", "cfg2_shard_1_of_4", "cfg2_shard_1_of_8", "cfg4_host_frames", "cfg3_host_env"):
        print(stats(tag, wl, src, dst))
    print(overlap(tag, src, dst))
    if glob.glob(f"{src}/pmc_*_FETCH_SIZE"):   # per-workload passes (round 3 on): tools/pmc_traffic_all.py
        import subprocess
        subprocess.check_call([sys.executable, f"{REPO}/tools/pmc_traffic_all.py", src, f"{dst}/{tag}_pmc_hbm_traffic.json"])
    else:
        print(pmc(tag, src, dst))
