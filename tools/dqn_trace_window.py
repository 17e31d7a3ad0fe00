#!/usr/bin/env python3
"""prints a window of a rocprofv3 --kernel-trace csv (kernel, stream/queue, start, duration) to see how the DeepQN
launches of two cohorts interleave.  usage: python tools/dqn_trace_window.py <dir> [first_row] [rows]"""
import csv, glob, sys
csv.field_size_limit(1 << 30)
d = sys.argv[1]; first = int(sys.argv[2]) if len(sys.argv) > 2 else 2000; n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-28:], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0 = rows[first][0]
for s, e, k, q, st in rows[first:first + n]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  q{q}/s{st}  {k}")
