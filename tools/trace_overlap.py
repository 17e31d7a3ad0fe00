"""diagnostic: summarise a rocprofv3 kernel-trace CSV - per queue kernel counts and pairwise overlap of the policy kernels"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "fc_policy" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"],
                         "mfma" if "mfma" in r["Kernel_Name"] else "light", r.get("Stream_Id", "?")))
rows.sort()
print("kernels:", len(rows), "queues:", collections.Counter(r[2] for r in rows), "streams:", collections.Counter(r[4] for r in rows))
lo = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
t0 = rows[lo][0]
for s, e, q, kind, st in rows[lo:lo + 16]:
    print(f"  {kind:5s} q={q} stream={st}  {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us")
