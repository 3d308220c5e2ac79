"""Kernel timeline of one graph-replayed bench step from a rocprofv3 --kernel-trace output directory:
   python tools/step_timeline.py <trace dir>       (start us, duration, gap to the previous kernel, name)"""
import csv, glob, sys

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "gate_logits_walk" in r["Kernel_Name"]]
pairs = [(a, b) for a, b in zip(idx, idx[1:]) if 900 < (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 < 1150]
a, b = pairs[len(pairs) // 2]
t0, prev = int(rows[a]["Start_Timestamp"]), None
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} gap {((s - prev) / 1e3 if prev else 0):5.1f}  {r['Kernel_Name'][:100]}")
    prev = e
