"""Sum rocprofv3 --pmc counter_collection CSV rows per kernel: python tools/pmc_by_kernel.py <dir> -> table of
counter averages per launch (diagnostic; used for the L2 hit-rate figures in DESIGN.md)."""
import collections
import csv
import glob
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0][:70]
            rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r["Dispatch_Id"])
names = sorted({c for v in rows.values() for c in v})
print(f"{'kernel':70s} {'n':>5s} " + " ".join(f"{c:>16s}" for c in names) + "   hit%")
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1].values())):
    n = len(launches[k])
    hit = v.get("TCC_HIT_sum", 0.0)
    miss = v.get("TCC_MISS_sum", 0.0)
    rate = f"{100 * hit / (hit + miss):6.1f}" if hit + miss else "     -"
    print(f"{k:70s} {n:5d} " + " ".join(f"{v.get(c, 0.0) / n:16.0f}" for c in names) + "  " + rate)
