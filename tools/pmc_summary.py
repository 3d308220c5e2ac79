"""Summarise a rocprofv3 --pmc counter_collection.csv per GEMM kernel instantiation."""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r"grouped_gemm_(?:mfma|ppp|pp_wgrad|pp)<([^>]*)>", n)
    if not m and len(sys.argv) < 3: continue
    key = "gemm<" + m.group(1) + ">" if m else re.sub(r"\(.*", "", n)[:60]
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key].add(r["Dispatch_Id"])
for k, v in agg.items():
    n = len(cnt[k])
    print(k, "dispatches", n)
    for c, val in sorted(v.items()):
        print(f"   {c:30s} {val / n:16.0f}")
