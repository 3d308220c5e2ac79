"""Print per-kernel average durations from a rocprofv3 --kernel-trace --stats CSV directory."""
import csv, glob, re, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    m = re.search(r"grouped_gemm_(?:mfma|ppp|pp_wgrad|pp)<([^>]*)>", n)
    name = "gemm<" + m.group(1) + ">" if m else re.sub(r"\(.*", "", n)[:70]
    print(f"{name:72s} calls={r['Calls']:>4} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
