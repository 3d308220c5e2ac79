"""HBM-side traffic of the grouped-GEMM launches from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python tools/gemm_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> > profiles/rNN_gemm_traffic.json

Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: both counters are in KiB;
FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane streaming reads (all of these kernels' operand fetches are
`buffer_load_dwordx4 ... lds`) at 64 bytes and is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections, csv, glob, json, re, sys


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        m = re.search(r"grouped_gemm_(?:mfma|ppp|pp_wgrad|pp)<([^>]*)>", r["Kernel_Name"])
        if not m or r["Counter_Name"] != counter:
            continue
        tot[m.group(1)] += float(r["Counter_Value"])
        n[m.group(1)].add(r["Dispatch_Id"])
    return {k: tot[k] / len(n[k]) for k in tot}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
kern = {}
for k in sorted(fetch):
    kern[k] = {"FETCH_SIZE_KiB": round(fetch[k]), "WRITE_SIZE_KiB": round(write.get(k, 0.0)),
               "hbm_bytes": int((2 * fetch[k] + write.get(k, 0.0)) * 1024)}
avg = sum(v["hbm_bytes"] for v in kern.values()) / len(kern)
print(json.dumps({"note": "per launch; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction); one "
                          "launch of each grouped-GEMM kernel per step, the weight-gradient launch computes both dW GEMMs",
                  "kernels": kern, "avg_hbm_bytes_per_launch": int(avg)}, indent=1))
