"""Per-kernel roofline table of one bench step from the rocprofv3 outputs that tools/collect_profiles.sh leaves.

    python tools/kernel_roofline.py gpurun_out/r01 > profiles/r01_kernel_roofline.txt
    (also writes profiles/<tag>_gemm_traffic.json next to it when --traffic-json PATH is given)

time      : average kernel duration from the --kernel-trace --stats pass (graph replay + instrumented eager steps)
HBM bytes : (2 * FETCH_SIZE + WRITE_SIZE) * 1024 from the two --pmc passes, as /opt/skills/guides/MI355X_MICROARCH.md
            prescribes for gfx950 (both counters in KiB; FETCH_SIZE counts the 128-byte requests of 16-byte-per-lane
            streaming reads at 64 bytes and is doubled; WRITE_SIZE is exact)
algorithmic bytes / flops: SURVEY.md 8(d) per-unit figures x the units of the launch (S tokens, K kept assignments)
MFMAbusy  : SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8): busy share of the matrix pipes at the
            clock the chip held during the counter pass (kernels run one at a time there, so it clocks higher than in the
            timed run; the flop-based fraction is against 2.5 PFLOP/s, i.e. 2.4 GHz)
parked%   : SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked on s_waitcnt / barriers); stall% : SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
"""
import collections, csv, glob, json, re, sys

d = sys.argv[1]
tj = sys.argv[sys.argv.index("--traffic-json") + 1] if "--traffic-json" in sys.argv else None
line = json.loads(open(d + "/bench_line.json").read().strip().splitlines()[-1])
S = line["config"]["tokens_per_gpu"]
K = line["config"]["kept_assignments"]
M, I, E, b = 768, 3072, 32, 2
HBM_PEAK, MFMA_PEAK = 8000.0, 2500.0  # GB/s, TFLOP/s


def short(n):
    m = re.search(r"(grouped_gemm_\w+<[^>]*>)", n)
    if m:
        return m.group(1)
    m = re.search(r"_ZN3eve(\d+)(\w+)", n)
    if m:
        return m.group(2)[: int(m.group(1))]
    n = re.sub(r"^void ", "", n)
    return re.sub(r"[<(].*", "", n).replace("eve::", "")[:48]


def stats(sub):
    f = glob.glob(f"{d}/{sub}/**/*kernel_stats.csv", recursive=True)[0]
    out = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        calls, tot = out.get(k, (0, 0.0))
        out[k] = (calls + int(r["Calls"]), tot + float(r["TotalDurationNs"]))
    return {k: (c, t / c / 1e3) for k, (c, t) in out.items()}


def pmc(sub):
    fs = glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True)
    tot, n = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(set)
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
    return {k: {c: v / len(n[k]) for c, v in cs.items()} for k, cs in tot.items()}


st, sq, fe, wr = stats("trace"), pmc("pmc_sq"), pmc("pmc_fetch"), pmc("pmc_write")
gemm_flops = 2.0 * M * I * K
alg = {  # kernel -> (algorithmic bytes, flops, note)
    "gate_logits_mfma_kernel": (S * M * b + 4 * S * E, 2.0 * S * M * E, "S*M*b + 4*S*E"),
    "gate_logits_walk_kernel": (S * M * b + 4 * S * E, 2.0 * S * M * E, "S*M*b + 4*S*E"),
    "gate_select32_kernel": (12 * E * S, 0, "logits + draws in, gates out: 12E B/token"),
    "gate_logits_select_kernel": (S * M * b + (4 * E + 24) * S, 2.0 * S * M * E, "rows in, gates + choices out (logits and noise never travel)"),
    "gate_assign32_kernel": (40 * S, 0, "idx/rank in, loc/slot/weight + slot maps out"),
    "dispatch_fwd_kernel": ((S + K) * M * b, 0, "(S+K)*M*b"),
    "dispatch_fwd_tok_kernel": ((S + K) * M * b, 0, "(S+K)*M*b"),
    "combine_fwd_np_kernel": ((S + K) * M * b, 0, "(K+S)*M*b"),
    "combine_bwd_np_kernel": ((S + 2 * K) * M * b, 0, "(S+2K)*M*b"),
    "route_bwd_bf16_kernel": ((S + K) * M * b, 2.0 * S * M * E, "(K+S)*M*b"),
    "combine_fwd_kernel": ((S + K) * M * b, 0, "(K+S)*M*b"),
    "combine_bwd_kernel": ((S + 2 * K) * M * b, 0, "(S+2K)*M*b"),
    "route_bwd_kernel": ((S + K) * M * b, 2.0 * S * M * E, "(K+S)*M*b"),
    "gate_wgrad_bf16_kernel": (S * M * b + 4 * S * E, 2.0 * S * M * E, "S*M*b + 4*S*E"),
    "gate_bwd32_kernel": (12 * E * S, 0, "gates, dlogits: ~12E B/token"),
    "bias_grads_kernel": (K * M * b, 0, "db2: dy rows in, K*M*b (+ the db1 partials, small)"),
    "route_wgrad_bf16_kernel": ((2 * S + K) * M * b + 4 * S * E, 4.0 * S * M * E,
                                "route_bwd (K+S)*M*b + gate_wgrad S*M*b + 4*S*E: the fused backward's combined launch"),
}
print(f"workload: {line['config']['workload']}")
print(f"S = {S} tokens, K = {K} kept assignments, bench line: {line['ms_per_step']} ms/step, {line['value']:.0f} tokens/s")
print()
hdr = f"{'kernel':58s} {'us':>7s} {'alg MB':>8s} {'alg GB/s':>9s} {'%HBM':>6s} {'HBM MB':>8s} {'HBM GB/s':>9s} {'TFLOP/s':>8s} {'%2.5PF':>7s} {'MFMAbusy':>8s} {'parked%':>7s} {'stall%':>6s}"
print(hdr)
traffic = {}
for k, (calls, us) in st.items():
    if not (k.startswith("grouped_gemm") or k in alg or "kernel" in k and ("eve" in k or k in sq)):
        if k not in alg:
            continue
    a_bytes, flops, _ = alg.get(k, (None, 0, ""))
    if k.startswith("grouped_gemm"):
        flops = 2 * gemm_flops if "wgrad" in k else gemm_flops
    hbm = None
    if k in fe and k in wr:
        hbm = (2 * fe[k].get("FETCH_SIZE", 0.0) + wr[k].get("WRITE_SIZE", 0.0)) * 1024
        if k.startswith("grouped_gemm"):
            traffic[k] = {"FETCH_SIZE_KiB": round(fe[k]["FETCH_SIZE"]), "WRITE_SIZE_KiB": round(wr[k]["WRITE_SIZE"]),
                          "hbm_bytes": int(hbm), "avg_us": round(us, 1)}
    busy = parked = stall = None
    c = sq.get(k)
    if c and c.get("GRBM_GUI_ACTIVE"):
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * 256 * cyc)
        if c.get("SQ_WAVE_CYCLES"):
            parked = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
            stall = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
    f = lambda v, w, p=1: (f"{v:{w}.{p}f}" if v is not None else " " * (w - 1) + "-")
    alg_gbs = a_bytes / us / 1e3 if a_bytes else None
    hbm_gbs = hbm / us / 1e3 if hbm else None
    tf = flops / us / 1e6 if flops and k.startswith("grouped_gemm") else None
    print(f"{k:58s} {us:7.1f} {f(a_bytes / 1e6 if a_bytes else None, 8)} {f(alg_gbs, 9, 0)} "
          f"{f(100 * alg_gbs / HBM_PEAK if alg_gbs else None, 6)} {f(hbm / 1e6 if hbm else None, 8)} {f(hbm_gbs, 9, 0)} "
          f"{f(tf, 8, 0)} {f(100 * tf / MFMA_PEAK if tf else None, 7)} {f(100 * busy if busy is not None else None, 8)} "
          f"{f(100 * parked if parked is not None else None, 7)} {f(100 * stall if stall is not None else None, 6)}")
if tj and traffic:
    launches = {k: (2 if "wgrad" in k else 1) for k in traffic}
    avg = sum(v["hbm_bytes"] for v in traffic.values()) / len(traffic)
    json.dump({"note": "per launch; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction); one launch "
                       "of each grouped-GEMM kernel per step, the weight-gradient launch computes both dW GEMMs",
               "source_hash": line["config"].get("library_source_hash"),
               "kernels": traffic, "avg_hbm_bytes_per_launch": int(avg)}, open(tj, "w"), indent=1)
