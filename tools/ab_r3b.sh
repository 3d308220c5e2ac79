#!/bin/bash
# round 3: weight-gradient tile order (EVE_FFN_NFAST) A/B on the MoE bench step, kernel times from HIP events
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3_ab2
mkdir -p $out
cd $root
export EVE_MOE_LIB=$root/eve_amd/libeve_moe_dbg.so
for v in 0 1 0 1; do
  EVE_FFN_NFAST=$v timeout -k 10 200 python bench.py --allow-foreign-lib --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs > $out/bench_nf$v.json 2> $out/bench_nf$v.err || { tail -5 $out/bench_nf$v.err; exit 3; }
  python - <<PY
import json
l=json.load(open("$out/bench_nf$v.json"))
print("NFAST=$v", l["ms_per_step"], l["roofline"]["frac"], l["phases_ms_per_step"]["eve_ffn_bwd"], l["phases_ms_per_step"]["eve_ffn_fwd"])
PY
done
