#!/bin/bash
# round 3: compact tile enumeration + dynamic wgrad slices (split FFN), persistent wgrad (MoE)
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3_ab1
mkdir -p $out
cd $root
timeout -k 10 600 python -u -m pytest tests/test_multiway_gpu.py tests/test_permute_ffn_gpu.py tests/test_fused_gpu.py tests/test_layer_gpu.py -x -q --timeout 300 --timeout-method thread > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -3 $out/tests.log
export EVE_MOE_LIB=$root/eve_amd/libeve_moe_dbg.so
for c in 0 1; do
  echo "== split FFN, EVE_FFN_COMPACT=$c"
  EVE_FFN_COMPACT=$c timeout -k 10 200 python tools/bench_split_ffn.py 2>&1 | tee $out/split_compact$c.log || exit 2
done
for p in 0 1; do
  echo "== MoE bench, EVE_FFN_WGRAD_PERSIST=$p"
  EVE_FFN_WGRAD_PERSIST=$p timeout -k 10 200 python bench.py --allow-foreign-lib --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs > $out/bench_wp$p.json 2> $out/bench_wp$p.err || { tail -5 $out/bench_wp$p.err; exit 3; }
  python - <<PY
import json
l=json.load(open("$out/bench_wp$p.json"))
print(l["ms_per_step"], l["roofline"]["frac"], l["phases_ms_per_step"])
PY
done
