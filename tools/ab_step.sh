#!/bin/bash
# whole-step A/B: bench.py against the shipped library and against eve_amd/libeve_moe_prev.so, alternating
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/ab_step
mkdir -p $out
cd $root
for v in main prev main prev; do
  if [ $v = prev ]; then export EVE_MOE_LIB=$root/eve_amd/libeve_moe_prev.so; extra=--allow-foreign-lib; else unset EVE_MOE_LIB; extra=; fi
  timeout -k 10 200 python bench.py $extra --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs > $out/b_$v.json 2> $out/b_$v.err || { tail -5 $out/b_$v.err; exit 3; }
  python - <<PY
import json
l=json.load(open("$out/b_$v.json"))
print("$v", l["ms_per_step"], l["value"], l["roofline"]["frac"])
PY
done
