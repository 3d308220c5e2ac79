#!/bin/bash
# GPU-box helper: per-kernel times inside the bench step (graph replay) for a list of library builds, each in its own
# process under rocprofv3 --kernel-trace --stats.   tools/ab_kernel.sh tag 'route_wgrad|final_sums' main NAME main NAME
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; pat=$2; shift 2
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  d=$out/v${i}_$v
  rm -rf "$d"
  if [ "$v" = "main" ]; then unset EVE_MOE_LIB; extra=; else export EVE_MOE_LIB=$root/eve_amd/libeve_moe_$v.so; extra=--allow-foreign-lib; [ -f "$EVE_MOE_LIB" ] || { echo "no $EVE_MOE_LIB"; exit 9; }; fi
  echo "== $i: $v" | tee -a "$out/summary.txt"
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 "$root/bench.py" $extra --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > "$d.json" 2> "$d.err" || { tail -n 5 "$d.err"; exit 2; }
  python3 -c "import json; l=json.load(open('$d.json')); print('step', l['ms_per_step'], 'frac', l['roofline']['frac'])" | tee -a "$out/summary.txt"
  python3 "$root/tools/kstats.py" "$d" | grep -E "$pat" | tee -a "$out/summary.txt"
done
