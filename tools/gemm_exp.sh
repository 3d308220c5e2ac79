#!/bin/bash
# GPU-box helper: per-kernel times of the expert-FFN launches (tools/ffn_only.py at the cfg-2 shape) for a list of
# debug-knob settings, each in its own process under rocprofv3 --kernel-trace --stats.
#   tools/gemm_exp.sh tag "ENV1=a ENV2=b" "ENV1=c" ...      ("-" = no knob)
# Uses eve_amd/libeve_moe_dbg.so (python eve_amd/build.py --debug-knobs).
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export EVE_MOE_LIB=${EVE_MOE_LIB:-$root/eve_amd/libeve_moe_dbg.so}
i=0
for v in "$@"; do
  i=$((i+1))
  d=$out/v$i
  rm -rf "$d"
  echo "== variant $i: $v" | tee -a "$out/summary.txt"
  if [ "$v" = "-" ]; then v=""; fi
  env $v true || exit 9
  ( [ -n "$v" ] && export $v; timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 "$root/tools/ffn_only.py" --iters ${ITERS:-12} ${FFN_ARGS:-} > "$d.log" 2>&1 ) || { tail -n 5 "$d.log"; exit 2; }
  tail -n 1 "$d.log" | tee -a "$out/summary.txt"
  python3 "$root/tools/kstats.py" "$d" | grep -E "gemm|reduce" | tee -a "$out/summary.txt"
done
