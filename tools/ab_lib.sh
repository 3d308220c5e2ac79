#!/bin/bash
# GPU-box helper: per-kernel times of the expert-FFN launches (tools/ffn_only.py, cfg-2 shape by default) for a list
# of library builds, each in its own process under rocprofv3 --kernel-trace --stats.
#   tools/ab_lib.sh tag main base main base     ("main" = eve_amd/libeve_moe.so, NAME = eve_amd/libeve_moe_NAME.so,
#                                                built with `python eve_amd/build.py --variant NAME [--define X=1]`)
# FFN_ARGS="--G 2 --E 2 ..." selects another shape; ITERS the launches per process.
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  d=$out/v${i}_$v
  rm -rf "$d"
  if [ "$v" = "main" ]; then lib=$root/eve_amd/libeve_moe.so; else lib=$root/eve_amd/libeve_moe_$v.so; fi
  [ -f "$lib" ] || { echo "no $lib"; exit 9; }
  echo "== $i: $v" | tee -a "$out/summary.txt"
  ( export EVE_MOE_LIB=$lib; timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 "$root/tools/ffn_only.py" --iters ${ITERS:-30} ${FFN_ARGS:-} > "$d.log" 2>&1 ) || { tail -n 5 "$d.log"; exit 2; }
  tail -n 1 "$d.log" | tee -a "$out/summary.txt"
  python3 "$root/tools/kstats.py" "$d" | grep -E "gemm|reduce|bias_grads" | tee -a "$out/summary.txt"
done
