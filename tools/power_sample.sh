#!/bin/bash
# GPU-box helper: power / clock readings (rocm-smi, read-only) while the bench step replays, to put numbers on
# "the GEMM launches are power-limited" (DESIGN.md).   tools/power_sample.sh  ->  gpurun_out/power_sample.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/power_sample.txt
: > "$out"
echo "== idle" >> "$out"
timeout -k 5 20 rocm-smi --showpower --showclocks --showmaxpower >> "$out" 2>&1
timeout -k 10 120 python3 "$root/bench.py" --steps 6000 --warmup 20 --no-other-configs --no-cpu-baseline > "$root/gpurun_out/power_bench.json" 2>/dev/null &
pid=$!
sleep 9
for i in 1 2 3; do
  echo "== under load, sample $i" >> "$out"
  timeout -k 5 20 rocm-smi --showpower --showclocks >> "$out" 2>&1
  sleep 1
done
wait $pid
echo "bench rc=$?" >> "$out"
grep -E "Power|sclk|mclk|Max" "$out" | head -40
