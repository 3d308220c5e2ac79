#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line, the rocprofv3 kernel trace of the same command and
# three separate --pmc passes (SQ/GRBM, FETCH_SIZE, WRITE_SIZE -- the TCC block cannot hold both sizes at once).
# The trace and counter passes leave out the other_configs / encoder12 / cpu_baseline legs of the line (other shapes
# of the same kernels would mix into the per-kernel averages).
# Everything lands under gpurun_out/<tag>/; tools/kernel_roofline.py turns it into the profiles/ summaries.
set -o pipefail
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python3 "$root/bench.py" --steps 50 --warmup 10 > "$out/bench_line.json" 2> "$out/bench.err" || exit 1
echo "bench done" && tail -c 300 "$out/bench_line.json"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > "$out/bench_line_under_rocprof.json" 2> "$out/trace.err" || exit 2
echo "trace done"
# counters: eager issue (PMC collection serialises kernels anyway), few steps, no CPU baseline
common="$root/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-other-configs"
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_sq" -- python3 $common > "$out/pmc_sq.log" 2>&1 || exit 3
echo "pmc sq done"
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 $common > "$out/pmc_fetch.log" 2>&1 || exit 4
echo "pmc fetch done"
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 $common > "$out/pmc_write.log" 2>&1 || exit 5
echo "pmc write done"
