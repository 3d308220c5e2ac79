#!/bin/bash
# round 3: kernel trace of the modality-split FFN step (tools/bench_split_ffn.py shape) and the 12-layer encoder step
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3_split
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/tools/bench_split_ffn.py --ours-only > $out/split.log 2>&1 || { tail -5 $out/split.log; exit 1; }
tail -2 $out/split.log
python3 $root/tools/kstats.py $out/trace | head -30
timeout -k 10 300 python3 $root/tools/bench_encoder.py --steps 8 --warmup 3 2>&1 | tail -5
