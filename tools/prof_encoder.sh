#!/bin/bash
# kernel trace of the 12-layer encoder step (tools/bench_encoder.py)
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r3_enc}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/tools/bench_encoder.py --steps 6 --warmup 2 > $out/enc.log 2>&1 || { tail -5 $out/enc.log; exit 1; }
tail -1 $out/enc.log
python3 $root/tools/kstats.py $out/trace | head -60
