#!/bin/bash
# kernel durations of the gate / permute kernels alone on the GPU: tools/side_bench.py under rocprofv3 --kernel-trace
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-side}
shift
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/tools/side_bench.py "$@" > $out/side.log 2>&1 || { tail -5 $out/side.log; exit 1; }
python3 $root/tools/kstats.py $out/trace | grep -v "at::native\|rocclr" | head -40
