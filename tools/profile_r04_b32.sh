#!/bin/bash
# Round-4 batch profile on the GPU box: per-kernel times of the lock-step 32-row decode (tools/batch_probe.py 32).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b32 -- python3 $R/tools/batch_probe.py 32 > $O/b32.log 2>&1 || exit 1
cd $R
python3 tools/summarize_rocprof.py stats $O/b32 $O/b32_kernel_stats.md || exit 2
cat $O/b32.log | tail -3
