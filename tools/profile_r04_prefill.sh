#!/bin/bash
# Round-4 prompt-pass profile on the GPU box: per-kernel times of a 780-position prefill (tools/prefill_probe.py 780).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pf780 -- python3 $R/tools/prefill_probe.py 780 > $O/pf780.log 2>&1 || exit 1
cd $R
python3 tools/summarize_rocprof.py stats $O/pf780 $O/pf780_kernel_stats.md || exit 2
grep "Lp=" $O/pf780.log
