#!/bin/bash
# Collects the round-3 profiles on the GPU box (run from the repo root through gpurun); raw output under gpurun_out/prof_r03/.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch-probe > $O/stats.log 2>&1 || exit 1
echo stats done
export FT_NO_GRAPH=1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch-probe > $O/fetch.log 2>&1 || exit 2
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch-probe > $O/write.log 2>&1 || exit 3
echo write done
unset FT_NO_GRAPH
cd $R
XL=1 timeout -k 10 120 tools/bin/mb_engine 28 150 32 1 > $O/mb_slow_xl.log 2>&1 || exit 6
timeout -k 10 120 tools/bin/mb_engine 28 150 8 1 > $O/mb_slow_r02form.log 2>&1 || exit 6
timeout -k 10 120 tools/bin/mb_engine 0 > $O/mb_fast.log 2>&1 || exit 7
echo harness done
python3 bench.py > $O/bench.json 2> $O/bench.err || exit 8
echo bench done
