#!/bin/bash
# Collects the round-4 profiles on the GPU box (run from the repo root through gpurun); raw output under gpurun_out/prof_r04/.
#   batch-1 bench: kernel stats + the two HBM counter passes (eager frames: counters and stream capture do not go together)
#   lock-step 32 rows: kernel stats (tools/batch_probe.py 32); 780-position prompt pass: kernel stats (tools/prefill_probe.py 780)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
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
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b32 -- python3 $R/tools/batch_probe.py 32 > $O/b32.log 2>&1 || exit 4
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pf780 -- python3 $R/tools/prefill_probe.py 780 > $O/pf780.log 2>&1 || exit 5
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma_pf -- python3 $R/tools/prefill_probe.py 780 > $O/mfma_pf.log 2>&1 || exit 5
echo batch and prompt-pass stats done
cd $R
python3 tools/summarize_rocprof.py stats $O/stats $O/kernel_stats.md || exit 6
python3 tools/summarize_rocprof.py stats $O/b32 $O/b32_kernel_stats.md || exit 6
python3 tools/summarize_rocprof.py stats $O/pf780 $O/pf780_kernel_stats.md || exit 6
python3 tools/summarize_rocprof.py mfma $O/mfma_pf $O/pmc_mfma_prefill.md || exit 6
python3 tools/summarize_rocprof.py traffic $O/fetch $O/write $O/traffic.md || exit 6
python3 bench.py > $O/bench.json 2> $O/bench.err || exit 8
python3 bench.py --config cfg4 --steps 3 --warmup 1 > $O/cfg4_n1.json 2> $O/cfg4.err || exit 9
echo bench done
