#!/bin/bash
# kernel trace of a short bench: dispatch-by-dispatch listing (gpurun_out/$1_sequence.txt) and rocprofv3's kernel stats
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; T=${1:-trace}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${T}_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_trace -- python3 $R/bench.py --cpu-steps 0 --no-roofline --headline-only "$@" > $O/${T}_trace.json 2> $O/${T}_trace.err
python3 $R/tools/kernel_sequence.py $O/${T}_trace 260 > $O/${T}_sequence.txt 2>&1
f=$(ls $O/${T}_trace/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-200
