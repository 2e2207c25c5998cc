#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + PMC passes of the bench command.
# Usage: tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_*
set +e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
# a pass that hits its time limit ends the script: no further GPU step after a killed one
guard() { rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass killed at its limit (rc $rc): stopping"; exit $rc; fi; }
B="python3 $R/bench.py --steps 60 --warmup 20 --cpu-steps 0 --no-large"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- $B > $R/gpurun_out/${TAG}_stats.json 2> $R/gpurun_out/${TAG}_stats.err
guard
B2="python3 $R/bench.py --steps 40 --warmup 20 --cpu-steps 0 --no-roofline --headline-only"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -- $B2 > /dev/null 2> $R/gpurun_out/${TAG}_fetch.err
guard
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_write -- $B2 > /dev/null 2> $R/gpurun_out/${TAG}_write.err
guard
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_sq -- $B2 > /dev/null 2> $R/gpurun_out/${TAG}_sq.err
guard
echo collected
