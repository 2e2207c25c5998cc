#!/bin/bash
# SQ / LDS / clock counters of the bench's kernels (own rocprofv3 --pmc passes). Usage: tools/pmc_sq.sh <tag>
set -e
TAG=${1:-sq}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
B2="python3 $R/bench.py --steps 10 --warmup 10 --cpu-steps 0 --no-roofline"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_sq1 -- $B2 > /dev/null 2> $R/gpurun_out/${TAG}_sq1.err
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/${TAG}_sq2 -- $B2 > /dev/null 2> $R/gpurun_out/${TAG}_sq2.err || true
echo collected
