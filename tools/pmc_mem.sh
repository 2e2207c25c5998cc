#!/bin/bash
# Memory-side counters of the bench's kernels (own rocprofv3 --pmc passes). Usage: tools/pmc_mem.sh <tag>
TAG=${1:-mem}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export MS_SPECULATE=0   # every dispatch a real one
B2="python3 $R/bench.py --steps 10 --warmup 10 --cpu-steps 0 --no-roofline"
i=0
for set in "MemUnitBusy MemUnitStalled WriteUnitStalled" "L2CacheHit VALUBusy LDSBankConflict" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_BUSY_avr"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_m$i -- $B2 > /dev/null 2> $R/gpurun_out/${TAG}_m$i.err || echo "pass $i failed"
done
echo collected
