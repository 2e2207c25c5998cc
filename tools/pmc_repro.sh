#!/bin/bash
# does the bench survive a rocprofv3 --pmc pass?  usage: tools/pmc_repro.sh "ENV=VAL ..." n
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for e in $1; do export $e; done
ok=0; bad=0
for i in $(seq 1 ${2:-3}); do
  if timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_repro -- python3 $R/bench.py --steps 30 --warmup 20 --cpu-steps 0 --no-roofline --headline-only > /dev/null 2> $R/gpurun_out/pmc_repro.err; then ok=$((ok+1)); else bad=$((bad+1)); grep -h "MembraneHipError" $R/gpurun_out/pmc_repro.err | tail -1; fi
done
echo "[$1] ok=$ok bad=$bad"
