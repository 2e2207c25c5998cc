#!/bin/bash
# repeat a rocprofv3 --pmc pass of the bench with the queue trace on until it fails (at most n times)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export MS_TRACE_QUEUE=1
for i in $(seq 1 ${1:-6}); do
  if ! timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_trace -- python3 $R/bench.py --steps 40 --warmup 20 --cpu-steps 0 --no-roofline --headline-only > /dev/null 2> $R/gpurun_out/pmc_trace.err; then
    echo "FAILED on try $i"; grep "msq\]" $R/gpurun_out/pmc_trace.err | tail -60 > $R/gpurun_out/pmc_trace_tail.txt; grep -h "MembraneHipError" $R/gpurun_out/pmc_trace.err | tail -1; exit 0
  fi
done
echo "no failure in ${1:-6} tries"
