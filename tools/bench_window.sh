#!/bin/bash
# usage: tools/bench_window.sh TAG [N=5]  -- the driver's setting (--steps 20 --warmup 5), headline only, N times + a cold trace
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; T=$1; N=${2:-5}
cd $R
python3 tools/cold_trace.py 30 > $O/${T}_cold.txt 2>&1
for i in $(seq 1 $N); do
  python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --headline-only --no-roofline 2>> $O/${T}_w.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w20', round(d['value']), d.get('steps_accepted'), d.get('line_search_trials'), {k:v for k,v in d.get('line_search_queue',{}).items() if k!='note'})"
done
python3 bench.py --cpu-steps 0 --headline-only --no-roofline 2>> $O/${T}_w.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w200', round(d['value']), d.get('steps_accepted'), d.get('line_search_trials'), {k:v for k,v in d.get('line_search_queue',{}).items() if k!='note'})"
MS_TRACE_STEPS=1 python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --headline-only --no-roofline 2>&1 >/dev/null | grep "^\[mss\]" > $O/${T}_steps.txt
