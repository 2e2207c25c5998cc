#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for e in 0 1 2 3 4; do
  for s in "20 5" "200 30"; do
    set -- $s
    v=$(MS_ESCALATE=$e python3 bench.py --steps $1 --warmup $2 --cpu-steps 0 --no-roofline --headline-only 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['steps_accepted'], d['line_search_trials'], {k:v for k,v in d['line_search_queue'].items() if k!='note'})")
    echo "MS_ESCALATE=$e steps=$1 warmup=$2: $v"
  done
done
