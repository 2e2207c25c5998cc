#!/bin/bash
# the headline bench across its flags (one line per variant): a smoke of the less-travelled paths of the step logic
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for a in "" "--deterministic" "--volume" "--freq 81" "--freq 81 --volume" "--tile 128" "--reuse-level 0" "--reuse-level 1" "--freq 40 --steps 400"; do
  v=$(python3 bench.py --cpu-steps 0 --no-roofline --headline-only $a 2>gpurun_out/matrix.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['steps_accepted'], d['line_search_trials'], {k:v for k,v in d['line_search_queue'].items() if k!='note'})" 2>&1)
  echo "[$a] $v"
  tail -2 gpurun_out/matrix.err | grep -i "error\|Traceback" 
done
python3 -c "import __graft_entry__ as g; g.smoke()"
