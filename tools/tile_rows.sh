#!/bin/bash
# usage: tools/tile_rows.sh TAG rows...   -- headline bench (200/30, headline only) for tiles owning the given numbers of rows
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; T=$1; shift
cd $R
for rows in "$@"; do
  python3 bench.py --tile $rows --cpu-steps 0 --headline-only 2>> $O/${T}.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernels') or {}
print('rows', $rows, round(d['value']), d.get('steps_accepted'), d.get('line_search_trials'), {n:round(v['avg_us'],1) for n,v in k.items()}, d['line_search_queue']['mismatches'])"
done
