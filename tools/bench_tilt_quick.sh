#!/bin/bash
# usage: tools/bench_tilt_quick.sh TAG [tests]  -- tools/bench_tilt.py in both forms (per-kernel averages), optionally the tilt GPU tests first
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; T=$1
cd $R
if [ "$2" = "tests" ]; then python -m pytest tests/test_gpu_bending_tilt.py tests/test_gpu_leaflet.py -m gpu -x -q > $O/${T}_tests.log 2>&1; tail -3 $O/${T}_tests.log; fi
for mode in "" "--leaflet"; do
  PYTHONPATH=$R python3 tools/bench_tilt.py $mode 2>> $O/${T}.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$mode', round(d['steps_per_s'],1), 'steps/s relax_ms', round(d['relax_ms'],2), {k:(round(v['avg_us'],1), round(v['launches_per_step'],1)) for k,v in d['kernels_per_step'].items()})"
done
