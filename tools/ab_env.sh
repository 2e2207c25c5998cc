#!/bin/bash
# A/B of one environment switch on ONE box: alternating runs of the headline bench (default and driver's setting)
# usage: tools/ab_env.sh VAR valA valB [n]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
V=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq 1 $N); do
  for x in $A $B; do
    for s in "200 30" "20 5"; do
      set -- $s
      v=$(env $V=$x python3 bench.py --steps $1 --warmup $2 --cpu-steps 0 --no-roofline --headline-only 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']))")
      echo "$V=$x steps=$1: $v"
    done
  done
done
