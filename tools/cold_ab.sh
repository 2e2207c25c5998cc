#!/bin/bash
# Run on the GPU box: the driver's short bench and the default one, with and without one policy switch.
# usage: tools/cold_ab.sh ENVVAR      (runs ENVVAR=0 and ENVVAR unset)
set +e
V=${1:-MS_ESCALATE}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/cold_ab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
F="--cpu-steps 0 --no-roofline --headline-only"
for i in 1 2 3; do
  env $V=0 python3 $R/bench.py --steps 20 --warmup 5 $F > $O/off_s20_$i.json 2> $O/off_s20_$i.err
  python3 $R/bench.py --steps 20 --warmup 5 $F > $O/on_s20_$i.json 2> $O/on_s20_$i.err
done
env $V=0 python3 $R/bench.py $F > $O/off_default.json 2> $O/off_default.err
python3 $R/bench.py $F > $O/on_default.json 2> $O/on_default.err
for f in $O/*.json; do python3 - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], round(d["value"]), d["steps_accepted"], d["line_search_trials"], round(d["evaluations_per_s"]))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
