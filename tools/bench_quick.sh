#!/bin/bash
# usage: tools/bench_quick.sh TAG [tests]   -- bench at the default and at the driver's setting (+ the GPU suite)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; T=$1
cd $R
if [ "$2" = "tests" ]; then python -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1; tail -3 $O/${T}_tests.log; fi
python3 bench.py --cpu-steps 0 > $O/${T}_bench.json 2> $O/${T}_bench.err
python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 > $O/${T}_bench20.json 2> $O/${T}_bench20.err
python3 - <<PY
import json
for f in ["$O/${T}_bench.json","$O/${T}_bench20.json"]:
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"FAILED",e); print(open(f.replace(".json",".err")).read()[-1500:]); continue
    print(f.split("/")[-1], round(d["value"]), d.get("steps_accepted"), d.get("line_search_trials"), {k:v for k,v in d.get("line_search_queue",{}).items() if k!="note"})
    for k,v in (d.get("kernels") or {}).items(): print("   ",k, round(v["avg_us"],2), v["launches"], round(v["share_of_profiled_ms"],3))
    if "deterministic_mode" in d: print("    det", round(d["deterministic_mode"]["value"]), "lvl0", round(d["all_reference_passes_repeated"]["value"]))
PY
