#!/bin/bash
# usage: tools/bench_full.sh TAG [tests] -- the GPU suite (optional) and the full default bench line with its secondary configurations
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; T=$1
cd $R
if [ "$2" = "tests" ]; then python -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1; tail -3 $O/${T}_tests.log; fi
python3 bench.py --cpu-steps 0 ${NOLARGE---no-large} > $O/${T}_full.json 2> $O/${T}_full.err
python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --headline-only --no-roofline > $O/${T}_b20.json 2> $O/${T}_b20.err
python3 - <<PY
import json
d=json.loads(open("$O/${T}_full.json").read().strip().splitlines()[-1])
print("headline", round(d["value"]), "busy", round(d["kernel_busy_share"]["value"],3), "det", round(d["deterministic_mode"]["value"]), "lvl0", round(d["all_reference_passes_repeated"]["value"]))
print("   ", {k:round(v["avg_us"],1) for k,v in d["kernels"].items()}, {k:v for k,v in d["line_search_queue"].items() if k!="note"})
for k in ("config2","config3_volume_row"):
    c=d.get(k)
    if not c: print(k,"missing"); continue
    print(k, round(c["value"]), "acc", c["steps_accepted"], "trials", c["line_search_trials"], "vs_no_row", c.get("vs_no_row"))
    print("   ", {kk:round(v,1) for kk,v in c["kernels_avg_us"].items()})
    print("   ", {kk:(round(v["avg_us"],1), round(v["frac_of_hbm_peak"],3)) for kk,v in c["gradient_instance"].items()}, c["line_search_queue"])
d=json.loads(open("$O/${T}_b20.json").read().strip().splitlines()[-1])
print("driver setting", round(d["value"]), {k:v for k,v in d["line_search_queue"].items() if k!="note"})
PY
