#!/bin/bash
# the headline bench at f = 640 (8.2 M facets: working set beyond the 256 MiB Infinity Cache) and the default line with its 16 M-facet leg
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; T=$1
cd $R
python3 bench.py --freq 640 --steps 100 --warmup 20 --cpu-steps 0 --headline-only > $O/${T}_f640.json 2> $O/${T}_f640.err
python3 bench.py --cpu-steps 0 > $O/${T}_large.json 2> $O/${T}_large.err
python3 - <<PY
import json
d=json.loads(open("$O/${T}_f640.json").read().strip().splitlines()[-1])
print("f640", round(d["value"]), {k:(round(v["avg_us"],1), round(v.get("frac_of_hbm_peak",0),3)) for k,v in d["kernels"].items()}, "E+grad frac", d["roofline"].get("energy_plus_gradient_evaluation_frac"))
d=json.loads(open("$O/${T}_large.json").read().strip().splitlines()[-1])
print("headline", round(d["value"]), "E+grad frac", d["roofline"].get("energy_plus_gradient_evaluation_frac"), "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"],3))
lg=d.get("strong_16M_facets")
print("16M", lg and (round(lg["value"],1), lg["steps_accepted"], lg["line_search_trials"], {k:round(v,1) for k,v in lg["kernels_avg_us"].items()}))
PY
