#!/bin/bash
# generic (runtime tile size) instances at three tile sizes, next to the default T = 256 instances
for cfg in "X=1 --tile 256" "MS_NO_FAST=1 --tile 256" "MS_NO_FAST=1 --tile 512" "MS_NO_FAST=1 --tile 128"; do
  e=${cfg%% *}; a=${cfg#* }
  env $e python bench.py --cpu-steps 0 --headline-only --steps 200 --warmup 30 $a > gpurun_out/tp.json 2> gpurun_out/tp.err || { echo "$cfg FAILED"; tail -n 3 gpurun_out/tp.err; continue; }
  python - "$cfg" <<'PY'
import json, sys
d = json.load(open("gpurun_out/tp.json"))
print(f"{sys.argv[1]:28s} {d['value']:8.0f} steps/s  acc {d['steps_accepted']} trials {d['line_search_trials']} ",
      {n: round(v["avg_us"], 1) for n, v in d["kernels"].items()})
PY
done
