#!/bin/bash
# headline bench under a list of environment settings; prints steps/s and per-kernel averages
for e in "$@"; do
  env $e python bench.py --cpu-steps 0 --headline-only --steps 200 --warmup 30 > gpurun_out/bk.json 2> gpurun_out/bk.err || { echo "$e FAILED"; tail -n 3 gpurun_out/bk.err; continue; }
  python - "$e" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bk.json"))
print(f"{sys.argv[1]:24s} {d['value']:8.0f} steps/s  acc {d['steps_accepted']} trials {d['line_search_trials']} ",
      {n: round(v["avg_us"], 1) for n, v in d["kernels"].items()})
PY
done
