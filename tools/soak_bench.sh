#!/bin/bash
# Run the headline bench N times (robustness soak): one line per run, failures shown.
N=${1:-10}
for i in $(seq 1 $N); do
  python bench.py --cpu-steps 0 --headline-only --no-roofline --steps 400 --warmup 20 > gpurun_out/sb.json 2> gpurun_out/sb.err \
    && python -c "import json; d=json.load(open('gpurun_out/sb.json')); print('run $i', round(d['value']), d['steps_accepted'], d['line_search_trials'], d['energy_end'])" \
    || { echo "run $i FAILED"; tail -n 5 gpurun_out/sb.err; }
done
