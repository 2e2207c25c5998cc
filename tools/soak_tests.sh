#!/bin/bash
# Run the GPU test suite N times in one process each (flakiness soak); prints a line per run.
N=${1:-5}
for i in $(seq 1 $N); do
  timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/soak_$i.log 2>&1
  echo "run $i: $(tail -1 gpurun_out/soak_$i.log)"
  grep -h "^FAILED" gpurun_out/soak_$i.log
done
