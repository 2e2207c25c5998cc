#!/bin/bash
# Run on the GPU box: the driver's short bench (--steps 20 --warmup 5) next to warmer runs, plus a kernel trace of it.
set +e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/cold
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
F="--cpu-steps 0 --no-roofline --headline-only"
for i in 1 2 3; do
  python3 $R/bench.py --steps 20 --warmup 5 $F > $O/s20_w5_$i.json 2> $O/s20_w5_$i.err
done
python3 $R/bench.py --steps 20 --warmup 200 $F > $O/s20_w200.json 2> $O/s20_w200.err
python3 $R/bench.py --steps 200 --warmup 5 $F > $O/s200_w5.json 2> $O/s200_w5.err
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 5 $F > $O/trace.json 2> $O/trace.err
python3 $R/tools/kernel_sequence.py $O/trace 400 > $O/sequence.txt 2>&1
echo done
