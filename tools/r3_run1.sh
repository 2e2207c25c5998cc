#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
F="--cpu-steps 0"
python3 $R/bench.py $F > $O/r3a_bench.json 2> $O/r3a_bench.err
python3 $R/bench.py --steps 20 --warmup 5 $F > $O/r3a_bench20.json 2> $O/r3a_bench20.err
$R/tools/gate_probe.sh > $O/gate_probe.txt 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r3a_pmc -- python3 $R/bench.py --steps 40 --warmup 20 --cpu-steps 0 --no-roofline --headline-only > $O/r3a_pmc_bench.json 2> $O/r3a_pmc_bench.err
echo "pmc bench rc=$?"
cut -c1-400 $O/r3a_bench.json; echo; cut -c1-400 $O/r3a_bench20.json; echo; cat $O/gate_probe.txt; cut -c1-300 $O/r3a_pmc_bench.json
