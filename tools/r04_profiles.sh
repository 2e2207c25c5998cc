#!/bin/bash
# Run on the GPU box: every BASELINE configuration under rocprofv3 --kernel-trace --stats (+ the headline's PMC passes)
# and the bench lines.  usage: tools/r04_profiles.sh <tag>     -> gpurun_out/<tag>_*
set +e
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
bash tools/collect_profiles.sh $TAG || exit $?
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
guard() { rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass killed at its limit (rc $rc): stopping"; exit $rc; fi; }
prof() {  # prof <name> <script and args...>
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_${name}_stats -- python3 "$@" \
      > $R/gpurun_out/${TAG}_${name}.json 2> $R/gpurun_out/${TAG}_${name}.err
  guard
  echo "profiled $name"
}
prof config2_resident $R/tools/bench_config2.py 2000
MS_RESIDENT=0 prof config2_kernel_per_phase $R/tools/bench_config2.py 400
prof config3_volume_row $R/tools/bench_config3v.py 100
prof config5_deck $R/tools/bench_config5.py 40
MS_EXEC=0 prof config5_deck_launch_per_kernel $R/tools/bench_config5.py 20
prof tilt_single_field_2M $R/tools/bench_tilt.py --steps 10
prof tilt_two_leaflets_2M $R/tools/bench_tilt.py --leaflet --steps 6
cd $R
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_s20w5.json 2> gpurun_out/${TAG}_bench_s20w5.err
MS_BENCH_FORCE_SHARDED=1 python bench.py --no-large > gpurun_out/${TAG}_bench_forced_sharded.json 2> gpurun_out/${TAG}_bench_forced_sharded.err
echo finished
