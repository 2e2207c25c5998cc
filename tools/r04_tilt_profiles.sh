#!/bin/bash
# Run on the GPU box: the tilt rows at 2 M facets once more (rocprofv3 kernel stats + the two bench_tilt measurements
# without the profiler).  usage: tools/r04_tilt_profiles.sh <tag>   -> gpurun_out/<tag>_*
set +e
TAG=${1:-r04d}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
guard() { rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass killed at its limit (rc $rc): stopping"; exit $rc; fi; }
prof() {
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_${name}_stats -- python3 "$@" \
      > $R/gpurun_out/${TAG}_${name}.json 2> $R/gpurun_out/${TAG}_${name}.err
  guard
  echo "profiled $name"
}
prof tilt_single_field_2M $R/tools/bench_tilt.py --steps 10
prof tilt_two_leaflets_2M $R/tools/bench_tilt.py --leaflet --steps 6
cd $R
python3 tools/bench_tilt.py --steps 10 --events-first > gpurun_out/${TAG}_tilt_single_field_2M_events_first.json 2>/dev/null
python3 tools/bench_tilt.py --leaflet --steps 6 --events-first > gpurun_out/${TAG}_tilt_two_leaflets_2M_events_first.json 2>/dev/null
echo finished
