#!/bin/bash
# Run on the GPU box: rocprofv3 passes + the two bench lines (default, the driver's setting) + the forced-sharded line.
# usage: tools/final_profiles.sh <tag>
set +e
TAG=${1:-r02d}
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh $TAG || exit $?
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_s20w5.json 2> gpurun_out/${TAG}_bench_s20w5.err
MS_BENCH_FORCE_SHARDED=1 python bench.py --no-large > gpurun_out/${TAG}_bench_forced_sharded.json 2> gpurun_out/${TAG}_bench_forced_sharded.err
echo finished
