set +e
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh ${1:-r02c}
python bench.py > gpurun_out/${1:-r02c}_bench.json 2> gpurun_out/${1:-r02c}_bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${1:-r02c}_bench_s20w5.json 2> gpurun_out/${1:-r02c}_bench_s20w5.err
echo finished
