set +e
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r02b
python bench.py > gpurun_out/r02b_bench.json 2> gpurun_out/r02b_bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r02b_bench_s20w5.json 2> gpurun_out/r02b_bench_s20w5.err
echo finished
