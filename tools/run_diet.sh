set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1
bash tools/bench_kernels.sh X=1 X=2 X=3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/diet_sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 20 --cpu-steps 0 --no-roofline --headline-only > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/diet_sq.err
python3 $GRAFT_REPO_ROOT/tools/pmc_quick.py $GRAFT_REPO_ROOT/gpurun_out/diet_sq > $GRAFT_REPO_ROOT/gpurun_out/diet_sq.txt
