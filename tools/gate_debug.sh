#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export MEMBRANE_HIP_LIB=$R/build_variants/lib_stamps.so
for i in 1 2 3; do
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/gate_dbg -- python3 $R/tools/gate_debug.py 2>&1 | grep -v "output_stream\|simple_timer\|tool.cpp\|amdgpu.ids"
done
