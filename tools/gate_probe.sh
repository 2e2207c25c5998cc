#!/bin/bash
# One plain run and one run under rocprofv3 --pmc of tools/gate_probe.py (variant build: tools/build_variant.sh
# gateprobe "-DMS_GATE_PROBE=1").  Results: gpurun_out/gate_probe_plain.json, gate_probe_pmc.json.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export MEMBRANE_HIP_LIB=$R/build_variants/lib_gateprobe.so
python3 $R/tools/gate_probe.py > $R/gpurun_out/gate_probe_plain.json 2> $R/gpurun_out/gate_probe_plain.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/gate_probe_pmc -- python3 $R/tools/gate_probe.py > $R/gpurun_out/gate_probe_pmc.json 2> $R/gpurun_out/gate_probe_pmc.err
cat $R/gpurun_out/gate_probe_plain.json $R/gpurun_out/gate_probe_pmc.json
