#!/bin/bash
# AddressSanitizer + UBSan run of the tile builder (host code only: CPU, no GPU needed) over tile sizes 64..512, rows per
# tile below the thread count, one and three shards, facets with bad indices.  usage: tools/asan_tiles.sh
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/membrane_solver_amd/csrc
mkdir -p /tmp/ms_asan
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 --cuda-host-only -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer \
  -std=c++17 -I$C -I$R/include $R/tools/micro/asan_tiles.cpp $C/ms_tiles.cpp -o /tmp/ms_asan/asan_tiles || exit 1
/tmp/ms_asan/asan_tiles | tail -4
echo "exit $?"
