#!/bin/bash
# Build an alternative libmembrane_hip.so into build_variants/lib_<name>.so with extra
# compiler flags for ms_kernels.hip (A/B and ablation builds; see tools/ab_variants.sh).
#   tools/build_variant.sh NAME "-DMS_FOO=1 ..." [api-flags]
set -e
name=$1; shift
kflags=$1; shift || true
aflags=$1
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/membrane_solver_amd/csrc
obj=$root/build_variants/obj_$name
mkdir -p "$obj"
common="-DMS_VARIANT_ENV=1 -DMS_ABL_NTILES_ENV=1 --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -ffp-contract=fast -I$root/include -I$src"
/opt/rocm/bin/hipcc $common $kflags -c "$src/ms_kernels.hip" -o "$obj/ms_kernels.o" &
/opt/rocm/bin/hipcc $common $kflags $aflags -x hip -c "$src/ms_api.cpp" -o "$obj/ms_api.o" &
/opt/rocm/bin/hipcc $common -x hip -c "$src/ms_tiles.cpp" -o "$obj/ms_tiles.o" &
wait
g++ -shared -o "$root/build_variants/lib_$name.so" "$obj/ms_kernels.o" "$obj/ms_api.o" "$obj/ms_tiles.o" -Wl,--allow-shlib-undefined
echo "built build_variants/lib_$name.so"
