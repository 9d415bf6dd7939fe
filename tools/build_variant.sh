#!/bin/bash
# usage: build_variant.sh <name> <extra -D flags...>   -> gama_tts_amd/lib_variants/libgama_vtm_<name>.so
set -e
name=$1; shift
cd "$(dirname "$(readlink -f "$0")")/../gama_tts_amd/csrc"
mkdir -p _obj_$name ../lib_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden "$@" -c vtm_kernels.hip -o _obj_$name/vtm_kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden "$@" -DGVTM_DIAGNOSTICS -x hip -c vtm_capi.cpp -o _obj_$name/vtm_capi.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_variants/libgama_vtm_$name.so _obj_$name/vtm_kernels.o _obj_$name/vtm_capi.o _obj/vtm_tracks.o _obj/vtm_design.o _obj/vtm_diag_kernels.o
echo built $name
