#!/bin/bash
# Builds libfmc variants that differ only in spatial_attn.hip compile flags: build_variants.sh name1:"-DSA_DBG=1" name2:"..." ...
# -> tools/ubench/libs/libfmc_<name>.so (git-ignored; travels with gpurun).  Other objects come from synfmc_amd/lib/obj.
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/ubench/libs /tmp/sa_obj
OBJ=synfmc_amd/lib/obj
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-honor-nans -Wno-unused-function $flags -c synfmc_amd/csrc/spatial_attn.hip -o /tmp/sa_obj/sa_$name.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC /tmp/sa_obj/sa_$name.o $OBJ/norm_kernels.o $OBJ/cond_kernels.o $OBJ/temporal_attn.o $OBJ/gemm_conv.o $OBJ/spatial_attn_bwd.o $OBJ/error.o -o tools/ubench/libs/libfmc_$name.so && echo built $name ) &
done
wait
