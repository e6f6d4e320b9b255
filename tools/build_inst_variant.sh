#!/bin/bash
# A/B variant of the SAMPLER kernels only, (KS, KO) = (3, 3): kernels_inst.hip compiled with the given flags and linked against the
# variant build's main object (make -C biolith_amd/csrc variant NAME=base first).  usage: tools/build_inst_variant.sh NAME [-D...]
set -e
cd "$(dirname "$0")/../biolith_amd/csrc"
NAME=$1; shift
F="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -ffp-contract=on -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value -DBL_ONLY33 -DBL_KS=3 -DBL_KO=3"
/opt/rocm/bin/hipcc $F "$@" -c kernels_inst.hip -o build/var_${NAME}_inst.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libbiolith_hip_${NAME}.so build/var_base_main.o build/var_${NAME}_inst.o
echo built libbiolith_hip_${NAME}.so
