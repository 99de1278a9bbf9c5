#!/bin/bash
# tools/build_variant.sh NAME "-DMACRO=.. ..."  -> pathtracing_amd/libpt_NAME.so (kernel tuning experiments)
set -e
cd "$(dirname "$0")/../pathtracing_amd/csrc"
mkdir -p build/var_$1
/opt/rocm/bin/hipcc $2 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage \
  -c kernels/pt_kernels.hip -o build/var_$1/pt_kernels.o 2> build/var_$1/usage.txt
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libpt_$1.so build/capi.o build/host/*.o build/var_$1/pt_kernels.o -lz -pthread
grep -A10 "search_kernel" build/var_$1/usage.txt | grep -E "VGPRs:|Scratch|VGPRs Spill" | tr '\n' ' '; echo
