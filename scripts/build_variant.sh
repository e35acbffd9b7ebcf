#!/bin/bash
# Experiment builds of the library next to the product one: scripts/build_variant.sh NAME "-DSP_EXP=1 ..." -> similaripy_amd/lib/var_NAME.so
# (run a script against it with SIMILARIPY_AMD_LIB=similaripy_amd/lib/var_NAME.so)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -amdgpu-atomic-optimizer-strategy=None -fPIC -shared \
  -Wno-unused-function $2 -I include -o similaripy_amd/lib/var_$1.so similaripy_amd/csrc/sp_knn.hip
