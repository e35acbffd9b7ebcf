#!/bin/bash
# Registers / scratch / LDS of every kernel in the library (compile-only, runs without a GPU).
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wno-unused-function -I include \
  -Rpass-analysis=kernel-resource-usage -c similaripy_amd/csrc/sp_knn.hip -o /tmp/sp_knn_res.o 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|SGPRs:" | sed 's/.*remark: [^ ]* *//' | paste - - - - - | grep -v "SGPRs: 0 "
