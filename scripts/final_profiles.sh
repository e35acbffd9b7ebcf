#!/bin/bash
# Run on the GPU box from the repo root (via gpurun): everything profiles/ quotes for one round, on the current build.
#   bash scripts/final_profiles.sh r03
set -u
TAG=${1:-r03}
REPO=$(pwd)
mkdir -p gpurun_out
timeout 900 bash scripts/profile_gpu.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.log
timeout 900 bash scripts/pmc_gpu.sh $TAG "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" -- > gpurun_out/pmc_sq_totals_$TAG.txt 2>&1
{
  for w in c3 c1 c4 c5; do
    echo "== bench.py --workload $w"; timeout 600 python bench.py --workload $w --no-cpu-baseline --no-end-to-end 2>/dev/null | python scripts/show_bench.py
  done
  echo "== scripts/c2_phases.py 200000"; timeout 600 python scripts/c2_phases.py 200000 2>&1 | grep -v amdgpu.ids | tail -3
  echo "== scripts/c2_phases.py 200000 c3"; timeout 600 python scripts/c2_phases.py 200000 c3 2>&1 | grep -v amdgpu.ids | tail -3
  echo "== scripts/profile_public_call.py c2"; timeout 600 python scripts/profile_public_call.py c2 2>&1 | grep "=="
  echo "== scripts/profile_public_call.py c4"; timeout 600 python scripts/profile_public_call.py c4 2>&1 | grep "=="
} > gpurun_out/other_workloads_$TAG.txt 2>&1
tail -3 gpurun_out/bench_$TAG.log; cat gpurun_out/bench_$TAG.json | cut -c1-600; cat gpurun_out/other_workloads_$TAG.txt; cat gpurun_out/pmc_sq_totals_$TAG.txt | tail -20; head -5 gpurun_out/prof_$TAG/kernel_stats.csv | cut -c1-160
