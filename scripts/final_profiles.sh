#!/bin/bash
# Run on the GPU box from the repo root (via gpurun): everything profiles/ quotes for one round, on the current build.
#   bash scripts/final_profiles.sh r04
set -u
TAG=${1:-r04}
REPO=$(pwd)
mkdir -p gpurun_out
# kernel statistics + FETCH_SIZE / WRITE_SIZE passes of the headline step (C2), resident operands, no public call in the process
timeout 900 bash scripts/profile_gpu.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
# the driver-style bench line: headline + live traffic + other_workloads c3 / c5 / c4 + end to end + cpu_baseline
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.log
# SQ / LDS counters of the headline kernel and of the wave kernel (c5)
timeout 600 bash scripts/pmc_gpu.sh $TAG "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" -- > gpurun_out/pmc_sq_totals_$TAG.txt 2>&1
timeout 600 bash scripts/pmc_gpu.sh ${TAG}_c5 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum SQ_BUSY_CU_CYCLES" -- --workload c5 > gpurun_out/pmc_sq_totals_${TAG}_c5.txt 2>&1
# the generic kernel's counters on configs[3], and its heavy and light rows timed apart
KERNEL_RE=sp_knn_generic timeout 600 bash scripts/pmc_gpu.sh ${TAG}_c4 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" -- --workload c4 > gpurun_out/pmc_sq_totals_${TAG}_c4.txt 2>&1
{
  for w in c3 c1 c4 c5; do
    echo "== bench.py --workload $w"; timeout 600 python bench.py --workload $w --no-cpu-baseline --no-end-to-end --no-traffic 2>/dev/null | python scripts/show_bench.py
  done
  echo "== rocprofv3 --kernel-trace --stats of bench.py --workload c5 (the wave kernel)"
  export TMPDIR=/tmp; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_c5_$TAG -o stats -- python $REPO/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-end-to-end > /dev/null 2>&1)
  find /tmp/rp_c5_$TAG -name "*kernel_stats.csv" -exec head -6 {} \; | cut -c1-200
  echo "== scripts/c2_phases.py 200000"; timeout 600 python scripts/c2_phases.py 200000 2>&1 | grep -v amdgpu.ids | tail -3
  echo "== scripts/c2_phases.py 200000 c3"; timeout 600 python scripts/c2_phases.py 200000 c3 2>&1 | grep -v amdgpu.ids | tail -3
  echo "== scripts/profile_public_call.py c2"; timeout 600 python scripts/profile_public_call.py c2 2>&1 | grep "=="
  echo "== scripts/profile_public_call.py c4"; timeout 600 python scripts/profile_public_call.py c4 2>&1 | grep "=="
  echo "== scripts/profile_public_call.py c5"; timeout 600 python scripts/profile_public_call.py c5 2>&1 | grep "=="
  echo "== scripts/c4_heavy_probe.py"; timeout 600 python scripts/c4_heavy_probe.py 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== scripts/public_variants.py"; timeout 900 python scripts/public_variants.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/other_workloads_$TAG.txt 2>&1
tail -8 gpurun_out/bench_$TAG.log; python scripts/show_bench.py < gpurun_out/bench_$TAG.json; cat gpurun_out/other_workloads_$TAG.txt; tail -20 gpurun_out/pmc_sq_totals_$TAG.txt; head -5 gpurun_out/prof_$TAG/kernel_stats.csv | cut -c1-160
