#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root:  bash scripts/profile_gpu.sh <tag> [bench args...]
# Produces under gpurun_out/prof_<tag>/:
#   kernel_stats.csv     rocprofv3 --kernel-trace --stats summary of `python bench.py ...`
#   pmc_fetch.csv / pmc_write.csv   FETCH_SIZE / WRITE_SIZE in separate --pmc passes
# (MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB units, separate passes, and FETCH_SIZE
#  under-reports wide coalesced reads by 2x on gfx950 — corrections are applied in profiles/README.md).
set -u
TAG=${1:-run}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-other-workloads --no-end-to-end $*"

rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats_$TAG -o stats -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.log"
find /tmp/rp_stats_$TAG -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find /tmp/rp_stats_$TAG -name "*kernel_trace.csv" -exec sh -c 'head -50 "$1" > "$2"' _ {} "$OUT/kernel_trace_head.csv" \;

for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/rp_pmc_${C}_$TAG -o pmc -- $BENCH > /dev/null 2> "$OUT/pmc_$C.log"
  f=$(find /tmp/rp_pmc_${C}_$TAG -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    # keep only our kernel's rows (header + sp_knn rows)
    (head -1 "$f"; grep -E "sp_knn_(sparse|wave)" "$f") > "$OUT/pmc_$C.csv"
  fi
done
ls -la "$OUT"
head -20 "$OUT/kernel_stats.csv"
for C in FETCH_SIZE WRITE_SIZE; do echo "== $C"; head -5 "$OUT/pmc_$C.csv"; done
