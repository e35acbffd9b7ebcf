#!/bin/bash
# Run on the GPU box from the repo root: per-kernel totals of the public item-item call (rocprofv3 --kernel-trace --stats).
#   bash scripts/kernels_of_public_call.sh [c2|ml]
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rp_pc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_pc -o pc -- python $REPO/scripts/trace_public_call.py ${1:-c2} > /tmp/pc_log.txt 2>&1
tail -5 /tmp/pc_log.txt; f=$(find /tmp/rp_pc -name "*kernel_stats.csv" | head -1)
python - "$f" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1e6:9.3f} ms  total {float(r["TotalDurationNs"]) / 1e6:9.2f} ms')
PY
