#!/bin/bash
# bash scripts/pmc_gpu.sh <tag> "<counters pass 1>" "<counters pass 2>" ... -- [bench args]
# Each quoted group is one rocprofv3 --pmc pass (hardware limits: SQ 8, TCC 4 per pass).  KERNEL_RE (env): the kernels counted, default the sparse-row kernels.
set -u
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
PMCG=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do PMCG+=("$1"); shift; done
[ $# -gt 0 ] && shift
BENCH="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-traffic --no-other-workloads --no-end-to-end $*"
i=0
for G in "${PMCG[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d /tmp/rp_${TAG}_$i -o pmc -- $BENCH > /dev/null 2> "$OUT/pass$i.log"
  f=$(find /tmp/rp_${TAG}_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then (head -1 "$f"; grep -E "${KERNEL_RE:-sp_knn_(sparse|wave)}" "$f") > "$OUT/pass$i.csv"; fi
done
python - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
tot = collections.OrderedDict()
for f in sorted(glob.glob(out + "/pass*.csv")):
    rows = list(csv.DictReader(open(f)))
    # one row per (dispatch, counter); keep the LAST dispatch of the kernel (the timed step)
    if not rows: continue
    # (a wave call launches its workgroup-per-row companion behind the wave kernel, on a handful of rows: the wave kernel is the one to count)
    if any("sp_knn_wave" in r["Kernel_Name"] for r in rows): rows = [r for r in rows if "sp_knn_wave" in r["Kernel_Name"]]
    # the dispatch that did the work: the one with the most of this pass's first counter (round 6: the two-per-CU shape is launched a second
    # time over the queue of heavier rows — empty for configs[1] — so the LAST dispatch of the name is no longer the step's kernel)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    first = rows[0]["Counter_Name"]
    best = max(per, key=lambda d: (per[d].get(first, 0.0), d))
    for name, v in per[best].items():
        tot[name] = tot.get(name, 0.0) + v
for k, v in tot.items():
    print(f"{k:32s} {v:20.0f}")
PY
