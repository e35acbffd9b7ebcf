"""Compact view of bench.py JSON lines (stdin): kernel time and cycles per row by phase."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r, cfg = d["roofline"], d["config"]
    ps = cfg["phase_share"]
    rows = max(1, cfg.get("rows_per_gpu", 1))
    wgs_cycles = ps["cycles_per_wg"]
    names = ("setup", "segments", "sweep1", "sweep2", "accumulate", "drain", "select", "output")
    tot_share = sum(ps[n] for n in names) or 1.0
    print(f"kernel {r['kernel_ms_avg']:.2f} ms  frac {r['frac']:.3f}  step {d['ms_per_step']:.2f} ms  value {d['value']:.3e}  "
          f"cycles/wg {wgs_cycles:.3e}  " + " ".join(f"{n}={ps[n]:.3f}" for n in names) +
          f"  sparse_rows={ps['rows_sparse_path']} fallback={ps.get('rows_fallback', ps.get('rows_fallback_cs_full'))}")
