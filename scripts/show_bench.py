"""Pretty-print a bench.py JSON line from stdin: rate, roofline fraction and per-row cycles by phase."""
import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    ph = d["config"]["phase_share"]
    rows_per_wg = d["config"]["rows_per_gpu"] / 256.0
    cyc_row = ph["cycles_per_wg"] / rows_per_wg
    print(f"{d['roofline']['kernel_ms_avg']:.1f} ms  {d['value']/1e6:.2f} M rows/s  frac {d['roofline']['frac']:.3f}  cycles/row {cyc_row:.0f}")
    names = ("setup", "segments", "accumulate", "drain", "select", "output", "sweep1", "sweep2", "csdrain")
    print("  " + "  ".join(f"{n}={ph[n]*cyc_row:.0f}" for n in names))
    print(f"  sparse rows {ph['rows_sparse_path']}  fallback {ph['rows_fallback_cs_full']}  generic windows {ph['generic_windows']}")
