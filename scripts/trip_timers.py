"""Profiling run with the instrumented build (-DSP_TRIPTIMERS=1): share of the sweeps that wave 0 spends waiting for its data.
usage: SIMILARIPY_AMD_LIB=similaripy_amd/lib/libsimilaripy_hip_timers.so python scripts/trip_timers.py [bench args]"""
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-end-to-end", "--steps", "2"] + sys.argv[1:], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
d = json.loads(line)
ps = d["config"]["phase_share"]
tot = ps["cycles_per_wg"]
rows = d["config"]["rows_per_gpu"]
# bench.py reports slot 8 as 'csdrain' share (of the sum of slots 0..8) and slot 11 as generic_windows (raw cycles summed over workgroups)
wgs = 256
names = ("setup", "segments", "sweep1", "sweep2", "accumulate", "drain", "select", "output", "csdrain")
per_row = {n: ps[n] * tot * wgs / rows for n in names}
s1_wait = ps["generic_windows"] / rows
print(f"kernel {d['roofline']['kernel_ms_avg']:.2f} ms;  cycles per row (wave 0 of each WG):")
print("  " + "  ".join(f"{n}={v:.0f}" for n, v in per_row.items()))
print(f"  sweep 1: {per_row['sweep1']:.0f} cycles of which waiting for data {s1_wait:.0f}")
print(f"  sweep 2: {per_row['sweep2']:.0f} cycles of which waiting for data {per_row['csdrain']:.0f}")
