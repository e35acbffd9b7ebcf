"""Launches of ONE call in a rocprofv3 --kernel-trace csv: those between the n-th and (n+1)-th sp_row_work_kernel.
usage: python scripts/show_trace_of_call.py trace.csv [n=2]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
idx = [i for i, r in enumerate(rows) if "sp_row_work_kernel" in r["Kernel_Name"]]
a = idx[n]
b = idx[n + 1] if n + 1 < len(idx) else len(rows)
t0 = int(rows[max(0, a - 4)]["Start_Timestamp"])
for r in rows[max(0, a - 4):b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:9.1f} us  {name:60s} grid {r.get('Grid_Size', '')} wg {r.get('Workgroup_Size', '')}")
