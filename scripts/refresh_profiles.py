"""Copy the outputs of `scripts/profile_gpu.sh <tag>` + `bench.py` from gpurun_out/ into profiles/ under the round's
names and derive profiles/hbm_traffic.json from the two PMC passes (see profiles/README.md for the corrections).
    python scripts/refresh_profiles.py r01"""
import csv, hashlib, json, shutil, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent


def lib_source_sha() -> str:
    """Same hash as bench.py: the traffic entry is valid for the build it was measured on only."""
    h = hashlib.sha256()
    for f in sorted((ROOT / "similaripy_amd" / "csrc").glob("*")):
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = ROOT / "gpurun_out" / f"prof_{tag}", ROOT / "profiles"
shutil.copy(src / "kernel_stats.csv", dst / f"{tag}_kernel_stats.csv")
shutil.copy(src / "kernel_trace_head.csv", dst / f"{tag}_kernel_trace_head.csv")
shutil.copy(src / "pmc_FETCH_SIZE.csv", dst / f"{tag}_pmc_FETCH_SIZE.csv")
shutil.copy(src / "pmc_WRITE_SIZE.csv", dst / f"{tag}_pmc_WRITE_SIZE.csv")
bench = ROOT / "gpurun_out" / f"bench_{tag}.json"
if bench.exists():
    shutil.copy(bench, dst / f"{tag}_bench.json")


def last_value(path, counter):
    # the dispatch of the kernel that did the step's work: the one with the largest value (round 6 launches the two-per-CU shape a second time
    # over the queue of heavier rows — empty for configs[1]: 33 us, a few hundred KiB)
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    per = {}
    for r in rows:
        per[int(r["Dispatch_Id"])] = per.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return max(per.values())


f, w = last_value(src / "pmc_FETCH_SIZE.csv", "FETCH_SIZE"), last_value(src / "pmc_WRITE_SIZE.csv", "WRITE_SIZE")
out = {"c2:1000000x100000x64:k100": {
    "bytes_per_launch": f * 1024 * 2 + w * 1024, "fetch_size_kib": round(f, 2), "write_size_kib": round(w, 2),
    "formula": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE reports half of a coalesced stream, see profiles/README.md)",
    "kernel": "sp_knn_sparse_kernel<512,true,1,true> (MODE 1 = the monotone variant, DUO = the two-per-CU shape of round 6)", "lib_source_sha": lib_source_sha(),
    "source": f"profiles/{tag}_pmc_FETCH_SIZE.csv, profiles/{tag}_pmc_WRITE_SIZE.csv"}}
(dst / "hbm_traffic.json").write_text(json.dumps(out, indent=1))
for line in open(dst / f"{tag}_kernel_stats.csv").read().splitlines()[:3]:
    print(line[:200])
print(json.dumps(out, indent=1))
