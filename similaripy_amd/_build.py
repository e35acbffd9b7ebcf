"""In-tree build of libsimilaripy_hip.so (hipcc, gfx950 only).

The shared object is written next to the sources (``similaripy_amd/lib/``): it is
git-ignored but travels with the gpurun snapshot, and the round-end driver can see
that it is the library the Python process actually loaded.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO_DIR = PKG_DIR.parent
CSRC_DIR = PKG_DIR / "csrc"
LIB_DIR = PKG_DIR / "lib"
LIB_PATH = LIB_DIR / "libsimilaripy_hip.so"

SOURCES = [CSRC_DIR / "sp_knn.hip"]
HEADERS = [REPO_DIR / "include" / "sp_knn.h", REPO_DIR / "include" / "sp_prep.h", *sorted(CSRC_DIR.glob("*.hpp")), *sorted(CSRC_DIR.glob("*.inc"))]

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-munsafe-fp-atomics",  # LDS float adds must lower to ds_add_f32, never a CAS loop
    # one lane's returning atomic on the row queue must stay PENDING until its value is used a row later: the atomic optimizer
    # rewrites it into a wave-aggregated form whose lanes need the result at once (s_waitcnt vmcnt(0) right behind the atomic)
    "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
    "-fPIC",
    "-shared",
    "-Wno-unused-function",
]


def find_hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def is_stale() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    return any(p.exists() and p.stat().st_mtime > t for p in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile the HIP library if missing or older than its sources. Returns its path."""
    srcs = [s for s in SOURCES if s.exists()]
    if not force and not is_stale():
        return LIB_PATH
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    tmp = LIB_PATH.with_suffix(".so.tmp%d" % os.getpid())
    extra = os.environ.get("SIMILARIPY_AMD_HIPCC_EXTRA", "").split()      # profiling builds only, e.g. -DSP_ABLATION=1
    cmd = [find_hipcc(), *HIPCC_FLAGS, *extra, "-I", str(REPO_DIR / "include"), "-o", str(tmp), *map(str, srcs)]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0 and "-mllvm" in cmd and ("amdgpu-atomic-optimizer-strategy" in proc.stderr or "Unknown command line argument" in proc.stderr):
        # the -mllvm option is an LLVM internal, not a stable hipcc interface: a toolchain that no longer knows it still builds the
        # library (a performance detail of the row queue's prefetch is lost, nothing else)
        cmd = [c for i, c in enumerate(cmd) if c != "-mllvm" and (i == 0 or cmd[i - 1] != "-mllvm")]
        proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout + proc.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
