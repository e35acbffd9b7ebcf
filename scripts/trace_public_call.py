"""Stage breakdown of the item-item public call (`sim.cosine(URM.T)`), library side (SIMILARIPY_AMD_TRACE=1) and Python side.
usage: python scripts/trace_public_call.py [c2|ml]"""
import os, sys, time
os.environ["SIMILARIPY_AMD_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import similaripy_amd as sim
from similaripy_amd import workloads, _host

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
urm = workloads.fixed_degree_csr(1_000_000, 100_000, 64, 12345).T.tocsr() if which == "c2" else workloads.movielens_like_urm()
item = urm.T
for fmt in ("csr", "coo"):
    for rep in range(3):
        print(f"-- cosine(URM.T, k=100, format_output={fmt!r}) run {rep}", file=sys.stderr, flush=True)
        out = S = None      # (released outside the timed regions: unmapping 0.8 GB of touched pages takes 30-40 ms)
        t0 = time.perf_counter()
        call = _host.prepare(item, k=100, l2=1.0, m2_on_device=True, norms_on_device=True, csc_direct=True, check_zeros=False)
        t1 = time.perf_counter()
        out = _host.run_hip(call, want_rows=(fmt != "csr"), check_zeros=True, csr_out=(fmt == "csr"))
        t2 = time.perf_counter()
        out = None
        t2b = time.perf_counter()
        S = sim.cosine(item, k=100, verbose=False, format_output=fmt)
        t3 = time.perf_counter()
        print(f"   python: prepare {1e3 * (t1 - t0):.1f} ms, run_hip {1e3 * (t2 - t1):.1f} ms; whole public call {1e3 * (t3 - t2b):.1f} ms (releasing the previous result: {1e3 * (t2b - t2):.1f} ms)", file=sys.stderr, flush=True)
