import sys, time, os
sys.path.insert(0, '.')
import numpy as np
import similaripy_amd as sim
from similaripy_amd import workloads
m = workloads.fixed_degree_csr(1_000_000, 100_000, 64, 12345)
Wt = sim.cosine(m[:200_000].T.tocsr(), k=100, verbose=False, format_output="csr").T.tocsr()
f = lambda: sim.dot_product(m, Wt, k=100, filter_cols=m, verbose=False, format_output="csr")
f(); f()
os.environ["SIMILARIPY_AMD_TRACE"] = "1"
t=time.perf_counter(); r=f(); print("total", time.perf_counter()-t, file=sys.stderr)
