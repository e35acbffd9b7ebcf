import sys; sys.path.insert(0,'.')
import numpy as np, scipy.sparse as sp
from similaripy_amd import _host, _abi, workloads
from similaripy_amd.device import DeviceProblem
import similaripy_amd as sim, torch
urm = workloads.fixed_degree_csr(1_000_000, 100_000, 64, 12345)
W = sim.cosine(urm[:200000].T.tocsr(), k=100, verbose=False, format_output="csr")
call = _host.prepare(urm, W.T.tocsr(), k=100, filter_cols=urm)
prob = DeviceProblem(call)
c,v,n,_ = prob.alloc_outputs()
prob.run(c,v,n); torch.cuda.synchronize()
print("configs[4] slice workspace bytes:", prob._ws.numel()/1e9, "GB")
m = workloads.fixed_degree_csr(1_000_000, 100_000, 64, 12345)
call = _host.prepare(m, m.T.tocsr(), k=100, l2=1)
prob = DeviceProblem(call); c,v,n,_ = prob.alloc_outputs(); prob.run(c,v,n); torch.cuda.synchronize()
print("configs[1] workspace bytes:", prob._ws.numel()/1e9, "GB")
