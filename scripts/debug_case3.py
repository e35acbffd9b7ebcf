import numpy as np, scipy.sparse as sp, sys
sys.path.insert(0, '.')
from similaripy_amd import _host
from oracle import splus_oracle as so
# 3 rows x 4 cols; m2 = m.T; make n_cols big by padding rows
n = 40000
rows = np.array([0,0,1,1,2,2]); cols = np.array([0,1,0,1,0,2]); vals = np.array([1,2,3,4,5,6], dtype=np.float32)
m = sp.csr_array((vals,(rows,cols)), shape=(n, 8))
call = _host.prepare(m, k=5, target_rows=[0,1,2])
r = _host.run_hip(call, time_kernel=True)
print("hip  :", r[1][:15].reshape(3,5), r[2][:15].reshape(3,5), r[3])
w = so.run_kernel(call, "port")
print("port :", w[1].reshape(3,5), w[2].reshape(3,5))
print(r[4])
