"""SIMILARIPY_AMD_RESERVE_CUS as a measured default: one N = 8 slice of configs[1] (125 k rows of the C2 matrix)
at 0 / 4 / 8 / 16 reserved CUs, alone and beside a device-to-device copy stream that stands in for the RCCL gather of the previous
sub-slab (175 MB per sub-slab at N = 8: four copies of that size are queued on a second stream in front of the slice's launch).
Reported: the slice's kernel time, and when the copies were done (a copy kernel needs a free CU: with every CU taken by the persistent
row kernel it only starts when workgroups retire).  python scripts/reserve_cus_probe.py > gpurun_out/reserve_cus.txt"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from similaripy_amd import _host, workloads                     # noqa: E402
from similaripy_amd.device import DeviceProblem                 # noqa: E402
from similaripy_amd.distributed import slice_call               # noqa: E402

m = workloads.fixed_degree_csr(1_000_000, 100_000, 64, 12345)
call = slice_call(_host.prepare(m, m.T.tocsr(), k=100, l2=1, c1=0.5, c2=0.5), 0, 125_000)
torch.cuda.set_device(0)
prob = DeviceProblem(call)
cols, vals, counts, _ = prob.alloc_outputs()
src = torch.empty(175_000_000 // 4, dtype=torch.int32, device="cuda")
dst = torch.empty_like(src)
side = torch.cuda.Stream()
print("reserved_cus  copy_stream  slice_ms(mean of 5)  copies_done_ms  (125 k rows of C2, phases as in one sub-launch)")
for reserve in (0, 4, 8, 16):
    os.environ["SIMILARIPY_AMD_RESERVE_CUS"] = str(reserve)
    for with_copy in (False, True):
        ks, cs = [], []
        for it in range(7):
            torch.cuda.synchronize()
            e0, e1, c1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            if with_copy:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(4):
                        dst.copy_(src, non_blocking=True)
                    c1.record()
            prob.run(cols, vals, counts)
            e1.record()
            torch.cuda.synchronize()
            if it >= 2:
                ks.append(e0.elapsed_time(e1))
                cs.append(e0.elapsed_time(c1) if with_copy else 0.0)
        print(f"{reserve:12d}  {'yes' if with_copy else 'no ':11s}  {np.mean(ks):8.3f}             {np.mean(cs):8.3f}")
