"""Developer tool (not part of the product or the tests); run on a GPU box from the repo root:  python tests/devtools/spmv_bench.py cube 64
(KNP_SPMV = lanes per node, KNP_SPMV_UNROLL = pairs in flight per lane, KNP_SPMV_MF=0 reads the stored time-invariant entries)"""
import sys, os, time; sys.path.insert(0,'tests'); import conftest
import torch, numpy as np
from parity_utils import *
kind, N = sys.argv[1], int(sys.argv[2])
p = make_problem(ci_config(N=N, steps=1, kind=kind)); be = p.create_backend(); be.assemble_matrix()
x = torch.randn(be.n_dof_local, dtype=torch.float64, device='cuda'); y = torch.empty_like(x)
for _ in range(5): be.spmv(x,y)
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): be.spmv(x,y)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1)*1e3/50
B = be.traffic_model()["spmv"]
print("G", os.environ.get("KNP_SPMV","auto"), "U", os.environ.get("KNP_SPMV_UNROLL","2"), "MF", os.environ.get("KNP_SPMV_MF","1"), kind, N, "n", be.n_dof_owned, f"{us:.1f} us  {B/1e6:.0f} MB  {B/us/1e3:.0f} GB/s", "checksum", float(y[:be.n_dof_owned].sum()))
