"""Developer tool (not part of the product or the tests); run on a GPU box from the repo root."""
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
B = 12.0*be.nnz + 4*(be.n_dof_owned+1) + 16*be.n_dof_owned
Bact = 8.0*be.nnz + 4*be.n_pairs + 4*(be.n_dof_owned+1) + 4*(be.n_nodes_owned+1) + 16*be.n_dof_owned
print(os.environ.get("KNP_SPMV","node"), kind, N, "n", be.n_dof_owned, "nnz", be.nnz, f"{us:.1f} us  CSR-bytes GB/s {B/us/1e3:.0f}  node-format GB/s {Bact/us/1e3:.0f}", "checksum", float(y[:be.n_dof_owned].sum()))
