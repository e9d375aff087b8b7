"""Runs the LightGCN propagation (3 layers) a few times -- the program to put under rocprofv3 for the SpMM kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gdmcf_amd
from gdmcf_amd import data
shape = sys.argv[1] if len(sys.argv) > 1 else "yelp"
cfg = data.SHAPES[shape]
indptr, indices, I = data.synth_csr(shape, seed=0)
U = cfg["n_users"]
users = np.repeat(np.arange(U), np.diff(indptr))
torch.manual_seed(0)
m = gdmcf_amd.LightGCN({"user_id_idx": users, "item_id_idx": indices}, U, I, 3, 64, device="cuda:0").to("cuda:0")
with torch.no_grad():
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
        m.propagate_through_layers()
torch.cuda.synchronize()
print("alg MB", m.algorithmic_bytes() / 1e6, "nnz", m.nnz)
