import sys, os, numpy as np, torch, cProfile, pstats, time
sys.path.insert(0, os.getcwd())
import gdmcf_amd, scipy.sparse as sp
from gdmcf_amd import data
from gdmcf_amd.data_utils import DeviceCSR
from gdmcf_amd.parallel import DataParallelStep
indptr, indices, I = data.synth_csr("yelp", n_rows=1600, seed=0)
dcsr = DeviceCSR(sp.csr_matrix((np.ones(len(indices), np.float32), indices, indptr), shape=(1600, I)), "cuda:0")
torch.manual_seed(0)
model = gdmcf_amd.DNN([I, 1000], [1000, I], 10).to("cuda:0").train()
d = gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, 5, "cuda:0")
opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=1e-5)
step = DataParallelStep(d, model, opt)
ids = [torch.arange(i * 400, (i + 1) * 400, device="cuda:0") for i in range(4)]
for i in range(20): step(dcsr.batch(ids[i % 4]), True)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter()
for i in range(300): step(dcsr.batch(ids[i % 4]), True)
el = time.perf_counter() - t
pr.disable(); torch.cuda.synchronize()
print("host enqueue ms/step (under cProfile)", el / 300 * 1e3)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
