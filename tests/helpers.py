"""Shared helpers for the parity tests: fixture loading + rebuilding oracle objects from them."""
import os

import numpy as np
import torch

from oracle import gdmcf_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAIN_CASES = ["tiny_x0", "tiny_eps", "ragged_x0", "ragged_eps_wd", "imp_T40", "deep_x0", "norm_x0", "cosine_eps",
               "binomial_x0", "deep_eps_norm"]
SAMPLE_CASES = ["tiny_x0", "ragged_x0", "ragged_eps", "norm_x0"]
ONEHOT_TRAIN_CASES = ["tiny_x0", "ragged_eps_wd", "deep_x0", "norm_eps"]
ONEHOT_SAMPLE_CASES = ["tiny_x0", "ragged_eps"]
ONEHOT_EMB_CASES = ["tiny_x0", "ragged_eps_wd"]


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def train_meta(fx):
    B, I, dims, T, mt, sch, scale, nmin, nmax, n_steps, lr, wd, emb, norm = str(fx["meta"][0]).split("|")
    return dict(B=int(B), I=int(I), dims=[int(d) for d in dims.split(",")], T=int(T), mean_type=mt, schedule=sch,
                scale=float(scale), nmin=float(nmin), nmax=float(nmax), n_steps=int(n_steps), lr=float(lr),
                wd=float(wd), emb=int(emb), norm=bool(int(norm)))


def sample_meta(fx):
    f = str(fx["meta"][0]).split("|")
    B, I, dims, T, mt, scale, nmin, nmax, k = f[:9]
    return dict(B=int(B), I=int(I), dims=[int(d) for d in dims.split(",")], T=int(T), mean_type=mt,
                scale=float(scale), nmin=float(nmin), nmax=float(nmax), k=int(k), norm=len(f) > 9 and f[9] == "1")


def state_dict_from(fx, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in fx.items() if k.startswith(prefix)}


def oracle_model(meta, fx):
    I, dims = meta["I"], meta["dims"]
    m = O.DNN([I] + dims, dims[::-1] + [I], meta.get("emb", 10), norm=meta.get("norm", False))
    m.load_state_dict(state_dict_from(fx))
    return m


def oracle_diffusion(meta, schedule=None):
    mt = {"x0": O.ModelMeanType.START_X, "eps": O.ModelMeanType.EPSILON}[meta["mean_type"]]
    return O.GaussianDiffusion(mt, schedule or meta.get("schedule", "linear-var"), meta["scale"], meta["nmin"],
                               meta["nmax"], meta["T"])


def step_inputs(fx, s):
    p = f"s{s}."
    return dict(x=torch.from_numpy(fx[p + "x_start"].astype(np.float32)), ts=torch.from_numpy(fx[p + "ts"]),
                pt=torch.from_numpy(fx[p + "pt"]), noise=torch.from_numpy(fx[p + "noise"]),
                drop_mask=torch.from_numpy(fx[p + "drop_mask"].astype(np.float32)))


def onehot_train_meta(fx):
    f = str(fx["meta"][0]).split("|")
    meta = train_meta({"meta": np.array(["|".join(f[:14])])})
    meta["discrete"] = float(f[14])
    return meta


def onehot_sample_meta(fx):
    B, I, dims, T, mt, scale, nmin, nmax, disc = str(fx["meta"][0]).split("|")
    return dict(B=int(B), I=int(I), dims=[int(d) for d in dims.split(",")], T=int(T), mean_type=mt, scale=float(scale),
                nmin=float(nmin), nmax=float(nmax), discrete=float(disc), norm=False, schedule="linear-var")


def oracle_onehot_pair(meta, fx):
    I, dims = meta["I"], meta["dims"]
    m = O.DNNOneHot([I] + dims, dims[::-1] + [I], meta.get("emb", 10), norm=meta.get("norm", False))
    m.load_state_dict(state_dict_from(fx))
    mt = {"x0": O.ModelMeanType.START_X, "eps": O.ModelMeanType.EPSILON}[meta["mean_type"]]
    d = O.GaussianDiffusionDiscrete(mt, meta.get("schedule", "linear-var"), meta["scale"], meta["nmin"], meta["nmax"],
                                    meta["T"], discrete=meta["discrete"], CatOneHot=True)
    return m, d


def onehot_emb_meta(fx):
    f = str(fx["meta"][0]).split("|")
    meta = train_meta({"meta": np.array(["|".join(f[:14])])})
    meta["discrete"], meta["U"] = float(f[14]), int(f[15])
    return meta


def oracle_onehot_emb_pair(meta, fx):
    I, dims = meta["I"], meta["dims"]
    m = O.DNNOneHotEmbedding([I] + dims, dims[::-1] + [I], 10, item_num=I, user_num=meta["U"])
    m.load_state_dict(state_dict_from(fx))
    mt = {"x0": O.ModelMeanType.START_X, "eps": O.ModelMeanType.EPSILON}[meta["mean_type"]]
    d = O.GaussianDiffusionDiscrete(mt, meta["schedule"], meta["scale"], meta["nmin"], meta["nmax"], meta["T"],
                                    discrete=meta["discrete"], CatOneHot=True)
    d.indexIn = True
    return m, d


def onehot_step_inputs(fx, s):
    p = f"s{s}."
    d = step_inputs(fx, s)
    d.update(ts_U=torch.from_numpy(fx[p + "ts_U"]), sampled=torch.from_numpy(fx[p + "sampled"].astype(np.int64)),
             drop_mask_U=torch.from_numpy(fx[p + "drop_mask_U"].astype(np.float32)))
    return d


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
