"""Golden-vector generator.  TEST INFRASTRUCTURE ONLY -- runs in the BUILD CONTAINER only.

Imports the real reference from /root/reference (never copied, never shipped to the GPU box)
and records inputs + outputs of its hot path as small .npz fixtures under tests/golden/.

  python oracle/gen_golden.py            # regenerates every fixture

How the reference is loaded:
  * models/gaussian_diffusion.py imports as-is (`sys.path.insert(0, "/root/reference")`).
  * models/DNN.py and evaluate_utils.py import modules absent from this image
    (torch_geometric, bottleneck) at module level, and lightGCN.py runs a training script at
    import.  We therefore `ast`-extract ONLY the hot-path definitions (class DNN +
    timestep_embedding; computeTopNAccuracy; class LightGCN) from the files where they lie
    and exec those definitions unmodified.  No stand-in libraries are created.
Fixtures are data only: inputs (weights, rows, timesteps, noise, dropout keep-masks, initial
Lt_history) and the reference's outputs.
"""
import ast
import math
import os
import sys

import numpy as np
import pandas as pd
import scipy.sparse as sp
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)


def _extract(path, names, ns):
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    assert {n.name for n in keep} == set(names), (path, names)
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


import models.gaussian_diffusion as gd  # noqa: E402  (the real reference)

_ns = dict(torch=torch, nn=nn, F=F, np=np, math=math)
RefDNN, ref_timestep_embedding = _extract(f"{REF}/models/DNN.py", ["DNN", "timestep_embedding"], _ns)
import copy  # noqa: E402
_ns["copy"] = copy
(RefDNNOneHot,) = _extract(f"{REF}/models/DNN.py", ["DNNOneHot"], _ns)
RefDNNOneHotEmbedding, _ref_nt_xent = _extract(f"{REF}/models/DNN.py", ["DNNOneHotEmbedding", "nt_xent_loss"], _ns)
(ref_topn,) = _extract(f"{REF}/evaluate_utils.py", ["computeTopNAccuracy"], dict(math=math, np=np, torch=torch))


def npy(t):
    return t.detach().cpu().numpy().copy()


def make_rows(B, I, density, gen):
    x = (torch.rand(B, I, generator=gen) < density).float()
    if B >= 4:
        x[1] = 0.0  # an empty user row
        x[2, : max(1, I // 3)] = 1.0  # a heavy user row
    return x


def sd_np(model, prefix="sd."):
    return {prefix + k: npy(v) for k, v in model.state_dict().items()}


# ----------------------------------------------------------------------------------------
def gen_schedules():
    out = {}
    combos = [
        ("linear-var", 0.01, 0.001, 0.01, 5),
        ("linear-var", 0.1, 0.001, 0.01, 100),
        ("linear-var", 0.005, 0.0005, 0.005, 40),
        ("linear", 0.1, 0.001, 0.01, 10),
        ("linear", 1.0, 0.0001, 0.02, 50),
        ("cosine", 1.0, 0.0, 0.0, 20),
        ("binomial", 1.0, 0.0, 0.0, 8),
    ]
    names = []
    for i, (sch, scale, mn, mx, T) in enumerate(combos):
        d = gd.GaussianDiffusion(gd.ModelMeanType.START_X, sch, scale, mn, mx, T, "cpu")
        key = f"c{i}"
        names.append(f"{key}|{sch}|{scale}|{mn}|{mx}|{T}")
        for tab in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next",
                    "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
                    "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                    "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
            out[f"{key}.{tab}"] = npy(getattr(d, tab))
        t = torch.arange(T)
        w = d.SNR(t - 1) - d.SNR(t)
        out[f"{key}.snr_weight_x0"] = npy(torch.where(t == 0, 1.0, w))
    out["combos"] = np.array(names)
    # timestep embedding + normal_kl known answers
    ts = torch.tensor([0, 1, 2, 3, 4, 7, 39, 99])
    for dim in (10, 7, 16):
        out[f"temb.{dim}"] = npy(ref_timestep_embedding(ts, dim))
    out["temb.ts"] = npy(ts)
    g = torch.Generator().manual_seed(5)
    m1, lv1, m2, lv2 = [torch.randn(6, 9, generator=g) for _ in range(4)]
    out["kl.m1"], out["kl.lv1"], out["kl.m2"], out["kl.lv2"] = map(npy, (m1, lv1, m2, lv2))
    out["kl.out"] = npy(gd.normal_kl(m1, lv1, m2, lv2))
    np.savez_compressed(os.path.join(OUT, "schedules.npz"), **out)


# ----------------------------------------------------------------------------------------
def gen_train(name, B, I, dims, T, mean_type, schedule="linear-var", scale=0.01, nmin=0.001, nmax=0.01,
              n_steps=3, lr=1e-3, wd=0.0, seed=0, density=0.1, importance=False, emb=10, norm=False):
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 100)
    model = RefDNN([I] + dims, dims[::-1] + [I], emb, time_type="cat", norm=norm)
    mt = {"x0": gd.ModelMeanType.START_X, "eps": gd.ModelMeanType.EPSILON}[mean_type]
    diff = gd.GaussianDiffusion(mt, schedule, scale, nmin, nmax, T, "cpu")
    opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=wd)
    out = dict(sd_np(model))
    out["meta"] = np.array([f"{B}|{I}|{','.join(map(str, dims))}|{T}|{mean_type}|{schedule}|{scale}|{nmin}|{nmax}|"
                            f"{n_steps}|{lr}|{wd}|{emb}|{int(norm)}"])
    if importance:  # pre-fill the ring buffer so that the importance branch is live
        diff.Lt_history = torch.rand(T, 10, generator=g, dtype=torch.float64) * 50 + 0.5
        diff.Lt_count = torch.full((T,), 10, dtype=torch.int64)
    out["Lt_history0"], out["Lt_count0"] = npy(diff.Lt_history), npy(diff.Lt_count)

    cap = {}
    orig_st, orig_q, orig_mn = diff.sample_timesteps, diff.q_sample, gd.th.multinomial

    def st(*a, **k):
        t, pt = orig_st(*a, **k)
        cap["ts"], cap["pt"] = t.clone(), pt.clone()
        return t, pt

    def q(x, t, noise=None):
        cap["noise"] = noise.clone()
        r = orig_q(x, t, noise)
        cap["x_t"] = r.clone()
        return r

    def mn(p, *a, **k):
        cap["p_all"] = p.clone()
        return orig_mn(p, *a, **k)

    diff.sample_timesteps, diff.q_sample = st, q
    gd.th.multinomial = mn
    model.drop.register_forward_hook(lambda m, i, o: cap.update(drop_in=i[0].clone(), drop_out=o.clone()))
    model.in_layers[-1].register_forward_hook(lambda m, i, o: cap.update(h_pre=o.clone()))
    model.register_forward_hook(lambda m, i, o: cap.update(model_output=o.clone()))

    model.train()
    for s in range(n_steps):
        x = make_rows(B, I, density, g)
        opt.zero_grad()
        terms = diff.training_losses(model, x, True)
        loss = terms["loss"].mean()
        loss.backward()
        mask = (cap["drop_out"] != 0)
        # the keep-mask is only ambiguous where the dropout input is exactly 0 (never with noise on)
        assert (cap["drop_in"] != 0).all()
        assert torch.equal(cap["drop_out"], cap["drop_in"] * mask.float() * 2.0)
        p = f"s{s}."
        out[p + "x_start"] = npy(x).astype(np.uint8)
        out[p + "ts"], out[p + "pt"] = npy(cap["ts"]), npy(cap["pt"])
        out[p + "noise"] = npy(cap["noise"])
        out[p + "drop_mask"] = npy(mask).astype(np.uint8)
        out[p + "x_t"] = npy(cap["x_t"])
        out[p + "h"] = npy(torch.tanh(cap["h_pre"]))
        out[p + "model_output"] = npy(cap["model_output"])
        out[p + "loss_vec"] = npy(terms["loss"])
        out[p + "loss"] = npy(loss)
        if "p_all" in cap:
            out[p + "p_all"] = npy(cap.pop("p_all"))
        if s == 0:
            for k, v in model.named_parameters():
                out["g0." + k] = npy(v.grad)
        opt.step()
        out[p + "Lt_history"], out[p + "Lt_count"] = npy(diff.Lt_history), npy(diff.Lt_count)
        if s == n_steps - 1:
            for k, v in model.named_parameters():
                out["pN." + k] = npy(v)
    for k, v in model.named_parameters():
        st_ = opt.state[v]
        out["m." + k], out["v." + k] = npy(st_["exp_avg"]), npy(st_["exp_avg_sq"])
    gd.th.multinomial = orig_mn
    np.savez_compressed(os.path.join(OUT, f"train_{name}.npz"), **out)
    ts_all = np.concatenate([out[f"s{s}.ts"] for s in range(n_steps)])
    print(f"train_{name}: loss0={float(out['s0.loss']):.6g} has_t0={bool((ts_all == 0).any())}")


# ----------------------------------------------------------------------------------------
def gen_sample(name, B, I, dims, T, mean_type, seed=0, density=0.08, scale=0.01, nmin=0.001, nmax=0.01, k=20, norm=False):
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 7)
    model = RefDNN([I] + dims, dims[::-1] + [I], 10, time_type="cat", norm=norm)
    # spread the outputs so that top-k gaps are far above fp32 summation-order noise
    with torch.no_grad():
        model.out_layers[-1].bias.normal_(0.0, 0.5, generator=g)
    model.eval()
    mt = {"x0": gd.ModelMeanType.START_X, "eps": gd.ModelMeanType.EPSILON}[mean_type]
    diff = gd.GaussianDiffusion(mt, "linear-var", scale, nmin, nmax, T, "cpu")
    x = make_rows(B, I, density, g)
    out = dict(sd_np(model))
    out["meta"] = np.array([f"{B}|{I}|{','.join(map(str, dims))}|{T}|{mean_type}|{scale}|{nmin}|{nmax}|{k}" + ("|1" if norm else "")])
    out["x_start"] = npy(x).astype(np.uint8)
    cap = {"noises": []}
    orig_randn = gd.th.randn_like

    def rl(t):
        n = orig_randn(t)
        cap["noises"].append(n.clone())
        return n

    gd.th.randn_like = rl
    with torch.no_grad():
        # (a) steps=0, deterministic (what the shipped YAML uses)
        out["pred_steps0"] = npy(diff.p_sample(model, x, 0, False))
        # (b) steps=T: q_sample at t=T-1 first (one randn_like), deterministic reverse loop
        cap["noises"].clear()
        out["pred_stepsT"] = npy(diff.p_sample(model, x, T, False))
        out["noise_stepsT"] = npy(cap["noises"][0])
        assert len(cap["noises"]) == 1
        # (c) steps=2 with sampling noise: 1 + T draws
        cap["noises"].clear()
        out["pred_noisy"] = npy(diff.p_sample(model, x, 2, True))
        assert len(cap["noises"]) == 1 + T
        out["noise_noisy0"] = npy(cap["noises"][0])
        out["noise_noisy_steps"] = np.stack([npy(n) for n in cap["noises"][1:]])
        # per-step pred_xstart / mean of case (a)
        x_t = x
        preds, means = [], []
        for i in list(range(T))[::-1]:
            o = diff.p_mean_variance(model, x_t, torch.tensor([i] * B))
            preds.append(npy(o["pred_xstart"]))
            means.append(npy(o["mean"]))
            x_t = o["mean"]
        out["step_pred_xstart"], out["step_mean"] = np.stack(preds), np.stack(means)
    gd.th.randn_like = orig_randn

    # masked top-k exactly as reference main.py:296-301 (history = the training rows themselves)
    pred = torch.from_numpy(out["pred_steps0"]).clone()
    his = sp.csr_matrix(npy(x))
    pred[his.nonzero()] = -np.inf
    vals, idx = torch.topk(pred, k + 1)
    out["topk_idx"] = npy(idx[:, :k])
    out["topk_gap"] = npy(vals[:, k - 1] - vals[:, k])
    out["topk_min_adjacent_gap"] = npy((vals[:, :-1] - vals[:, 1:]).min(dim=1).values)
    # ground truth for the metric: random held-out items per user
    gt = []
    for b in range(B):
        n = int(torch.randint(0, 6, (1,), generator=g))
        gt.append(sorted(set(torch.randint(0, I, (n,), generator=g).tolist())))
    topN = [1, 5, 10, k]
    res = ref_topn(gt, idx[:, :k].tolist(), topN)
    out["gt_flat"] = np.array([i for r in gt for i in r], dtype=np.int64)
    out["gt_ptr"] = np.cumsum([0] + [len(r) for r in gt]).astype(np.int64)
    out["topN"] = np.array(topN)
    out["metrics"] = np.array(res, dtype=np.float64)  # rows: precision, recall, NDCG, MRR
    np.savez_compressed(os.path.join(OUT, f"sample_{name}.npz"), **out)
    print(f"sample_{name}: min top-k gap {out['topk_gap'].min():.3g}, min adjacent {out['topk_min_adjacent_gap'].min():.3g}")


# ----------------------------------------------------------------------------------------
# one-hot variant (first slice): GaussianDiffusionDiscrete(CatOneHot=True), indexIn False, DNNOneHot
class _Args:  # the reference reads only args.user_guided (:720)
    user_guided = False


class _Index:  # the reference calls index.cuda() (:744) and, with indexIn False, never uses the result
    def cuda(self):
        return self


def _onehot_pair(I, dims, T, mean_type, schedule, scale, nmin, nmax, norm, emb=10):
    import contextlib
    import io
    out_dims = dims[::-1] + [I]
    in_dims = [I] + dims
    with contextlib.redirect_stdout(io.StringIO()):
        model = RefDNNOneHot(in_dims, out_dims, emb, time_type="cat", norm=norm)
        mt = {"x0": gd.ModelMeanType.START_X, "eps": gd.ModelMeanType.EPSILON}[mean_type]
        diff = gd.GaussianDiffusionDiscrete(mt, schedule, scale, nmin, nmax, T, "cpu", discrete=0.99, CatOneHot=True,
                                            args=_Args())
    return model, diff


def gen_train_onehot(name, B, I, dims, T, mean_type, schedule="linear-var", scale=0.01, nmin=0.001, nmax=0.01, n_steps=2,
                     lr=1e-3, wd=0.0, seed=0, density=0.1, norm=False):
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 100)
    model, diff = _onehot_pair(I, dims, T, mean_type, schedule, scale, nmin, nmax, norm)
    opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=wd)
    out = dict(sd_np(model))
    out["meta"] = np.array([f"{B}|{I}|{','.join(map(str, dims))}|{T}|{mean_type}|{schedule}|{scale}|{nmin}|{nmax}|"
                            f"{n_steps}|{lr}|{wd}|10|{int(norm)}|0.99"])
    cap = {"st": [], "drops": []}
    orig_st, orig_q, orig_sd = diff.sample_timesteps, diff.q_sample, diff.sample_discrete_features

    def st(*a, **k):
        cap["depth"] = cap.get("depth", 0) + 1  # 'importance' falls back to a nested 'uniform' call until the history is full
        t, pt = orig_st(*a, **k)
        cap["depth"] -= 1
        if cap["depth"] == 0:
            cap["st"].append((t.clone(), pt.clone()))
        return t, pt

    def q(x, t, noise=None):
        cap["noise"] = noise.clone()
        r = orig_q(x, t, noise)
        cap["x_t"] = r.clone()
        return r

    def sd(probX):
        cap["probX"] = probX.clone()
        r = orig_sd(probX)
        cap["sampled"] = r.clone()
        return r

    diff.sample_timesteps, diff.q_sample, diff.sample_discrete_features = st, q, sd
    model.drop.register_forward_hook(lambda m, i, o: cap["drops"].append((i[0].clone(), o.clone())))
    model.register_forward_hook(lambda m, i, o: cap.update(model_output=o.clone(), x_tU=i[2].clone()))
    model.train()
    for s in range(n_steps):
        x = make_rows(B, I, density, g)
        cap["st"].clear()
        cap["drops"].clear()
        opt.zero_grad()
        terms = diff.training_losses(model, x, True)
        loss = terms["loss"].mean()
        loss.backward()
        assert len(cap["st"]) == 2 and len(cap["drops"]) == 2
        (din, dout), (din_u, dout_u) = cap["drops"]
        mask, mask_u = (dout != 0), (dout_u != 0)
        assert torch.equal(dout, din * mask.float() * 2.0) and torch.equal(dout_u, din_u * mask_u.float() * 2.0)
        p = f"s{s}."
        out[p + "x_start"] = npy(x).astype(np.uint8)
        out[p + "ts_U"] = npy(cap["st"][0][0])
        out[p + "ts"], out[p + "pt"] = npy(cap["st"][1][0]), npy(cap["st"][1][1])
        out[p + "sampled"] = npy(cap["sampled"]).astype(np.uint8)
        out[p + "prob1"] = npy(cap["probX"][..., 1])
        out[p + "x_tU"] = npy(cap["x_tU"]).astype(np.uint8)
        out[p + "noise"] = npy(cap["noise"])
        out[p + "drop_mask"] = npy(mask).astype(np.uint8)
        out[p + "drop_mask_U"] = npy(mask_u).astype(np.uint8)  # [B, 2I]; 0 where the input bit was 0 (irrelevant there)
        out[p + "x_t"] = npy(cap["x_t"])
        out[p + "model_output"] = npy(cap["model_output"])
        out[p + "loss_vec"] = npy(terms["loss"])
        out[p + "loss"] = npy(loss)
        if s == 0:
            for k, v in model.named_parameters():
                out["g0." + k] = npy(v.grad)
        opt.step()
        out[p + "Lt_history"], out[p + "Lt_count"] = npy(diff.Lt_history), npy(diff.Lt_count)
    for k, v in model.named_parameters():
        out["pN." + k] = npy(v)
        out["m." + k], out["v." + k] = npy(opt.state[v]["exp_avg"]), npy(opt.state[v]["exp_avg_sq"])
    np.savez_compressed(os.path.join(OUT, f"onehot_train_{name}.npz"), **out)
    print(f"onehot_train_{name}: loss0={float(out['s0.loss']):.6g} kept bits {int(out['s0.x_tU'].sum())}")


def gen_sample_onehot(name, B, I, dims, T, mean_type, seed=0, density=0.08, scale=0.01, nmin=0.001, nmax=0.01):
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 7)
    model, diff = _onehot_pair(I, dims, T, mean_type, "linear-var", scale, nmin, nmax, False)
    with torch.no_grad():
        model.out_layers[-1].bias.normal_(0.0, 0.5, generator=g)
    model.eval()
    x = make_rows(B, I, density, g)
    out = dict(sd_np(model))
    out["meta"] = np.array([f"{B}|{I}|{','.join(map(str, dims))}|{T}|{mean_type}|{scale}|{nmin}|{nmax}|0.99"])
    out["x_start"] = npy(x).astype(np.uint8)
    cap = {"noises": [], "sampled": []}
    orig_randn, orig_sd = gd.th.randn_like, diff.sample_discrete_features

    def rl(t):
        n = orig_randn(t)
        cap["noises"].append(n.clone())
        return n

    def sd(probX):
        r = orig_sd(probX)
        cap["sampled"].append(r.clone())
        return r

    gd.th.randn_like, diff.sample_discrete_features = rl, sd
    with torch.no_grad():
        out["pred_steps0"] = npy(diff.p_sample(model, x, 0, False, index=_Index()))
        cap["noises"].clear()
        cap["sampled"].clear()
        out["pred_stepsT"] = npy(diff.p_sample(model, x, T, False, index=_Index()))
        out["noise_stepsT"], out["sampled_stepsT"] = npy(cap["noises"][0]), npy(cap["sampled"][0]).astype(np.uint8)
        assert len(cap["noises"]) == 1 and len(cap["sampled"]) == 1 + T  # + one graph draw per reverse step
        cap["noises"].clear()
        cap["sampled"].clear()
        out["pred_noisy"] = npy(diff.p_sample(model, x, 2, True, index=_Index()))
        assert len(cap["noises"]) == 1 + T
        out["noise_noisy0"], out["sampled_noisy0"] = npy(cap["noises"][0]), npy(cap["sampled"][0]).astype(np.uint8)
        out["noise_noisy_steps"] = np.stack([npy(n) for n in cap["noises"][1:]])
    gd.th.randn_like = orig_randn
    np.savez_compressed(os.path.join(OUT, f"onehot_sample_{name}.npz"), **out)
    print(f"onehot_sample_{name}: |pred0| {np.abs(out['pred_steps0']).mean():.3g}")


# ----------------------------------------------------------------------------------------
def gen_lightgcn(name, U, It, d, L, nnz, seed=0):
    rng = np.random.default_rng(seed)
    users = rng.integers(0, U, nnz)
    items = np.minimum((rng.pareto(1.2, nnz) * It / 20).astype(np.int64), It - 1)  # skewed popularity
    # every node gets at least one edge except the last user / item (zero-degree edge case)
    users = np.concatenate([users, np.arange(U - 1)])
    items = np.concatenate([items, rng.integers(0, It - 1, U - 1)])
    items = np.where(items == It - 1, 0, items)
    df = pd.DataFrame({"user_id_idx": users, "item_id_idx": items})
    ns = dict(torch=torch, nn=nn, sp=sp, np=np, n_users=U, n_items=It)
    (RefLightGCN,) = _extract(f"{REF}/lightGCN.py", ["LightGCN"], ns)
    torch.manual_seed(seed)
    m = RefLightGCN(df, U, It, L, d)
    A = m.norm_adj_mat_sparse_tensor.coalesce()
    with torch.no_grad():
        fu, fi, iu, ii = m.propagate_through_layers()
        E = m.E0.weight
        layers = []
        for _ in range(L):
            E = torch.sparse.mm(m.norm_adj_mat_sparse_tensor, E)
            layers.append(npy(E))
    out = dict(users=users, items=items, meta=np.array([f"{U}|{It}|{d}|{L}"]), E0=npy(m.E0.weight),
               A_row=npy(A.indices()[0]), A_col=npy(A.indices()[1]), A_val=npy(A.values()),
               layers=np.stack(layers), final_user=npy(fu), final_item=npy(fi))
    np.savez_compressed(os.path.join(OUT, f"lightgcn_{name}.npz"), **out)
    print(f"lightgcn_{name}: nnz(A)={A.values().numel()}")


def gen_metrics():
    """Known-answer vectors for computeTopNAccuracy incl. the survey's hand case."""
    out = {}
    gt = [[1, 2], [3], []]
    pred = [[1, 5, 2], [4, 3, 9], [0, 1, 2]]
    out["hand"] = np.array(ref_topn(gt, pred, [1, 3]), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "metrics_hand.npz"), **out)


def gen_data_load():
    """reference data_utils.data_load on tiny (uid, iid) lists incl. a duplicated pair; inputs + CSR outputs."""
    import contextlib
    import io
    import tempfile
    import data_utils as ref_du  # the real reference module
    rng = np.random.default_rng(3)
    tr = np.stack([rng.integers(0, 12, 60), rng.integers(0, 9, 60)], axis=1)
    tr = np.concatenate([tr, tr[:2]])  # duplicates -> value 2.0
    tr[0] = (11, 8)
    va = np.stack([rng.integers(0, 12, 10), rng.integers(0, 9, 10)], axis=1)
    te = np.stack([rng.integers(0, 12, 14), rng.integers(0, 9, 14)], axis=1)
    with tempfile.TemporaryDirectory() as d:
        paths = []
        for n, a in (("train", tr), ("valid", va), ("test", te)):
            paths.append(os.path.join(d, n + "_list.npy"))
            np.save(paths[-1], a)
        with contextlib.redirect_stdout(io.StringIO()):
            t, v, e, nu, ni = ref_du.data_load(*paths)
    np.savez_compressed(os.path.join(OUT, "data_load.npz"), train_list=tr, valid_list=va, test_list=te,
                        train=t.toarray(), valid=v.toarray(), test=e.toarray(), n_user=nu, n_item=ni)
    print("data_load:", nu, ni, t.max())


class _IndexT:  # training_losses / p_sample call index.cuda() (:888, :744); the model then does index.to(x.device)
    def __init__(self, t):
        self.t = t

    def cuda(self):
        return self.t


def gen_train_onehot_emb(name, B, I, U, dims, T, mean_type, schedule="linear-var", scale=0.01, nmin=0.001, nmax=0.01, n_steps=2,
                         lr=1e-3, wd=0.0, seed=0, density=0.1):
    """indexIn backbone DNNOneHotEmbedding (main.py:239-242) under GaussianDiffusionDiscrete(CatOneHot=True)."""
    import contextlib
    import io
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 100)
    with contextlib.redirect_stdout(io.StringIO()):
        model = RefDNNOneHotEmbedding([I] + dims, dims[::-1] + [I], 10, time_type="cat", norm=False, item_num=I, user_num=U)
        mt = {"x0": gd.ModelMeanType.START_X, "eps": gd.ModelMeanType.EPSILON}[mean_type]
        diff = gd.GaussianDiffusionDiscrete(mt, schedule, scale, nmin, nmax, T, "cpu", discrete=0.99, CatOneHot=True,
                                            args=_Args())
    diff.indexIn = True
    opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=wd)
    out = dict(sd_np(model))
    out["meta"] = np.array([f"{B}|{I}|{','.join(map(str, dims))}|{T}|{mean_type}|{schedule}|{scale}|{nmin}|{nmax}|"
                            f"{n_steps}|{lr}|{wd}|10|0|0.99|{U}"])
    cap = {"st": [], "drops": []}
    orig_st, orig_q, orig_sd = diff.sample_timesteps, diff.q_sample, diff.sample_discrete_features

    def st(*a, **k):
        cap["depth"] = cap.get("depth", 0) + 1
        t, pt = orig_st(*a, **k)
        cap["depth"] -= 1
        if cap["depth"] == 0:
            cap["st"].append((t.clone(), pt.clone()))
        return t, pt

    def q(x, t, noise=None):
        cap["noise"] = noise.clone()
        r = orig_q(x, t, noise)
        cap["x_t"] = r.clone()
        return r

    def sd(probX):
        r = orig_sd(probX)
        cap["sampled"] = r.clone()
        return r

    diff.sample_timesteps, diff.q_sample, diff.sample_discrete_features = st, q, sd
    model.drop.register_forward_hook(lambda m, i, o: cap["drops"].append((i[0].clone(), o.clone())))
    model.register_forward_hook(lambda m, i, o: cap.update(model_output=o[0].clone(), closs=o[1].clone()))
    model.train()
    for s in range(n_steps):
        x = make_rows(B, I, density, g)
        index = torch.randperm(U, generator=g)[:B]
        cap["st"].clear()
        cap["drops"].clear()
        opt.zero_grad()
        terms = diff.training_losses(model, x, True, index=_IndexT(index))
        loss = terms["loss"].mean()
        loss.backward()
        assert len(cap["st"]) == 2 and len(cap["drops"]) == 2
        (din, dout), (din_u, dout_u) = cap["drops"]
        mask, mask_u = (dout != 0), (dout_u != 0)
        p = f"s{s}."
        out[p + "x_start"] = npy(x).astype(np.uint8)
        out[p + "index"] = npy(index)
        out[p + "ts_U"] = npy(cap["st"][0][0])
        out[p + "ts"], out[p + "pt"] = npy(cap["st"][1][0]), npy(cap["st"][1][1])
        out[p + "sampled"] = npy(cap["sampled"]).astype(np.uint8)
        out[p + "noise"] = npy(cap["noise"])
        out[p + "drop_mask"] = npy(mask).astype(np.uint8)
        out[p + "drop_mask_U"] = npy(mask_u).astype(np.uint8)
        out[p + "x_t"] = npy(cap["x_t"])
        out[p + "model_output"] = npy(cap["model_output"])
        out[p + "closs"] = npy(cap["closs"])
        out[p + "loss_vec"] = npy(terms["loss"])
        out[p + "loss"] = npy(loss)
        if s == 0:
            for k, v in model.named_parameters():
                if v.grad is not None:  # out_layers take no part in this backbone's forward
                    out["g0." + k] = npy(v.grad)
        opt.step()
        out[p + "Lt_history"], out[p + "Lt_count"] = npy(diff.Lt_history), npy(diff.Lt_count)
    for k, v in model.named_parameters():
        out["pN." + k] = npy(v)
        if v in opt.state and "exp_avg" in opt.state[v]:
            out["m." + k], out["v." + k] = npy(opt.state[v]["exp_avg"]), npy(opt.state[v]["exp_avg_sq"])
    # evaluation path on the trained weights: steps = 0 and steps = T with injected noise / classes
    diff.q_sample = orig_q
    model.eval()
    x = make_rows(B, I, density, g)
    index = torch.randperm(U, generator=g)[:B]
    cap2 = {"noises": [], "sampled": []}
    orig_randn = gd.th.randn_like

    def rl(t):
        n = orig_randn(t)
        cap2["noises"].append(n.clone())
        return n

    def sd2(probX):
        r = orig_sd(probX)
        cap2["sampled"].append(r.clone())
        return r

    gd.th.randn_like, diff.sample_discrete_features = rl, sd2
    with torch.no_grad():
        out["e.x_start"], out["e.index"] = npy(x).astype(np.uint8), npy(index)
        out["e.pred_steps0"] = npy(diff.p_sample(model, x, 0, False, index=_IndexT(index)))
        cap2["noises"].clear()
        cap2["sampled"].clear()
        out["e.pred_stepsT"] = npy(diff.p_sample(model, x, T, False, index=_IndexT(index)))
        out["e.noise_stepsT"], out["e.sampled_stepsT"] = npy(cap2["noises"][0]), npy(cap2["sampled"][0]).astype(np.uint8)
    gd.th.randn_like = orig_randn
    np.savez_compressed(os.path.join(OUT, f"onehot_emb_{name}.npz"), **out)
    print(f"onehot_emb_{name}: loss0={float(out['s0.loss']):.6g} closs0={float(out['s0.closs']):.4g}")


def gen_onehot():
    gen_train_onehot("tiny_x0", 8, 64, [16], 5, "x0", seed=31)
    gen_train_onehot("ragged_eps_wd", 12, 131, [24], 5, "eps", seed=32, density=0.05, wd=0.01, schedule="linear", scale=0.1)
    gen_train_onehot("deep_x0", 10, 90, [32, 16], 5, "x0", seed=33, density=0.06)
    gen_train_onehot("norm_eps", 9, 77, [20], 6, "eps", seed=34, density=0.08, norm=True, schedule="cosine", scale=0.05)
    gen_train_onehot_emb("tiny_x0", 8, 64, 40, [16], 5, "x0", seed=51)
    gen_train_onehot_emb("ragged_eps_wd", 12, 131, 90, [24], 5, "eps", seed=52, density=0.05, wd=0.01, schedule="linear", scale=0.1)
    gen_sample_onehot("tiny_x0", 8, 64, [16], 5, "x0", seed=41)
    gen_sample_onehot("ragged_eps", 10, 131, [24], 5, "eps", seed=42, scale=50.0)


def gen_bpr(seed=7):
    """`bpr_loss` of the reference's LightGCN script (lightGCN.py:207-219, AST-extracted: the module itself trains at
    import): losses and the gradients w.r.t. every input on random embeddings, two batch sizes."""
    (ref_bpr,) = _extract(f"{REF}/lightGCN.py", ["bpr_loss"], dict(torch=torch))
    g = torch.Generator().manual_seed(seed)
    out = {}
    for tag, (B, d) in (("a", (64, 16)), ("b", (257, 64))):
        ins = [torch.randn(B, d, generator=g, requires_grad=True) for _ in range(6)]
        users = torch.arange(B)
        mf, reg = ref_bpr(users, *ins)
        (mf + 1e-4 * reg).backward()
        out[f"{tag}.mf"], out[f"{tag}.reg"] = npy(mf), npy(reg)
        for k, t in zip(("users_emb", "pos_emb", "neg_emb", "userEmb0", "posEmb0", "negEmb0"), ins):
            out[f"{tag}.{k}"] = npy(t)
            out[f"{tag}.g_{k}"] = npy(t.grad)
    np.savez_compressed(os.path.join(OUT, "bpr_loss.npz"), **out)
    print("bpr_loss:", float(out["a.mf"]), float(out["b.reg"]))


class _ArgsGuided:
    user_guided = True


def gen_graph_guided(name, B, I, T, user_guided, seed, density=0.1, scale=0.01):
    """Degree-guided graph of GaussianDiffusionDiscrete.p_sample (:706-744): the real reverse loop on the CPU with
    indexIn = True (so that `graph=` reaches the model, :1073) around a stand-in denoiser that records it.  Captured per
    reverse step: the classes drawn by apply_noise on the accumulated graph, the one bit per user drawn from its degree,
    the transition probabilities, and the graph handed to the model."""
    import contextlib
    import io
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 100)
    with contextlib.redirect_stdout(io.StringIO()):
        diff = gd.GaussianDiffusionDiscrete(gd.ModelMeanType.START_X, "linear-var", scale, 0.001, 0.01, T, "cpu", discrete=0.99,
                                            CatOneHot=True, args=_ArgsGuided() if user_guided else _Args())
    diff.indexIn = True
    x = make_rows(B, I, density, g)
    x[0] = 0.0  # a user without interactions (degree probability 0) ...
    x[1, : I // 2] = 1.0  # ... and the one with the largest degree (probability 1)
    cap = dict(sampled=[], probX=[], pick=[], graph=[])
    orig_sd = diff.sample_discrete_features

    def sd(probX):
        r = orig_sd(probX)
        cap["sampled"].append(r.clone())
        cap["probX"].append(probX[..., 1].clone())
        return r

    diff.sample_discrete_features = sd
    orig_mn = torch.Tensor.multinomial

    def mn(self, *a, **k):
        r = orig_mn(self, *a, **k)
        if self.dim() == 2 and self.shape == (B, 2):
            cap["pick"].append(r.clone())
        return r

    def model(x_t, t, x_tU, index=None, graph=None):
        cap["graph"].append(graph.argmax(dim=2).clone())
        return x_t * 0.5

    torch.Tensor.multinomial = mn
    try:
        pred = diff.p_sample(model, x, 0, False, index=_Index())
    finally:
        torch.Tensor.multinomial = orig_mn
    assert len(cap["graph"]) == T and len(cap["pick"]) == T and len(cap["sampled"]) == T
    out = dict(meta=np.array([f"{B}|{I}|{T}|{int(user_guided)}|{scale}|0.99|{seed}"]), x_start=npy(x).astype(np.uint8),
               sampled=np.stack([npy(t) for t in cap["sampled"]]).astype(np.uint8),
               pick=np.stack([npy(t)[:, 0] for t in cap["pick"]]).astype(np.uint8),
               prob1=np.stack([npy(t) for t in cap["probX"]]).astype(np.float32),
               graph=np.stack([npy(t) for t in cap["graph"]]).astype(np.uint8), pred=npy(pred))
    np.savez_compressed(os.path.join(OUT, f"graph_guided_{name}.npz"), **out)
    print(f"graph_guided_{name}: edges per step {out['graph'].reshape(T, -1).sum(1).tolist()}, picks {out['pick'].sum(1).tolist()}")


def gen_round2():
    gen_bpr()
    gen_graph_guided("plain", 12, 97, 5, False, seed=61)
    gen_graph_guided("guided", 12, 97, 5, True, seed=62)
    gen_graph_guided("guided_T9", 20, 64, 9, True, seed=63, density=0.2)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ["onehot"]:  # only the fixtures of the one-hot variant
        gen_onehot()
        sys.exit(0)
    if sys.argv[1:] == ["round2"]:  # bpr_loss + the degree-guided graph of the reverse loop
        gen_round2()
        sys.exit(0)
    gen_data_load()
    gen_schedules()
    gen_metrics()
    gen_train("tiny_x0", 8, 64, [16], 5, "x0", seed=1)
    gen_train("tiny_eps", 8, 64, [16], 5, "eps", seed=3)
    gen_train("ragged_x0", 32, 515, [100], 5, "x0", seed=2, density=0.03)
    gen_train("ragged_eps_wd", 24, 515, [100], 5, "eps", seed=4, density=0.03, wd=0.01, schedule="linear", scale=0.1, n_steps=2)
    gen_train("imp_T40", 16, 600, [64], 40, "x0", seed=5, density=0.02, importance=True, scale=0.005,
              nmin=0.0005, nmax=0.005)
    gen_train("deep_x0", 16, 200, [48, 24], 5, "x0", seed=6, density=0.05, n_steps=2)
    gen_train("norm_x0", 8, 96, [32], 5, "x0", seed=8, norm=True, n_steps=1)
    gen_train("cosine_eps", 12, 130, [24], 8, "eps", seed=21, density=0.06, schedule="cosine", scale=0.05, n_steps=2)
    gen_train("binomial_x0", 12, 130, [24], 6, "x0", seed=22, density=0.06, schedule="binomial", n_steps=2)
    gen_train("deep_eps_norm", 10, 150, [40, 20], 5, "eps", seed=23, density=0.08, norm=True, wd=0.02, n_steps=2)
    gen_sample("tiny_x0", 8, 64, [16], 5, "x0", seed=11, k=10)
    gen_sample("ragged_x0", 32, 515, [100], 5, "x0", seed=12, k=20)
    gen_sample("ragged_eps", 16, 515, [100], 5, "eps", seed=13, k=20, scale=50.0)
    gen_sample("norm_x0", 12, 150, [32], 5, "x0", seed=14, k=10, norm=True)
    gen_lightgcn("small", 50, 40, 8, 3, 300, seed=0)
    gen_lightgcn("mid", 600, 400, 64, 3, 6000, seed=1)
    gen_onehot()
    gen_round2()
