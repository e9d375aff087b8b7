"""CPU: pin the oracle (oracle/gdmcf_oracle.py) against outputs of the REAL reference
(tests/golden/*.npz, produced by oracle/gen_golden.py in the build container)."""
import numpy as np
import pytest
import torch

from oracle import gdmcf_oracle as O
from tests import helpers as H


def test_schedule_tables_match_reference():
    fx = H.load("schedules")
    for c in fx["combos"]:
        key, sch, scale, mn, mx, T = str(c).split("|")
        d = O.GaussianDiffusion(O.ModelMeanType.START_X, sch, float(scale), float(mn), float(mx), int(T))
        for tab in O.TABLE_NAMES:
            np.testing.assert_array_equal(getattr(d, tab).numpy(), fx[f"{key}.{tab}"], err_msg=f"{c} {tab}")
        t = torch.arange(int(T))
        w = torch.where(t == 0, 1.0, d.SNR(t - 1) - d.SNR(t))
        np.testing.assert_array_equal(w.numpy(), fx[f"{key}.snr_weight_x0"])


def test_survey_pinned_constants():
    # SURVEY.md section 8(a) rows a3/a4/a10 (captured from the reference during the survey)
    d = O.GaussianDiffusion(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, 5)
    np.testing.assert_allclose(d.betas.numpy(), [1e-05, 2.2500225002275442e-05, 2.2500731273855656e-05,
                                                 2.2501237567973398e-05, 2.2501743885183778e-05], rtol=0, atol=0)
    np.testing.assert_allclose(d.posterior_mean_coef1.numpy(), [1.000000000004551, 0.6923111538730923,
                               0.4090975569816447, 0.29033056484445835, 0.22500871925716165], rtol=1e-15)
    np.testing.assert_allclose(d.posterior_mean_coef2.numpy(), [0.0, 0.30768884609844477, 0.590902442927905,
                               0.7096694350006049, 0.7749912805248048], rtol=1e-15)


def test_timestep_embedding_and_kl():
    fx = H.load("schedules")
    ts = torch.from_numpy(fx["temb.ts"])
    for dim in (10, 7, 16):
        np.testing.assert_array_equal(O.timestep_embedding(ts, dim).numpy(), fx[f"temb.{dim}"])
    kl = O.normal_kl(*[torch.from_numpy(fx[f"kl.{k}"]) for k in ("m1", "lv1", "m2", "lv2")])
    np.testing.assert_array_equal(kl.numpy(), fx["kl.out"])


@pytest.mark.parametrize("case", H.TRAIN_CASES)
def test_train_steps_match_reference(case):
    """Same weights + same (ts, pt, noise, keep-mask) -> bit-identical loss, grads, AdamW state, history."""
    fx = H.load("train_" + case)
    meta = H.train_meta(fx)
    torch.manual_seed(0)
    model = H.oracle_model(meta, fx)
    diff = H.oracle_diffusion(meta)
    diff.Lt_history = torch.from_numpy(fx["Lt_history0"].copy())
    diff.Lt_count = torch.from_numpy(fx["Lt_count0"].copy())
    opt = O.make_optimizer(model, meta["lr"], meta["wd"])
    model.train()
    for s in range(meta["n_steps"]):
        inp = H.step_inputs(fx, s)
        if f"s{s}.p_all" in fx:
            np.testing.assert_array_equal(diff.importance_probs().numpy(), fx[f"s{s}.p_all"])
        cap = {}
        opt.zero_grad()
        terms = diff.training_losses(model, inp["x"], True, ts=inp["ts"], pt=inp["pt"], noise=inp["noise"],
                                     drop_mask=inp["drop_mask"], capture=cap)
        loss = terms["loss"].mean()
        loss.backward()
        np.testing.assert_array_equal(cap["x_t"].numpy(), fx[f"s{s}.x_t"])
        np.testing.assert_array_equal(cap["model_output"].detach().numpy(), fx[f"s{s}.model_output"])
        np.testing.assert_array_equal(terms["loss"].detach().numpy(), fx[f"s{s}.loss_vec"])
        assert terms["loss"].dtype == torch.float64
        np.testing.assert_array_equal(loss.detach().numpy(), fx[f"s{s}.loss"])
        if s == 0:
            for k, v in model.named_parameters():
                np.testing.assert_array_equal(v.grad.numpy(), fx["g0." + k], err_msg=k)
        opt.step()
        np.testing.assert_array_equal(diff.Lt_history.numpy(), fx[f"s{s}.Lt_history"])
        np.testing.assert_array_equal(diff.Lt_count.numpy(), fx[f"s{s}.Lt_count"])
    for k, v in model.named_parameters():
        np.testing.assert_array_equal(v.detach().numpy(), fx["pN." + k], err_msg=k)
        np.testing.assert_array_equal(opt.state[v]["exp_avg"].numpy(), fx["m." + k])
        np.testing.assert_array_equal(opt.state[v]["exp_avg_sq"].numpy(), fx["v." + k])


def test_oracle_rng_call_order_matches_reference():
    """With randomness NOT injected the oracle must consume torch's generator in the reference's order;
    replaying the fixture's seed reproduces the reference's loss bit for bit."""
    fx = H.load("train_tiny_x0")
    meta = H.train_meta(fx)
    torch.manual_seed(1)  # seed used by gen_golden for this case
    model = O.DNN([meta["I"]] + meta["dims"], meta["dims"][::-1] + [meta["I"]], 10)
    for k, v in model.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), fx["sd." + k], err_msg=f"init draw order: {k}")


@pytest.mark.parametrize("case", H.SAMPLE_CASES)
def test_p_sample_and_topk_match_reference(case):
    fx = H.load("sample_" + case)
    meta = H.sample_meta(fx)
    model = H.oracle_model(meta, fx).eval()
    diff = H.oracle_diffusion(meta)
    x = torch.from_numpy(fx["x_start"].astype(np.float32))
    T = meta["T"]
    with torch.no_grad():
        cap = {}
        p0 = diff.p_sample(model, x, 0, False, capture=cap)
        np.testing.assert_array_equal(p0.numpy(), fx["pred_steps0"])
        np.testing.assert_array_equal(torch.stack(cap["pred_xstart"]).numpy(), fx["step_pred_xstart"])
        np.testing.assert_array_equal(torch.stack(cap["mean"]).numpy(), fx["step_mean"])
        pT = diff.p_sample(model, x, T, False, noise0=torch.from_numpy(fx["noise_stepsT"]))
        np.testing.assert_array_equal(pT.numpy(), fx["pred_stepsT"])
        pn = diff.p_sample(model, x, 2, True, noise0=torch.from_numpy(fx["noise_noisy0"]),
                           step_noise=torch.from_numpy(fx["noise_noisy_steps"]))
        np.testing.assert_array_equal(pn.numpy(), fx["pred_noisy"])
    rows, cols = np.nonzero(fx["x_start"])
    idx = O.masked_topk(p0, torch.from_numpy(rows), torch.from_numpy(cols), meta["k"])
    np.testing.assert_array_equal(idx.numpy(), fx["topk_idx"])  # fixtures have no exact ties
    gt = [fx["gt_flat"][a:b].tolist() for a, b in zip(fx["gt_ptr"][:-1], fx["gt_ptr"][1:])]
    res = O.computeTopNAccuracy(gt, idx.tolist(), fx["topN"].tolist())
    np.testing.assert_array_equal(np.array(res, dtype=np.float64), fx["metrics"])


def test_metric_hand_case():
    fx = H.load("metrics_hand")
    res = O.computeTopNAccuracy([[1, 2], [3], []], [[1, 5, 2], [4, 3, 9], [0, 1, 2]], [1, 3])
    np.testing.assert_array_equal(np.array(res), fx["hand"])
    assert res == ([0.3333, 0.3333], [0.1667, 0.6667], [0.3333, 0.5169], [0.3333, 0.5])


@pytest.mark.parametrize("case", ["small", "mid"])
def test_lightgcn_matches_reference(case):
    fx = H.load("lightgcn_" + case)
    U, It, d, L = [int(v) for v in str(fx["meta"][0]).split("|")]
    A = O.lightgcn_norm_adj(fx["users"], fx["items"], U, It)
    coo = A.tocoo()
    order = np.lexsort((coo.col, coo.row))
    np.testing.assert_array_equal(coo.row[order], fx["A_row"])
    np.testing.assert_array_equal(coo.col[order], fx["A_col"])
    np.testing.assert_allclose(coo.data[order], fx["A_val"], rtol=2e-7, atol=0)
    fu, fi, iu, ii, layers = O.lightgcn_propagate(A, fx["E0"], L, U)
    for l in range(L):
        np.testing.assert_allclose(layers[l + 1], fx["layers"][l], rtol=0, atol=2e-7)
    np.testing.assert_allclose(fu, fx["final_user"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(fi, fx["final_item"], rtol=0, atol=2e-7)


# ---- one-hot / discrete-noise variant (first slice) ---------------------------------------------------
@pytest.mark.parametrize("case", H.ONEHOT_TRAIN_CASES)
def test_onehot_train_steps_match_reference(case):
    """GaussianDiffusionDiscrete(CatOneHot=True) + DNNOneHot: same weights and the same randomness (both timestep
    draws, the sampled classes, noise, both keep-masks) -> bit-identical transition probabilities, kept bits, model
    output, loss vector, gradients, AdamW state and history."""
    fx = H.load("onehot_train_" + case)
    meta = H.onehot_train_meta(fx)
    model, diff = H.oracle_onehot_pair(meta, fx)
    opt = O.make_optimizer(model, meta["lr"], meta["wd"])
    model.train()
    for s in range(meta["n_steps"]):
        inp = H.onehot_step_inputs(fx, s)
        onehot = torch.nn.functional.one_hot(inp["x"].long(), num_classes=2).float()
        _, probX = diff.apply_noise(inp["ts_U"], onehot, inp["sampled"])
        np.testing.assert_array_equal(probX[..., 1].numpy(), fx[f"s{s}.prob1"])
        cap = {}
        opt.zero_grad()
        terms = diff.training_losses(model, inp["x"], True, ts=inp["ts"], pt=inp["pt"], noise=inp["noise"],
                                     drop_mask=inp["drop_mask"], capture=cap, ts_U=inp["ts_U"], sampled=inp["sampled"],
                                     drop_mask_U=inp["drop_mask_U"])
        loss = terms["loss"].mean()
        loss.backward()
        np.testing.assert_array_equal(cap["x_tU"].numpy().astype(np.uint8), fx[f"s{s}.x_tU"])
        np.testing.assert_array_equal(cap["x_t"].numpy(), fx[f"s{s}.x_t"])
        np.testing.assert_array_equal(cap["model_output"].detach().numpy(), fx[f"s{s}.model_output"])
        np.testing.assert_array_equal(terms["loss"].detach().numpy(), fx[f"s{s}.loss_vec"])
        np.testing.assert_array_equal(loss.detach().numpy(), fx[f"s{s}.loss"])
        if s == 0:
            for k, v in model.named_parameters():
                np.testing.assert_array_equal(v.grad.numpy(), fx["g0." + k], err_msg=k)
        opt.step()
        np.testing.assert_array_equal(diff.Lt_history.numpy(), fx[f"s{s}.Lt_history"])
        np.testing.assert_array_equal(diff.Lt_count.numpy(), fx[f"s{s}.Lt_count"])
    for k, v in model.named_parameters():
        np.testing.assert_array_equal(v.detach().numpy(), fx["pN." + k], err_msg=k)
        np.testing.assert_array_equal(opt.state[v]["exp_avg"].numpy(), fx["m." + k])
        np.testing.assert_array_equal(opt.state[v]["exp_avg_sq"].numpy(), fx["v." + k])


@pytest.mark.parametrize("case", H.ONEHOT_SAMPLE_CASES)
def test_onehot_p_sample_matches_reference(case):
    fx = H.load("onehot_sample_" + case)
    meta = H.onehot_sample_meta(fx)
    model, diff = H.oracle_onehot_pair(meta, fx)
    model.eval()
    x = torch.from_numpy(fx["x_start"].astype(np.float32))
    T = meta["T"]
    with torch.no_grad():
        np.testing.assert_array_equal(diff.p_sample(model, x, 0, False, sampled0=torch.zeros(1)).numpy(), fx["pred_steps0"])
        got = diff.p_sample(model, x, T, False, noise0=torch.from_numpy(fx["noise_stepsT"]),
                            sampled0=torch.from_numpy(fx["sampled_stepsT"].astype(np.int64)))
        np.testing.assert_array_equal(got.numpy(), fx["pred_stepsT"])
        got = diff.p_sample(model, x, 2, True, noise0=torch.from_numpy(fx["noise_noisy0"]),
                            sampled0=torch.from_numpy(fx["sampled_noisy0"].astype(np.int64)),
                            step_noise=[torch.from_numpy(n) for n in fx["noise_noisy_steps"]])
        np.testing.assert_array_equal(got.numpy(), fx["pred_noisy"])


def test_onehot_rng_call_order_matches_reference():
    """Without injected randomness the oracle draws from torch's generator in the reference's order (timesteps, class
    draw, timesteps, randn_like, dropout x, dropout x_U): seeding like the generator did reproduces its first step."""
    fx = H.load("onehot_train_tiny_x0")
    meta = H.onehot_train_meta(fx)
    model, diff = H.oracle_onehot_pair(meta, fx)
    model.train()
    torch.manual_seed(31)
    O.DNNOneHot([meta["I"]] + meta["dims"], meta["dims"][::-1] + [meta["I"]], 10)  # consumes the init draws like the generator
    x = torch.from_numpy(fx["s0.x_start"].astype(np.float32))
    terms = diff.training_losses(model, x, True)
    np.testing.assert_array_equal(terms["loss"].detach().numpy(), fx["s0.loss_vec"])


@pytest.mark.parametrize("case", H.ONEHOT_EMB_CASES)
def test_onehot_embedding_backbone_matches_reference(case):
    """indexIn backbone DNNOneHotEmbedding (user / item embedding tables, cosine scores, NT-Xent term x 0.1) under
    GaussianDiffusionDiscrete(CatOneHot=True): training steps and p_sample bit-identical to the reference."""
    fx = H.load("onehot_emb_" + case)
    meta = H.onehot_emb_meta(fx)
    model, diff = H.oracle_onehot_emb_pair(meta, fx)
    opt = O.make_optimizer(model, meta["lr"], meta["wd"])
    model.train()
    for s in range(meta["n_steps"]):
        inp = H.onehot_step_inputs(fx, s)
        cap = {}
        opt.zero_grad()
        terms = diff.training_losses(model, inp["x"], True, ts=inp["ts"], pt=inp["pt"], noise=inp["noise"],
                                     drop_mask=inp["drop_mask"], capture=cap, ts_U=inp["ts_U"], sampled=inp["sampled"],
                                     drop_mask_U=inp["drop_mask_U"], index=torch.from_numpy(fx[f"s{s}.index"]))
        loss = terms["loss"].mean()
        loss.backward()
        np.testing.assert_array_equal(cap["model_output"].detach().numpy(), fx[f"s{s}.model_output"])
        np.testing.assert_array_equal(cap["closs"].detach().numpy(), fx[f"s{s}.closs"])
        np.testing.assert_array_equal(terms["loss"].detach().numpy(), fx[f"s{s}.loss_vec"])
        if s == 0:
            for k, v in model.named_parameters():
                if v.grad is None:
                    assert k.startswith("out_layers") and "g0." + k not in fx
                else:
                    np.testing.assert_array_equal(v.grad.numpy(), fx["g0." + k], err_msg=k)
        opt.step()
        np.testing.assert_array_equal(diff.Lt_history.numpy(), fx[f"s{s}.Lt_history"])
    for k, v in model.named_parameters():
        np.testing.assert_array_equal(v.detach().numpy(), fx["pN." + k], err_msg=k)
    model.eval()
    x, idx = torch.from_numpy(fx["e.x_start"].astype(np.float32)), torch.from_numpy(fx["e.index"])
    with torch.no_grad():
        np.testing.assert_array_equal(diff.p_sample(model, x, 0, False, sampled0=torch.zeros(1), index=idx).numpy(),
                                      fx["e.pred_steps0"])
        got = diff.p_sample(model, x, meta["T"], False, noise0=torch.from_numpy(fx["e.noise_stepsT"]),
                            sampled0=torch.from_numpy(fx["e.sampled_stepsT"].astype(np.int64)), index=idx)
        np.testing.assert_array_equal(got.numpy(), fx["e.pred_stepsT"])


def test_gcn_backbone_restatement_ignores_the_graph_on_user_rows():
    """DNNOneHotEmbeddingGCN (parity UNPINNED: torch_geometric is not vendored).  Property of the restated GCNConv that the
    HIP path relies on: user nodes are never the target of an edge, so the rows of the GCN output that the backbone uses
    equal a plain 2-layer perceptron of hc -- scores and gradients do not depend on `graph`; and gcn_conv itself equals
    the dense D^-1/2 (A + I)^T-aggregation formula on a small random directed graph."""
    torch.manual_seed(0)
    n, cin, cout = 9, 5, 4
    x, w, b = torch.randn(n, cin), torch.randn(cout, cin), torch.randn(cout)
    ei = torch.tensor([[0, 0, 1, 2, 2, 7], [4, 5, 5, 6, 8, 3]])
    A = torch.zeros(n, n)
    A[ei[1], ei[0]] = 1.0  # A[i, j] = 1 for an edge j -> i
    A = A + torch.eye(n)
    dis = A.sum(1).pow(-0.5)
    ref = (dis[:, None] * A * dis[None, :]) @ (x @ w.t()) + b
    np.testing.assert_allclose(O.gcn_conv(x, ei, w, b).numpy(), ref.numpy(), rtol=1e-6, atol=1e-6)

    I, hid, B, U, T = 40, 6, 7, 30, 5
    m = O.DNNOneHotEmbeddingGCN([I, hid], [hid, I], 10, item_num=I, user_num=U, hidden_dim=8)
    with torch.no_grad():
        m.sumW.fill_(0.3)
        m.gcn_model.conv1.bias.normal_()
        m.gcn_model.conv2.bias.normal_()
    m.eval()
    xr = (torch.rand(B, I) < 0.2).float()
    ts, idx = torch.randint(0, T, (B,)), torch.randperm(U)[:B]
    xU = torch.nn.functional.one_hot(xr.long(), 2).float()
    g1 = torch.nn.functional.one_hot(xr.long(), 2)
    g2 = torch.nn.functional.one_hot((torch.rand(B, I) < 0.5).long(), 2)
    g0 = torch.nn.functional.one_hot(torch.zeros(B, I, dtype=torch.long), 2)
    outs = [m(xr, ts, xU, index=idx, graph=g) for g in (g1, g2, g0)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # ... and they are the perceptron blend
    with torch.no_grad():
        ref_m = O.DNNOneHotEmbedding([I, hid], [hid, I], 10, item_num=I, user_num=U)
        ref_m.load_state_dict({k: v for k, v in m.state_dict().items() if not k.startswith(("gcn_model", "sumW"))})
        ref_m.eval()
        emb = m.emb_layer(O.timestep_embedding(ts, 10))
        h = torch.tanh(m.in_layers[0](torch.cat([xr, emb], 1)))
        hU = torch.tanh(m.in_layers2[0](torch.cat([xU.reshape(B, -1), emb], 1)))
        hc = torch.cat([h, hU, m.embedding_user(idx)], 1)
        z = torch.relu(hc @ m.gcn_model.conv1.lin.weight.t() + m.gcn_model.conv1.bias)
        z = z @ m.gcn_model.conv2.lin.weight.t() + m.gcn_model.conv2.bias
        u = hc * 0.3 + z * 0.7
        V = m.embedding_item.weight
        want = (u @ V.t()) / (u.norm(dim=1, keepdim=True) * V.norm(dim=1))
    np.testing.assert_allclose(outs[0].detach().numpy(), want.numpy(), rtol=1e-5, atol=1e-6)


def test_bpr_loss_matches_reference():
    """bpr_loss (reference lightGCN.py:207-219, AST-extracted by oracle/gen_golden.py): the oracle's restatement AND the
    product's (gdmcf_amd.lightgcn.bpr_loss -- plain torch expressions around the HIP propagation) reproduce the
    reference's losses and input gradients bit for bit."""
    from gdmcf_amd.lightgcn import bpr_loss as product_bpr
    fx = H.load("bpr_loss")
    names = ("users_emb", "pos_emb", "neg_emb", "userEmb0", "posEmb0", "negEmb0")
    for tag in ("a", "b"):
        for fn in (O.bpr_loss, product_bpr):
            ins = [torch.from_numpy(fx[f"{tag}.{k}"]).clone().requires_grad_(True) for k in names]
            mf, reg = fn(torch.arange(ins[0].shape[0]), *ins)
            (mf + 1e-4 * reg).backward()
            np.testing.assert_array_equal(mf.detach().numpy(), fx[f"{tag}.mf"])
            np.testing.assert_array_equal(reg.detach().numpy(), fx[f"{tag}.reg"])
            for k, t in zip(names, ins):
                np.testing.assert_array_equal(t.grad.numpy(), fx[f"{tag}.g_{k}"], err_msg=f"{tag} {k}")


@pytest.mark.parametrize("case", ["plain", "guided", "guided_T9"])
def test_degree_guided_graph_of_the_reverse_loop_matches_reference(case):
    """The per-step degree-guided graph of GaussianDiffusionDiscrete.p_sample (reference :706-744): under the same seed the
    oracle draws the same classes and user bits and hands the same accumulated graph to the model at every reverse step."""
    fx = H.load("graph_guided_" + case)
    B, I, T, guided, scale, disc, seed = str(fx["meta"][0]).split("|")
    B, I, T, guided, seed = int(B), int(I), int(T), bool(int(guided)), int(seed)
    torch.manual_seed(seed)
    od = O.GaussianDiffusionDiscrete(O.ModelMeanType.START_X, "linear-var", float(scale), 0.001, 0.01, T, discrete=float(disc),
                                     CatOneHot=True, user_guided=guided)
    od.indexIn = True
    seen = []

    def model(x_t, t, x_tU, index=None, graph=None):
        seen.append(graph.argmax(dim=2).clone())
        return x_t * 0.5

    cap = {}
    pred = od.p_sample(model, torch.from_numpy(fx["x_start"].astype(np.float32)), 0, False, capture=cap)
    assert len(seen) == T
    np.testing.assert_array_equal(np.stack([g.numpy() for g in seen]), fx["graph"])
    np.testing.assert_array_equal(np.stack([g.numpy() for g in cap["pick"]]), fx["pick"])
    np.testing.assert_array_equal(pred.numpy(), fx["pred"])
    # the graph only ever grows, and with user guidance only in rows whose user bit was drawn
    g = fx["graph"].astype(np.int64)
    assert (np.diff(g, axis=0) >= 0).all()
    if guided:
        new = np.diff(np.concatenate([np.zeros_like(g[:1]), g]), axis=0)
        assert not new[fx["pick"] == 0].any()
