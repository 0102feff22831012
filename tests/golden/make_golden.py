"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself on CPU.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

Each .npz holds weights (every state_dict entry, prefixed ``w::``), inputs and the reference's
outputs, so nothing depends on RNG-stream equality between torch versions.  Only *data* is stored:
no reference source, no checkpoint bytes (G4 stores the SHA-256 of the shipped checkpoint and the
inputs/outputs; the weights are re-read from /root/reference when the test runs here).

The reference imports ``torch_geometric.data.Data`` (content_providers.py:1, annotation only) and
``seaborn`` (plots.py:3); neither is installed, so two empty in-memory stub modules are registered
before the import.  GraphNCF cannot be imported (needs torch_geometric.nn) -> no golden for it
(parity unpinned; see oracle/ncf_oracle.py header).

Round 3 adds the deterministic train-mode pins:
  g3_att_train_*   AttentionNCF in .train() with dropout_rate=0.0 / message_dropout=None: target masking (:195-205)
  g7_grads_*       loss = MSELoss(reduction='sum') (datasets/base.py:19-20,31-32), .backward(): every parameter gradient
  g8_create_graph_*  content_providers/graph_providers.py:10-66 create_graph, binary and weighted
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(REF, "src"))
    tg = types.ModuleType("torch_geometric")
    tgd = types.ModuleType("torch_geometric.data")

    class Data:  # attribute bag: the reference uses it as an annotation (content_providers.py:61) and, in create_graph
        # (graph_providers.py:58-66), as a keyword container — no PyG behaviour is involved in either
        def __init__(self, **kw):
            self.__dict__.update(kw)

    tgd.Data = Data
    tg.data = tgd
    sys.modules["torch_geometric"] = tg
    sys.modules["torch_geometric.data"] = tgd
    sys.modules["seaborn"] = types.ModuleType("seaborn")


def _state_arrays(model):
    return {"w::" + k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}


def _save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    _import_reference()
    import torch
    from neural_collaborative_filtering.models.basic_ncf import BasicNCF
    from neural_collaborative_filtering.models.mf import MF
    from neural_collaborative_filtering.models.attention_ncf import AttentionNCF
    from neural_collaborative_filtering.util import build_MLP_layers

    import pandas as pd
    torch.set_num_threads(1)

    def onehot(pos, n):
        x = torch.zeros((len(pos), n), dtype=torch.float32)
        x[torch.arange(len(pos)), torch.as_tensor(pos)] = 1.0
        return x

    # ---------------- G1: BasicNCF, one-hot and dense-profile inputs ----------------
    for tag, U, I, E, mlp, dr in (("g1_basic_onehot_small", 60, 40, 8, [16], 0.2),
                                  ("g1_basic_onehot_e32", 60, 40, 32, [256, 128], 0.2),
                                  ("g1_basic_onehot_nodrop", 60, 40, 8, [16, 8], None)):
        torch.manual_seed(101)
        m = BasicNCF(item_dim=I, user_dim=U, item_emb=E, user_emb=E, mlp_dense_layers=mlp, dropout_rate=dr).eval()
        rng = np.random.default_rng(7)
        up = rng.integers(0, U, 64)
        ip = rng.integers(0, I, 64)
        with torch.no_grad():
            out = m(onehot(up, U), onehot(ip, I))
        _save(tag, user_pos=up, item_pos=ip, out=out.numpy(),
              kwargs=np.array(json.dumps(m.kwargs)), **_state_arrays(m))

    # kernel-shaped cases: these dims hit the fused / small-batch / folded HIP instances directly
    for tag, E, mlp in (("g1_basic_onehot_e64", 64, [256, 128]), ("g1_basic_onehot_e128", 128, [256, 128]),
                        ("g1_basic_onehot_e64_h256", 64, [256])):
        torch.manual_seed(105)
        U, I = 600, 400
        m = BasicNCF(item_dim=I, user_dim=U, item_emb=E, user_emb=E, mlp_dense_layers=mlp).eval()
        rng = np.random.default_rng(12)
        up = rng.integers(0, U, 700)
        ip = rng.integers(0, I, 700)
        with torch.no_grad():
            out = m(onehot(up, U), onehot(ip, I))
        _save(tag, user_pos=up, item_pos=ip, out=out.numpy(), kwargs=np.array(json.dumps(m.kwargs)), **_state_arrays(m))

    torch.manual_seed(102)
    m = BasicNCF(item_dim=24, user_dim=24, item_emb=16, user_emb=8, mlp_dense_layers=[32, 16]).eval()
    Xu = torch.rand(48, 24)
    Xi = torch.rand(48, 24)
    with torch.no_grad():
        out = m(Xu, Xi)
    _save("g1_basic_dense_profiles", X_user=Xu.numpy(), X_item=Xi.numpy(), out=out.numpy(),
          kwargs=np.array(json.dumps(m.kwargs)), **_state_arrays(m))

    # ---------------- G2: MF ----------------
    torch.manual_seed(103)
    m = MF(item_dim=40, user_dim=60, item_emb=16, user_emb=16).eval()
    rng = np.random.default_rng(8)
    up = rng.integers(0, 60, 64)
    ip = rng.integers(0, 40, 64)
    with torch.no_grad():
        out = m(onehot(up, 60), onehot(ip, 40))
    _save("g2_mf_onehot", user_pos=up, item_pos=ip, out=out.numpy(),
          kwargs=np.array(json.dumps(m.kwargs)), **_state_arrays(m))

    # ---------------- G3: AttentionNCF variants ----------------
    def att_inputs(B, I, Fdim, seed):
        g = torch.Generator().manual_seed(seed)
        rated = torch.rand(I, Fdim, generator=g)
        cand = torch.rand(B, Fdim, generator=g)
        um = torch.zeros(B, I)
        mask = torch.rand(B, I, generator=g) < 0.5
        ratings = torch.randint(1, 11, (B, I), generator=g).float() * 0.5  # 0.5 .. 5.0
        mean = torch.rand(B, 1, generator=g) * 2 + 2.5
        um[mask] = (ratings - (mean + 2.5) / 2)[mask]
        um[1, :] = 0.0  # a user with no rated items: softmax over all -inf -> NaN -> 0 (:208-209)
        # a rated position whose normalised value is exactly 0.0 is dropped by `!= 0` (:158)
        um[2, 3] = 0.0
        um[0, 0] = 1.25
        return cand, rated, um

    for tag, kw in (("g3_att_dense8", dict(att_dense=8)),
                    ("g3_att_none", dict(att_dense=None)),
                    ("g3_att_cos", dict(use_cos_sim_instead=True))):
        torch.manual_seed(104)
        m = AttentionNCF(item_dim=20, item_emb=16, user_emb=16, mlp_dense_layers=[32, 16], **kw).eval()
        cand, rated, um = att_inputs(8, 12, 20, 9)
        with torch.no_grad():
            out, att = m(cand, rated, um, return_attention_weights=True)
        _save(tag, candidate_items=cand.numpy(), rated_items=rated.numpy(), user_matrix=um.numpy(),
              out=out.numpy(), att=att.numpy(), kwargs=np.array(json.dumps(m.kwargs)), **_state_arrays(m))

    # kernel-shaped attention cases (vector paths of the HIP kernel: A = 128, IE = UE = 64 / 128)
    for tag, Fdim, IE, A, B, I in (("g3_att_vec64", 50, 64, 128, 64, 200), ("g3_att_vec128", 40, 128, 128, 33, 129)):
        torch.manual_seed(106)
        m = AttentionNCF(item_dim=Fdim, item_emb=IE, user_emb=IE, att_dense=A, mlp_dense_layers=[256, 128]).eval()
        g = torch.Generator().manual_seed(13)
        rated = torch.rand(I, Fdim, generator=g)
        cand = torch.rand(B, Fdim, generator=g)
        um = torch.zeros(B, I)
        mask = torch.rand(B, I, generator=g) < 0.3
        um[mask] = (torch.randint(1, 11, (B, I), generator=g).float() * 0.5 - 2.9)[mask]
        um[0] = 0.0
        with torch.no_grad():
            out, att = m(cand, rated, um, return_attention_weights=True)
        _save(tag, candidate_items=cand.numpy(), rated_items=rated.numpy(), user_matrix=um.numpy(),
              out=out.numpy(), att=att.numpy(), kwargs=np.array(json.dumps(m.kwargs)), **_state_arrays(m))

    # ---------------- G3-train: AttentionNCF.train(), deterministic (dropout 0, no message dropout) ----------------
    # attention_ncf.py:195-205: in training a candidate that equals one of the rated rows (isclose on the embeddings) is
    # masked out of its own softmax row.  Candidates 0..n_self-1 ARE rated rows of the batch; the loss is the reference's
    # MSELoss(reduction='sum') against y (datasets/base.py:19-20,31-32) and every parameter gradient is stored.
    def att_train_case(tag, Fdim, IE, UE, B, I, n_self, seed, mlp, density=0.5, **kw):
        torch.manual_seed(seed)
        m = AttentionNCF(item_dim=Fdim, item_emb=IE, user_emb=UE, mlp_dense_layers=mlp, dropout_rate=0.0,
                         message_dropout=None, **kw).train()
        g = torch.Generator().manual_seed(seed + 1)
        rated = torch.rand(I, Fdim, generator=g)
        cand = torch.rand(B, Fdim, generator=g)
        self_cols = torch.randperm(I, generator=g)[:n_self]
        cand[:n_self] = rated[self_cols]                      # these candidates are rated items of the batch
        um = torch.zeros(B, I)
        mask = torch.rand(B, I, generator=g) < density
        um[mask] = (torch.randint(1, 11, (B, I), generator=g).float() * 0.5 - 2.9)[mask]
        for b in range(n_self):                               # the user HAS rated the candidate (that is what gets masked)
            um[b, self_cols[b]] = 1.35 if b % 2 == 0 else -0.65
        um[B - 1] = 0.0                                       # all-unrated row in train mode
        if n_self > 1:                                        # a row whose ONLY rated entry is the candidate itself -> all -inf -> 0
            um[1] = 0.0
            um[1, self_cols[1]] = 0.85
        y = torch.randint(1, 11, (B,), generator=g).float() * 0.5
        out, att = m(cand, rated, um, return_attention_weights=True)
        loss = torch.nn.MSELoss(reduction='sum')(out, y.view(-1, 1).float())
        loss.backward()
        grads = {"g::" + k: p.grad.detach().numpy().copy() for k, p in m.named_parameters()}
        assert all(np.isfinite(v).all() for v in grads.values())
        _save(tag, candidate_items=cand.numpy(), rated_items=rated.numpy(), user_matrix=um.numpy(), y=y.numpy(),
              self_cols=self_cols.numpy(), out=out.detach().numpy(), att=att.numpy(), loss=np.array(loss.item()),
              kwargs=np.array(json.dumps(m.kwargs)), **_state_arrays(m), **grads)

    att_train_case("g3_att_train_dense8", 20, 16, 16, 8, 12, 3, 201, [32, 16], att_dense=8)
    att_train_case("g3_att_train_cos", 20, 16, 16, 8, 12, 3, 202, [32, 16], use_cos_sim_instead=True)
    att_train_case("g3_att_train_vec64", 48, 64, 64, 40, 150, 9, 203, [256, 128], density=0.3, att_dense=128)
    att_train_case("g3_att_train_ue50", 24, 16, 50, 8, 12, 2, 204, [32], att_dense=8)   # user_emb not a multiple of 4

    # ---------------- G7: gradients of BasicNCF / MF under the reference's loss ----------------
    def grads_case(tag, m, up, ip, U, I, seed):
        g = torch.Generator().manual_seed(seed)
        y = torch.randint(1, 11, (len(up),), generator=g).float() * 0.5
        out = m(onehot(up, U), onehot(ip, I))
        loss = torch.nn.MSELoss(reduction='sum')(out, y.view(-1, 1).float())
        loss.backward()
        grads = {"g::" + k: p.grad.detach().numpy().copy() for k, p in m.named_parameters()}
        _save(tag, user_pos=up, item_pos=ip, y=y.numpy(), out=out.detach().numpy(), loss=np.array(loss.item()),
              kwargs=np.array(json.dumps(m.kwargs)), **_state_arrays(m), **grads)

    rng = np.random.default_rng(21)
    torch.manual_seed(301)
    U, I = 60, 40
    m = BasicNCF(item_dim=I, user_dim=U, item_emb=8, user_emb=8, mlp_dense_layers=[16, 8], dropout_rate=None).train()
    grads_case("g7_grads_basic_small", m, rng.integers(0, U, 96), rng.integers(0, I, 96), U, I, 31)   # duplicates in the batch
    torch.manual_seed(302)
    U, I = 300, 200
    m = BasicNCF(item_dim=I, user_dim=U, item_emb=64, user_emb=64, mlp_dense_layers=[256, 128], dropout_rate=None).train()
    grads_case("g7_grads_basic_e64", m, rng.integers(0, U, 500), rng.integers(0, I, 500), U, I, 32)
    torch.manual_seed(303)
    m = BasicNCF(item_dim=I, user_dim=U, item_emb=32, user_emb=32, mlp_dense_layers=[256], dropout_rate=None).train()
    grads_case("g7_grads_basic_h256", m, rng.integers(0, U, 130), rng.integers(0, I, 130), U, I, 33)
    torch.manual_seed(304)
    m = MF(item_dim=40, user_dim=60, item_emb=16, user_emb=16).train()
    grads_case("g7_grads_mf", m, rng.integers(0, 60, 96), rng.integers(0, 40, 96), 60, 40, 34)

    # ---------------- G8: create_graph (content_providers/graph_providers.py:10-66) ----------------
    from content_providers.graph_providers import create_graph
    rng = np.random.default_rng(41)
    all_users = np.arange(100, 145)
    all_items = np.arange(0, 100)
    n = 300
    users = rng.integers(100, 140, n)
    items = rng.integers(7, 30, n) * 3
    ratings = rng.integers(1, 11, n) * 0.5
    inter = pd.DataFrame({"userId": users, "movieId": items, "rating": ratings})
    item_to_node = {int(i): k for k, i in enumerate(sorted(all_items))}                     # graph_providers.py:77-80
    user_to_node = {int(u): len(all_items) + k for k, u in enumerate(sorted(all_users))}
    for binary in (False, True):
        gr = create_graph(inter, None, None, item_to_node, user_to_node, binary=binary)
        pos = gr.pos_df.reset_index()[["Id1", "Id2", "pos"]].to_numpy().astype(np.int64)
        extra = {}
        if not binary:
            extra = dict(user2item_edge_attr=gr.user2item_edge_attr.numpy(), item2user_edge_attr=gr.item2user_edge_attr.numpy())
        _save("g8_create_graph_" + ("binary" if binary else "weighted"), all_users=all_users, all_items=all_items,
              userId=users, movieId=items, rating=ratings,
              user2item_edge_index=gr.user2item_edge_index.numpy(), item2user_edge_index=gr.item2user_edge_index.numpy(),
              pos=pos, **extra)

    # ---------------- G4: shipped checkpoint (weights NOT copied) ----------------
    ck = os.path.join(REF, "models/runs/AttentionNCF_with_features_attNet128_3mil.pt")
    sha = hashlib.sha256(open(ck, "rb").read()).hexdigest()
    state, kwargs = torch.load(ck, map_location="cpu", weights_only=True)
    m = AttentionNCF(**kwargs).eval()
    m.load_state_dict(state)
    g = torch.Generator().manual_seed(10)
    Fdim = kwargs["item_dim"]
    rated = (torch.rand(16, Fdim, generator=g) < 0.02).float() + torch.rand(16, Fdim, generator=g) * 0.1
    cand = (torch.rand(4, Fdim, generator=g) < 0.02).float() + torch.rand(4, Fdim, generator=g) * 0.1
    um = torch.zeros(4, 16)
    mask = torch.rand(4, 16, generator=g) < 0.6
    um[mask] = (torch.randint(1, 11, (4, 16), generator=g).float() * 0.5 - 3.1)[mask]
    with torch.no_grad():
        out, att = m(cand, rated, um, return_attention_weights=True)
    _save("g4_att_shipped_ckpt", candidate_items=cand.numpy(), rated_items=rated.numpy(), user_matrix=um.numpy(),
          out=out.numpy(), att=att.numpy(), kwargs=np.array(json.dumps(kwargs)),
          ckpt_sha256=np.array(sha), ckpt_relpath=np.array("models/runs/AttentionNCF_with_features_attNet128_3mil.pt"))

    # ---------------- G5: eval_ranking (eval.py:25-75) ----------------
    import pandas as pd
    from neural_collaborative_filtering.eval import eval_ranking
    rng = np.random.default_rng(11)
    uid = np.repeat(np.arange(1, 7), 5)
    uid = np.concatenate([uid, [99]])  # a user with a single row (skipped, :38)
    mid = rng.integers(1, 100, len(uid))
    rating = rng.integers(1, 11, len(uid)) * 0.5
    rating[5:10] = 3.0  # one user whose ratings are all ties (ideal == worst, :57)
    pred = rating + rng.normal(0, 1.0, len(uid))
    pred[2] = pred[3]  # a tie in the predictions (sklearn tie-averaged DCG)
    df = pd.DataFrame({"userId": uid, "movieId": mid, "rating": rating, "prediction": pred})
    res = {}
    for k in (5, 10):
        nd, adj = eval_ranking(df, cutoff=k)
        res[k] = (nd, adj)
    _save("g5_eval_ranking", userId=uid, movieId=mid, rating=rating, prediction=pred,
          ndcg5=np.array(res[5]), ndcg10=np.array(res[10]))

    # ---------------- G6: build_MLP_layers key names ----------------
    keys = {}
    for name, dr in (("with_dropout", 0.2), ("no_dropout", None)):
        seq = build_MLP_layers(10, [8, 4], dropout_rate=dr)
        keys[name] = list(seq.state_dict().keys())
    with open(os.path.join(OUT, "g6_mlp_keys.json"), "w") as f:
        json.dump(keys, f, indent=1)

    # ---------------- cfg 1 (BASELINE.json configs[0]): ML-1M-scale BasicNCF ----------------
    U, I = 6040, 3706
    torch.manual_seed(0)
    m = BasicNCF(item_dim=I, user_dim=U, item_emb=32, user_emb=32, mlp_dense_layers=[256], dropout_rate=0.2).eval()
    rng = np.random.default_rng(42)
    n = 20000
    up = rng.integers(0, U, n)
    ip = rng.integers(0, I, n)
    rating = rng.integers(1, 11, n) * 0.5
    outs = []
    with torch.no_grad():
        for s in range(0, n, 512):  # val_batch_size = 512 (globals.py:36)
            outs.append(m(onehot(up[s:s + 512], U), onehot(ip[s:s + 512], I)))
    out = torch.cat(outs).numpy()
    mse = float(((out.reshape(-1).astype(np.float64) - rating) ** 2).sum() / n)
    df = pd.DataFrame({"userId": up + 1, "movieId": ip + 1, "rating": rating,
                       "prediction": out.reshape(-1).astype(np.float64)})
    nd = {k: eval_ranking(df, cutoff=k) for k in (5, 10, 20)}
    _save("cfg1_basic_ml1m", user_pos=up.astype(np.int32), item_pos=ip.astype(np.int32), rating=rating, out=out,
          mse=np.array(mse), ndcg5=np.array(nd[5]), ndcg10=np.array(nd[10]), ndcg20=np.array(nd[20]),
          kwargs=np.array(json.dumps(m.kwargs)), **_state_arrays(m))


if __name__ == "__main__":
    main()
