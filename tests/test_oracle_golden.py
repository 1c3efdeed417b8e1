"""CPU: the oracle restatement against golden vectors produced by executing the
reference's own files (tests/golden/make_golden.py) and against scipy."""

import ast

import numpy as np
import pytest
import torch

from oracle import tribe_ref, xt_encoder


def _load(golden_dir, name):
    return np.load(golden_dir / name, allow_pickle=False)


def test_g1_subject_layers(golden_dir):
    g = _load(golden_dir, "g1_subject_layers.npz")
    x, w, b = (torch.from_numpy(g[k]) for k in ("x", "w", "b"))
    subj = torch.from_numpy(g["subj"])
    y = tribe_ref.subject_layers_fwd(x, w, b, subj)
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=1e-5, atol=1e-6)
    y = tribe_ref.subject_layers_fwd(x, w, None, subj.flatten())
    np.testing.assert_allclose(y.numpy(), g["y_nobias"], rtol=1e-5, atol=1e-6)
    with pytest.raises(AssertionError):  # common.py:53-55
        tribe_ref.subject_layers_fwd(x, w, b, subj + 2)


G2_CASES = [(fa, la, v) for fa in ("cat", "sum") for la in ("cat", "mean") for v in ("tri", "one_none", "ndim3")]


@pytest.mark.parametrize("fa,la,variant", G2_CASES)
def test_g2_aggregate_features(golden_dir, fa, la, variant):
    g = _load(golden_dir, "g2_aggregate_features.npz")
    fdims = {
        "tri": {"text": (2, 24), "audio": (2, 8), "video": (2, 12)},
        "one_none": {"text": (2, 24), "audio": None, "video": (2, 12)},
        "ndim3": {"text": (1, 24), "audio": (1, 8), "video": (1, 12)},
    }[variant]
    dims = tribe_ref.EncoderDims(depth=0)  # projectors only; hidden stays 3072 as in model.py:61
    m = tribe_ref.FmriEncoderRef(fdims, 7, 3, 4, feature_aggregation=fa, layer_aggregation=la, dims=dims).eval()
    with torch.no_grad():
        tribe_ref.fill_params_(m, seed=3)
    data = tribe_ref.synthetic_batch(2, 6, fdims, 4, seed=11)
    if variant == "ndim3":
        data = {k: (v[:, 0] if v.ndim == 4 else v) for k, v in data.items()}
    key = f"{fa}_{la}_{variant}"
    if key + "_raises" in g:
        with pytest.raises(RuntimeError):
            m.aggregate_features(data)
        return
    with torch.no_grad():
        y = m.aggregate_features(data)
    np.testing.assert_allclose(y[..., ::8].numpy(), g[key], rtol=1e-5, atol=1e-6)
    sums = np.array([y.double().sum().item(), y.double().abs().sum().item()])
    np.testing.assert_allclose(sums, g[key + "_sum"], rtol=1e-6)


@pytest.mark.parametrize("subj_emb", [False, True])
def test_g3_full_forward(golden_dir, subj_emb):
    """Reference glue (model.py:113-174) around the restated encoder, full width 3072 x 8 layers."""
    g = _load(golden_dir, "g3_forward.npz")
    fdims = {"text": (2, 24), "audio": (2, 8), "video": (2, 12)}
    m = tribe_ref.FmriEncoderRef(fdims, 37, 5, 4, subject_embedding=subj_emb).eval()
    with torch.no_grad():
        tribe_ref.fill_params_(m, seed=5)
    data = tribe_ref.synthetic_batch(2, 14, fdims, 4, seed=17)
    data["subject_id"] = torch.tensor([[1], [3]])
    tag = "se" if subj_emb else "nose"
    with torch.no_grad():
        np.testing.assert_allclose(m(data).numpy(), g[f"pooled_{tag}"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(m(data, pool_outputs=False).numpy(), g[f"unpooled_{tag}"], rtol=1e-4, atol=1e-5)


def test_g4_pearson_loss(golden_dir):
    g = _load(golden_dir, "g4_pearson_loss.npz")
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    np.testing.assert_allclose(tribe_ref.pearson_loss(x, y, "mean").numpy(), g["mean"], rtol=1e-6)
    np.testing.assert_allclose(tribe_ref.pearson_loss(x, y, "sum").numpy(), g["sum"], rtol=1e-6)
    with pytest.raises(ValueError):
        tribe_ref.pearson_loss(x, y, "none")


def test_g5_info_nce(golden_dir):
    g = _load(golden_dir, "g5_info_nce.npz")
    q, k = torch.from_numpy(g["q"]), torch.from_numpy(g["k"])
    np.testing.assert_allclose(tribe_ref.info_nce(q, k, 0.07).numpy(), g["loss"], rtol=1e-5)
    np.testing.assert_allclose(tribe_ref.info_nce(q, k, 1.0).numpy(), g["loss_tau1"], rtol=1e-5)


def test_g6_run_step(golden_dir):
    g = _load(golden_dir, "g6_run_step.npz")
    loss, p, t, groups = tribe_ref.run_step(
        torch.from_numpy(g["y_pred"]), torch.from_numpy(g["y_true"]), torch.from_numpy(g["sid"]))
    np.testing.assert_allclose(loss.numpy(), g["loss"], rtol=1e-6)
    np.testing.assert_array_equal(p.numpy(), g["pred_flat"])
    np.testing.assert_array_equal(t.numpy(), g["true_flat"])
    np.testing.assert_array_equal(groups.numpy(), g["groups"])


def test_g8_aggregate_layers(golden_dir):
    g = _load(golden_dir, "g8_aggregate_layers.npz")
    layer_sets = ast.literal_eval(str(g["layer_sets_json"]))
    n = 0
    for cls in ("LLAMA3p2", "Wav2VecBert", "VJEPA2"):
        for n_states in (25, 29, 41):
            lat = np.arange(n_states * 3, dtype=np.float32).reshape(n_states, 3) ** 1.5
            for tag, layers in layer_sets.items():
                for agg in (None, "group_mean"):
                    want = g[f"{cls}_{n_states}_{tag}_{agg}"]
                    got = tribe_ref.aggregate_layers(lat, layers, agg)
                    assert got.shape == want.shape
                    np.testing.assert_allclose(got, want, rtol=1e-6)
                    n += 1
    assert n == 3 * 3 * 6 * 2


def test_g7_pearson_vs_scipy():
    """main.py:474-477 uses scipy.stats.pearsonr per parcel; the sufficient-statistics form
    (what the HIP kernel accumulates) and the restated streaming metric must agree with it."""
    g = torch.Generator().manual_seed(7)
    N, V = 1024, 40
    pred = torch.randn(N, V, generator=g)
    true = 0.3 * pred + torch.randn(N, V, generator=g)
    ref = tribe_ref.scipy_pearson_columns(pred.numpy(), true.numpy())
    r = tribe_ref.pearson_from_stats(tribe_ref.pearson_stats(pred, true), N)
    np.testing.assert_allclose(r.numpy(), ref, atol=2e-6)
    sp = tribe_ref.StreamingPearson(V)
    for chunk in range(0, N, 256):
        sp.update(pred[chunk:chunk + 256], true[chunk:chunk + 256])
    np.testing.assert_allclose(sp.compute_per_output().numpy(), ref, atol=1e-5)
    np.testing.assert_allclose(float(sp.compute()), ref.mean(), atol=1e-5)


@pytest.mark.parametrize("t_in,t_out", [(298, 100), (14, 5), (1024, 1024), (100, 7), (7, 7), (5, 3)])
def test_adaptive_pool_matches_torch(t_in, t_out):
    """model.py:60 uses torch's own nn.AdaptiveAvgPool1d: pin the restated window table on it."""
    x = torch.randn(2, 3, t_in)
    np.testing.assert_allclose(
        tribe_ref.adaptive_avg_pool1d(x, t_out).numpy(),
        torch.nn.AdaptiveAvgPool1d(t_out)(x).numpy(), rtol=1e-5, atol=1e-6)


def test_encoder_restatement_self_consistency():
    """Unpinned encoder: structural checks only (state_dict key names of the library, rotary
    partial dim, both rotary pairings are orthogonal transforms, ScaleNorm output norm)."""
    enc = xt_encoder.Encoder(dim=256, depth=2, heads=4, attn_dim_head=64)
    keys = set(enc.state_dict().keys())
    for k in ("layers.0.0.0.g", "layers.0.1.to_q.weight", "layers.0.1.to_out.weight", "layers.0.2.residual_scale",
              "layers.1.1.ff.0.0.weight", "layers.1.1.ff.0.0.bias", "layers.1.1.ff.2.weight", "final_norm.g",
              "rotary_pos_emb.inv_freq"):
        assert k in keys, k
    assert enc.rotary_emb_dim == 32  # max(64 // 2, 32)
    assert xt_encoder.Encoder(dim=3072, depth=0, heads=8, attn_dim_head=384).rotary_emb_dim == 192
    x = torch.randn(2, 9, 256)
    y = enc(x)
    np.testing.assert_allclose(y.norm(dim=-1).detach().numpy(), 16.0, rtol=1e-4)  # sqrt(256) * g
    for inter in (True, False):
        rot = xt_encoder.RotaryEmbedding(32, interleaved=inter)
        q = torch.randn(1, 2, 9, 64)
        qr = xt_encoder.apply_rotary_pos_emb(q, rot(9), inter)
        np.testing.assert_allclose(qr.norm(dim=-1).numpy(), q.norm(dim=-1).numpy(), rtol=1e-5)
        np.testing.assert_array_equal(qr[..., 32:].numpy(), q[..., 32:].numpy())
        np.testing.assert_allclose(qr[:, :, 0].numpy(), q[:, :, 0].numpy(), atol=1e-6)  # position 0: identity
