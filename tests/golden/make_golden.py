"""Generate golden vectors by EXECUTING the reference's own source files.

Run in the build container only (needs /root/reference, which never travels to
the GPU box):

    python tests/golden/make_golden.py

The reference cannot be imported as packages here (`exca`, `lightning`,
`wandb`, `torchvision`, `x_transformers`, `torchmetrics` are absent -- ordinary
ModuleNotFoundError).  Its hot-path files are therefore loaded one by one with
`importlib.util.spec_from_file_location` after registering inert stub modules
for the missing third-party names.  The only stub that carries arithmetic is
`x_transformers.Encoder`, filled with this build's restatement
(oracle/xt_encoder.py) -- so everything AROUND the encoder is pinned by the
reference's code and the encoder itself stays "parity unpinned".

Only data is written (inputs, expected outputs, as .npz next to this file).
Parameters are not stored: both sides regenerate them with
`oracle.tribe_ref.fill_params_` (name-keyed, order independent).
"""

from __future__ import annotations

import ast
import dataclasses
import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import torch
from torch import nn

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT))

from oracle import tribe_ref, xt_encoder  # noqa: E402


def _shell(name: str) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__path__ = []  # mark as package
    sys.modules[name] = m
    return m


def _load(name: str, path: Path) -> types.ModuleType:
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    # --- inert third-party stubs -------------------------------------------------
    tv = _shell("torchvision")
    tv.ops = _shell("torchvision.ops")

    class _NeverMLP:  # torchvision.ops.MLP: never instantiated on the hot path (common.py:124-128)
        def __init__(self, *a, **k):
            raise RuntimeError("torchvision MLP branch is off the hot path")

    tv.ops.MLP = _NeverMLP
    xt = _shell("x_transformers")
    xt.Encoder = xt_encoder.Encoder
    xt.Decoder = xt_encoder.Decoder

    pl_root = _shell("lightning")
    pl = _shell("lightning.pytorch")
    pl_root.pytorch = pl

    class _LightningModule(nn.Module):
        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    pl.LightningModule = _LightningModule
    tm = _shell("torchmetrics")
    tm.Metric = type("Metric", (nn.Module,), {})

    # --- package shells + the one boundary type ------------------------------------
    du = _shell("data_utils")
    dl = _shell("data_utils.dataloader")

    @dataclasses.dataclass
    class SegmentData:  # dataloader.py:27-53 boundary type (2 fields)
        data: dict
        segments: list

    dl.SegmentData = SegmentData
    du.dataloader = dl
    mu = _shell("modeling_utils")
    mum = _shell("modeling_utils.models")
    _shell("modeling_utils.losses")
    muo = _shell("modeling_utils.optimizers")
    muo.OptimizerConfig = object
    _shell("algonauts2025")

    # --- the reference's own files ---------------------------------------------------
    common = _load("modeling_utils.models.common", REF / "modeling_utils/modeling_utils/models/common.py")
    transformer = _load("modeling_utils.models.transformer", REF / "modeling_utils/modeling_utils/models/transformer.py")
    mum.common, mum.transformer = common, transformer
    losses = _load("modeling_utils.losses.losses", REF / "modeling_utils/modeling_utils/losses/losses.py")
    model = _load("algonauts2025.model", REF / "algonauts2025/model.py")
    pl_module = _load("algonauts2025.pl_module", REF / "algonauts2025/pl_module.py")
    _ = mu
    return common, transformer, losses, model, pl_module, SegmentData


def _extract_method(path: Path, cls: str, fn: str):
    """Compile ONE method out of a reference file (its module imports need packages
    that are absent) and return it as a plain function."""
    tree = ast.parse(path.read_text())
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == fn:
                    sub.returns = None
                    for a in sub.args.args:
                        a.annotation = None
                    m = ast.Module(body=[sub], type_ignores=[])
                    ns = {"np": np}
                    exec(compile(ast.fix_missing_locations(m), str(path), "exec"), ns)
                    return ns[fn]
    raise KeyError((cls, fn))


def main() -> None:
    common, transformer, losses, model, pl_module, SegmentData = load_reference()
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1234)
    out: dict[str, np.ndarray] = {}

    # ---- G1: SubjectLayers forward ----------------------------------------------------
    B, C, T, S, V = 5, 48, 11, 4, 13
    sl = common.SubjectLayers(C, V, S, bias=True)
    w = torch.randn(S, C, V, generator=g) / C**0.5
    b = torch.randn(S, V, generator=g) / C**0.5
    with torch.no_grad():
        sl.weights.copy_(w)
        sl.bias.copy_(b)
    x = torch.randn(B, C, T, generator=g)
    subj = torch.tensor([[2], [0], [3], [2], [1]])
    with torch.no_grad():
        y = sl(x, subj)
        sl_nb = common.SubjectLayers(C, V, S, bias=False)
        sl_nb.weights.copy_(w)
        y_nb = sl_nb(x, subj.flatten())
    np.savez(HERE / "g1_subject_layers.npz", x=x.numpy(), w=w.numpy(), b=b.numpy(), subj=subj.numpy(),
             y=y.numpy(), y_nobias=y_nb.numpy())

    # ---- G2: aggregate_features over the config grid ----------------------------------
    g2: dict[str, np.ndarray] = {}
    B, T = 2, 6
    cases = []
    for fa in ("cat", "sum"):
        for la in ("cat", "mean"):
            for variant in ("tri", "one_none", "ndim3"):
                cases.append((fa, la, variant))
    for fa, la, variant in cases:
        if variant == "tri":
            fdims = {"text": (2, 24), "audio": (2, 8), "video": (2, 12)}
        elif variant == "one_none":
            fdims = {"text": (2, 24), "audio": None, "video": (2, 12)}
        else:
            fdims = {"text": (1, 24), "audio": (1, 8), "video": (1, 12)}
        cfg = model.FmriEncoderConfig(n_subjects=4, feature_aggregation=fa, layer_aggregation=la)
        # building the full encoder costs 0.9 G params; aggregate_features needs only projectors,
        # so swap the encoder builder for a no-op during construction.
        real_build = transformer.TransformerEncoderConfig.build
        transformer.TransformerEncoderConfig.build = lambda self, dim: nn.Identity()
        try:
            m = cfg.build(fdims, n_outputs=7, n_output_timesteps=3)
        finally:
            transformer.TransformerEncoderConfig.build = real_build
        m.eval()
        with torch.no_grad():
            tribe_ref.fill_params_(m, seed=3)
        data = tribe_ref.synthetic_batch(B, T, fdims, 4, seed=11)
        if variant == "ndim3":
            data = {k: (v[:, 0] if v.ndim == 4 else v) for k, v in data.items()}
        key = f"{fa}_{la}_{variant}"
        try:
            with torch.no_grad():
                y = m.aggregate_features(SegmentData(data=data, segments=[None] * B))
        except RuntimeError:
            # reference behaviour: 'sum' fusion with a missing modality adds a 1024-wide
            # zero block to 3072-wide projections and raises (model.py:143-144,163-164)
            g2[key + "_raises"] = np.array(1)
            continue
        g2[key] = y[..., ::8].numpy()
        g2[key + "_sum"] = np.array([y.double().sum().item(), y.double().abs().sum().item()])
    np.savez(HERE / "g2_aggregate_features.npz", **g2)

    # ---- G3: full FmriEncoder.forward (reference glue + restated encoder in the stub slot) ----
    fdims = {"text": (2, 24), "audio": (2, 8), "video": (2, 12)}
    B, T, Tp, V = 2, 14, 5, 37
    g3 = {}
    for subj_emb in (False, True):
        cfg = model.FmriEncoderConfig(n_subjects=4, subject_embedding=subj_emb)
        m = cfg.build(fdims, n_outputs=V, n_output_timesteps=Tp).eval()
        with torch.no_grad():
            tribe_ref.fill_params_(m, seed=5)
        data = tribe_ref.synthetic_batch(B, T, fdims, 4, seed=17)
        data["subject_id"] = torch.tensor([[1], [3]])
        batch = SegmentData(data=data, segments=[None] * B)
        with torch.no_grad():
            tag = "se" if subj_emb else "nose"
            g3[f"pooled_{tag}"] = m(batch).numpy()
            g3[f"unpooled_{tag}"] = m(batch, pool_outputs=False).numpy()
            if not subj_emb:
                g3["latents_stats"] = np.array(
                    [m.get_brain_latents(batch).double().abs().mean().item()])
        del m
    np.savez(HERE / "g3_forward.npz", **g3)

    # ---- G4: PearsonLoss (mean / sum, with a constant column -> eps path) ---------------
    N, V = 64, 9
    x = torch.randn(N, V, generator=g)
    yv = 0.5 * x + torch.randn(N, V, generator=g)
    x[:, 3] = 1.25  # constant prediction column
    np.savez(HERE / "g4_pearson_loss.npz", x=x.numpy(), y=yv.numpy(),
             mean=losses.PearsonLoss("mean")(x, yv).numpy(), sum=losses.PearsonLoss("sum")(x, yv).numpy())

    # ---- G5: InfoNCE ----------------------------------------------------------------------
    q = torch.randn(3, 5, 32, generator=g)
    k = 0.7 * q + torch.randn(3, 5, 32, generator=g)
    np.savez(HERE / "g5_info_nce.npz", q=q.numpy(), k=k.numpy(),
             loss=model.FmriEncoder._info_nce(q, k, tau=0.07).numpy(),
             loss_tau1=model.FmriEncoder._info_nce(q, k, tau=1.0).numpy())

    # ---- G6: BrainModule._run_step flatten order / loss / repeat_interleave ----------------
    class _FixedModel(nn.Module):
        def __init__(self, y):
            super().__init__()
            self.y = y

        def forward(self, batch):
            return self.y

    B, V, Tp = 3, 4, 5
    y_pred = torch.randn(B, V, Tp, generator=g)
    y_true = torch.randn(B, V, Tp, generator=g)
    sid = torch.tensor([[2], [0], [1]])

    class _Rec:
        def __init__(self):
            self.calls = []

        def update(self, p, t, groups=None):
            self.calls.append((p.clone(), t.clone(), None if groups is None else groups.clone()))

    class GroupedRec(_Rec):
        pass

    grouped, plain = GroupedRec(), _Rec()
    bm = pl_module.BrainModule(_FixedModel(y_pred), nn.MSELoss(), None,
                               {"val/subj_pearson": grouped, "val/pearson": plain, "test/x": _Rec()})
    loss, yp_cpu, yt_cpu = bm._run_step(SegmentData(data={"fmri": y_true, "subject_id": sid}, segments=[None] * B), 0, "val")
    np.savez(HERE / "g6_run_step.npz", y_pred=y_pred.numpy(), y_true=y_true.numpy(), sid=sid.numpy(),
             loss=loss.numpy(), pred_flat=plain.calls[0][0].numpy(), true_flat=plain.calls[0][1].numpy(),
             groups=grouped.calls[0][2].numpy())

    # ---- G8: _aggregate_layers index logic (text.py:129-149; audio/video hold the same body) ----
    g8 = {}
    layer_sets = {"a": [0.5, 0.75, 1.0], "b": [0.5, 1.0], "c": [0.0, 0.25, 0.5, 0.75, 1.0], "d": [1.0],
                  "e": [0, 0.2, 0.4, 0.6, 0.8, 1.0], "f": [0.6, 0.8, 1.0]}
    for cls, path in (("LLAMA3p2", "text.py"), ("Wav2VecBert", "audio.py"), ("VJEPA2", "video.py")):
        fn = _extract_method(REF / "data_utils/data_utils/features" / path, cls, "_aggregate_layers")
        for n_states in (25, 29, 41):
            lat = np.arange(n_states * 3, dtype=np.float32).reshape(n_states, 3) ** 1.5
            for tag, layers in layer_sets.items():
                for agg in (None, "group_mean"):
                    self_ = types.SimpleNamespace(layers=layers, layer_aggregation=agg)
                    g8[f"{cls}_{n_states}_{tag}_{agg}"] = fn(self_, lat)
    g8["layer_sets_json"] = np.array(repr(layer_sets))
    np.savez(HERE / "g8_aggregate_layers.npz", **g8)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
