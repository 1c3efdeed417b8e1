"""CPU (no GPU): the C-ABI library loads and exports every symbol include/tribe_hip.h declares,
the ctypes structs match the C layout, and the host-side mirror keeps the reference's plugin
surface (names, config fields, state_dict keys, error behaviour).  No compute call is made."""

import ctypes
import re
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "tribe_hip.h"


def _declared_symbols() -> list[str]:
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(tribe_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from tribe_hip import _lib

    handle = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/tribe_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(declared)
    assert handle.tribe_version() == _lib.ABI_VERSION


def test_ctypes_struct_layout_matches_c(tmp_path):
    """sizeof / offsetof from a C translation unit vs the ctypes mirrors."""
    from tribe_hip import _lib

    src = tmp_path / "layout.c"
    src.write_text(textwrap.dedent(f"""
        #include <stdio.h>
        #include <stddef.h>
        #include "{HEADER}"
        int main(void) {{
          printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(tribe_gemm_desc), offsetof(tribe_gemm_desc, alpha),
                 offsetof(tribe_gemm_desc, ld_gadd), sizeof(tribe_encoder_layer), sizeof(tribe_encoder_desc),
                 offsetof(tribe_encoder_desc, layers_host), sizeof(tribe_feature_piece), offsetof(tribe_feature_piece, dst_first));
          return 0;
        }}"""))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [ctypes.sizeof(_lib.GemmDesc), _lib.GemmDesc.alpha.offset, _lib.GemmDesc.ld_gadd.offset,
            ctypes.sizeof(_lib.EncoderLayer), ctypes.sizeof(_lib.EncoderDesc), _lib.EncoderDesc.layers_host.offset,
            _lib.FEATURE_PIECE_DTYPE.itemsize, _lib.FEATURE_PIECE_DTYPE.fields["dst_first"][1]]
    assert got == want


def test_argument_errors_do_not_need_a_gpu():
    """Negative return + message for malformed descriptors (validated before any launch)."""
    from tribe_hip import _lib

    handle = _lib.lib()
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.batch1, d.batch0 = 8, 8, 48, 1, 1  # K not a multiple of 64
    rc = handle.tribe_gemm_bf16(ctypes.byref(d), None)
    assert rc < 0 and b"multiple of 64" in handle.tribe_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, "tribe_gemm_bf16")
    assert handle.tribe_scalenorm_fwd(None, 4, 8, None, 1.0, 1e-5, None, 0, None) < 0
    assert handle.tribe_attention_workspace_bytes(4, 1024, 8, 384) > 0


@pytest.mark.parametrize("tiles,nk", [(576, 256), (64, 256), (120, 136), (6, 344), (272, 128), (144, 256), (432, 256), (4, 2), (300, 9), (257, 64)])
def test_stream_k_schedule_covers_every_k_step_once(tiles, nk):
    """tribe_gemm_stream_k_plan is the schedule the weight-gradient kernel decodes from blockIdx (csrc/gemm.hip, SkSched).  The decode is
    replayed here: every K-step of every tile must be covered exactly once, whole tiles by one workgroup, split tiles by parts that the
    reduce kernel finds under the slots it computes, and the second parts must be queued longest first."""
    from tribe_hip import _lib

    out = (ctypes.c_int32 * 260)()
    grid = _lib.lib().tribe_gemm_stream_k_plan(tiles, nk, out)
    tiles_dp, rem_units, q, workers = out[0], out[1], out[2], out[3]
    second_worker = list(out[4:260])
    if grid == tiles:                      # nothing split
        assert tiles_dp >= tiles
        return
    assert 0 <= tiles_dp < tiles and tiles_dp % workers == 0 and rem_units == (tiles - tiles_dp) * nk and q * workers >= rem_units
    assert (tiles - tiles_dp) * 2 <= workers and q >= 8
    cover = {t: [0] * nk for t in range(tiles)}
    slots = {}                             # (tile, first K-step) -> workspace slot of a partial workgroup
    second_lengths = []
    for b in range(grid):
        if b < tiles_dp:
            tile, k0, n = b, 0, nk
        else:
            sidx = b - tiles_dp
            second = sidx >= workers
            w = second_worker[sidx - workers] if second else sidx
            u0 = w * q
            if u0 >= rem_units:
                continue
            run = min(q, rem_units - u0)
            t_loc, x = divmod(u0, nk)
            first_len = min(run, nk - x)
            if second and first_len >= run:
                continue
            tile, k0, n = tiles_dp + t_loc + int(second), (0 if second else x), (run - first_len if second else first_len)
            if second:
                second_lengths.append(n)
            if n != nk:
                slots[(tile, k0)] = 2 * w + int(second)
        for k in range(k0, k0 + n):
            cover[tile][k] += 1
    assert all(c == 1 for row in cover.values() for c in row)
    assert second_lengths == sorted(second_lengths, reverse=True)
    # the reduce kernel's view: the parts of split tile t_loc are the runs w_lo .. w_hi, slot 2 w + (run began in the previous tile)
    for t_loc in range(tiles - tiles_dp):
        u_lo, u_hi = t_loc * nk, t_loc * nk + nk - 1
        w_lo, w_hi = u_lo // q, u_hi // q
        if w_lo == w_hi:
            assert not any(t == tiles_dp + t_loc for t, _ in slots)
            continue
        found = sorted(2 * w + int(w * q < u_lo) for w in range(w_lo, w_hi + 1))
        assert found == sorted(v for (t, _), v in slots.items() if t == tiles_dp + t_loc)


def test_split_planner_leaves_the_headline_shapes_whole():
    """tribe_gemm_stream_k_workspace_bytes is the planner's answer without a launch: the encoder GEMMs of BASELINE config 1 (M = 128 rows) are
    split over K (2-8 workgroups per 128 x 128 tile, >= 8 K-steps each, at most one round), those of the B = 4 and B = 64 batches never are;
    weight gradients take the stream-K schedule only for remainders up to half a round."""
    from tribe_hip import _lib

    def nbytes(M, N, K, trans_ab=0, **kw):
        d = _lib.GemmDesc()
        d.M, d.N, d.K, d.batch1, d.batch0, d.alpha, d.stream_k, d.trans_ab = M, N, K, 1, 1, 1.0, 1, trans_ab
        d.c_dtype = kw.get("c_dtype", _lib.F32)
        d.act = kw.get("act", _lib.ACT_NONE)
        return _lib.lib().tribe_gemm_stream_k_workspace_bytes(ctypes.byref(d))

    for (N, K), splits in {(9216, 3072): 3, (3072, 3072): 6, (12288, 3072): 2, (3072, 12288): 8}.items():
        assert nbytes(128, N, K) == splits * 128 * N * 4, (N, K)
        for M in (4096, 65536):
            assert nbytes(M, N, K) == 0, (M, N, K)
    assert nbytes(128, 3072, 3072, act=_lib.ACT_SWIGLU) == 0          # an operator the second launch does not know
    assert nbytes(128, 3072, 192) == 0                                # K too short to share out (and below the ring kernel's K)
    slot = 2 * 256 * 256 * 256 * 4                                    # stream-K: two 256 x 256 f32 slots per worker
    assert nbytes(12288, 3072, 16384, trans_ab=1) == slot and nbytes(1024, 4096, 16384, trans_ab=1) == slot
    assert nbytes(9216, 3072, 16384, trans_ab=1) == 0 and nbytes(3072, 3072, 16384, trans_ab=1) == 0


def test_host_surface_matches_reference_names():
    from algonauts2025.model import FmriEncoder, FmriEncoderConfig
    from algonauts2025.pl_module import BrainModule
    from data_utils.dataloader import SegmentData
    from modeling_utils.losses import PearsonLossConfig, TorchLossConfig
    from modeling_utils.models import MlpConfig, SubjectLayers, TransformerEncoderConfig
    from oracle import tribe_ref

    cfg = FmriEncoderConfig(n_subjects=4)
    for field, default in dict(name="FmriEncoder", feature_aggregation="cat", layer_aggregation="cat", subject_embedding=False,
                               modality_dropout=0.0, contrastive_enabled=False, contrastive_modalities=["video"],
                               contrastive_weight=0.1, contrastive_temperature=0.07).items():
        assert getattr(cfg, field) == default  # model.py:20-33
    with pytest.raises(Exception):
        FmriEncoderConfig(n_subjects=4, bogus=1)  # extra="forbid"
    fdims = {"text": (2, 24), "audio": None, "video": (2, 12)}
    small = FmriEncoderConfig(n_subjects=4, hidden=768, depth=2, heads=4, subject_embedding=True, contrastive_enabled=True)
    m = small.build(fdims, n_outputs=37, n_output_timesteps=5)
    assert isinstance(m, FmriEncoder)
    ref = tribe_ref.FmriEncoderRef(fdims, 37, 5, 4, subject_embedding=True, contrastive_modalities=["video"],
                                   dims=tribe_ref.EncoderDims(hidden=768, depth=2, heads=4))
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert "audio" not in m.projectors and set(m.contrastive_heads) == {"video"}
    # transformer.py:46-53
    with pytest.raises(ValueError):
        TransformerEncoderConfig(heads=7).build(dim=768)
    with pytest.raises(ValueError):
        TransformerEncoderConfig(heads=4).build(dim=128)
    assert isinstance(MlpConfig(norm_layer="layer", activation_layer="gelu").build(6144, 1024), torch.nn.Linear)  # common.py:124-128
    sl = SubjectLayers(48, 13, 4, bias=True)
    assert tuple(sl.weights.shape) == (4, 48, 13) and tuple(sl.bias.shape) == (4, 13) and repr(sl) == "SubjectLayers(48, 13, 4)"
    with pytest.raises(ValueError):
        SubjectLayers(4, 5, 2, init_id=True)
    # dataloader.py:33-53
    with pytest.raises(ValueError):
        SegmentData(data={}, segments=[])
    with pytest.raises(RuntimeError):
        SegmentData(data={"x": torch.zeros(2, 3)}, segments=[None])
    sd = SegmentData(data={"x": torch.zeros(2, 3)}, segments=[None, None])
    with pytest.raises(RuntimeError):
        sd["x"]
    assert sd.to("cpu").data["x"].shape == (2, 3)
    assert type(TorchLossConfig(name="MSELoss").build()).__name__ == "MSELoss"
    assert type(TorchLossConfig(name="HuberLoss").build()).__module__.startswith("torch.nn")
    assert PearsonLossConfig(reduction="sum").build().reduction == "sum"
    bm = BrainModule(m, TorchLossConfig(name="MSELoss").build(), None, {})
    for name in ("forward", "_run_step", "training_step", "validation_step", "test_step", "on_validation_epoch_end"):
        assert hasattr(bm, name)


def test_hot_path_refuses_cpu_tensors():
    """No CPU fallback: the product path raises when handed host tensors."""
    from algonauts2025.model import FmriEncoderConfig
    from tribe_hip import TribeHipError

    m = FmriEncoderConfig(n_subjects=2, hidden=768, depth=1, heads=4).build({"text": (1, 8)}, 5, 3)
    with pytest.raises(TribeHipError):
        m({"text": torch.zeros(1, 1, 8, 4), "subject_id": torch.zeros(1, 1, dtype=torch.long)})


def test_missing_library_fails_loudly(tmp_path):
    code = ("import sys; sys.path.insert(0, %r); from tribe_hip import _lib; _lib.lib()" % str(ROOT / "algonauts-2025_amd"))
    env = {"TRIBE_HIP_LIB": str(tmp_path / "nope.so"), "PATH": "/usr/bin:/bin"}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU / PyTorch fallback" in r.stderr


def test_layer_grouping_matches_oracle():
    import numpy as np

    from data_utils.features import aggregate_layers
    from oracle import tribe_ref

    lat = np.random.default_rng(0).normal(size=(29, 7)).astype(np.float32)
    for layers in ([0.5, 0.75, 1.0], [1.0], [0, 0.2, 0.4, 0.6, 0.8, 1.0], [0.5, 1.0]):
        for agg in (None, "group_mean"):
            np.testing.assert_allclose(aggregate_layers(lat, layers, agg), tribe_ref.aggregate_layers(lat, layers, agg))
    with pytest.raises(ValueError):
        aggregate_layers(lat, [0.5, 1.0], "bogus")


def test_attention_kernels_own_their_accumulator_registers():
    """The DH = 384 attention kernels keep O^T (and part of Q^T) in LITERAL accumulator registers written by inline asm
    (csrc/attn_acc_regs.h).  That is only sound while the compiler allocates nothing of its own there: the audit recompiles
    attention.hip and requires no scratch and no compiler-generated v_accvgpr_* / MFMA outside the asm blocks, within the
    register budget of one (512) / two (256) waves per SIMD.  Needs hipcc (present in the build container and on the GPU box)."""
    import shutil
    import subprocess
    import sys
    from pathlib import Path

    import pytest

    if not Path("/opt/rocm/bin/hipcc").exists() and shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    root = Path(__file__).resolve().parent.parent
    p = subprocess.run([sys.executable, str(root / "scripts" / "check_attn_wide_isa.py"), "--no-gemm"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    # the generated header is in step with its generator
    before = (root / "algonauts-2025_amd" / "csrc" / "attn_acc_regs.h").read_text()
    subprocess.run([sys.executable, str(root / "scripts" / "gen_attn_acc_regs.py")], check=True, capture_output=True)
    assert (root / "algonauts-2025_amd" / "csrc" / "attn_acc_regs.h").read_text() == before
