"""GPU parity of every HIP kernel / C-ABI entry point against the CPU oracle (or plain fp32 torch
on the host for a bare GEMM), on seeded inputs.  Tolerances are stated per test: the MFMA path
multiplies bf16-rounded operands exactly and accumulates in fp32, so against an fp32 reference fed
the SAME bf16-rounded operands only accumulation order (~1e-6 rel) and, for bf16 outputs, one
final rounding (2^-9 rel) remain."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tribe_ref, xt_encoder  # noqa: E402


def bf(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tribe_hip import ops as _ops

    return _ops


def _dev(x):
    return x.cuda()


# ------------------------------------------------------------------------------------------------
def test_gemm_identity_asymmetric(ops):
    """A = I with an asymmetric B catches a swapped C/D register->(row, col) map."""
    K = 128
    a = torch.eye(K)
    b = torch.arange(96 * K, dtype=torch.float32).reshape(96, K) % 251 - 125  # exactly representable in bf16
    out = ops.gemm_nt(_dev(a).bfloat16(), _dev(b).bfloat16())
    torch.testing.assert_close(out.cpu(), b.t().contiguous(), rtol=0, atol=0)


@pytest.mark.parametrize("tile_hint", [1, 2, 3, 4])  # 1 = 128x128 double-buffered, 2 = 256x256 8-wave (counted-vmcnt pipeline), 3 = 128x128 ring, 4 = 256x192
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 128), (77, 1000, 192), (1000, 298, 3072), (4, 5, 64), (129, 129, 64),
                                   (256, 256, 64), (512, 768, 320), (257, 511, 128), (1024, 1024, 384)])
def test_gemm_shapes(ops, M, N, K, tile_hint):
    g = torch.Generator().manual_seed(M * 7 + N)
    a, b = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g))
    out = ops.gemm_nt(_dev(a).bfloat16(), _dev(b).bfloat16(), tile_hint=tile_hint)
    ref = a.double() @ b.double().t()
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=1e-4 * K**0.5)


def test_gemm_big_tiles_race_screen(ops):
    """The 256x256 kernel keeps LDS-DMA loads in flight across barriers behind counted waits: an ordering bug
    shows as rare wrong tiles, so hammer one long-K problem repeatedly and demand bit-identical results."""
    g = torch.Generator().manual_seed(99)
    a, b = _dev(torch.randn(1024, 4096, generator=g)).bfloat16(), _dev(torch.randn(1536, 4096, generator=g)).bfloat16()
    first = ops.gemm_nt(a, b, tile_hint=2)
    ref = ops.gemm_nt(a, b, tile_hint=1)
    torch.testing.assert_close(first, ref, rtol=1e-5, atol=2e-3)  # different K-summation order only
    for _ in range(20):
        assert torch.equal(ops.gemm_nt(a, b, tile_hint=2), first)
    # the round-3 kernels keep loads in flight across barriers as well: 256 x 192 tiles (4 + 3 loads) and the 128^2 ring (3 K-tiles)
    for hint in (3, 4):
        first = ops.gemm_nt(a, b, tile_hint=hint)
        torch.testing.assert_close(first, ref, rtol=1e-5, atol=2e-3)
        for _ in range(20):
            assert torch.equal(ops.gemm_nt(a, b, tile_hint=hint), first)


@pytest.mark.parametrize("M,N,hint", [(96, 160, 0), (96, 158, 0), (384, 320, 2), (70, 256, 1), (1001, 512, 2), (257, 256, 2), (1000, 1024, 0),
                                      (70, 256, 3), (96, 158, 3), (200, 128, 3), (384, 384, 4), (257, 192, 4), (1001, 576, 4), (300, 200, 4)])
def test_gemm_epilogues(ops, M, N, hint):
    """Every operator of the epilogue, on tiles that take the generic path (N not a multiple of the tile, row / gathered adds)
    and on tiles that take the wait-free interior path (N a multiple of the tile; M = 70 / 257 / 1001 end inside a quad of
    lanes, which must still all take part in the register transpose)."""
    g = torch.Generator().manual_seed(3)
    K, T = 128, 32
    a, b = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) / K**0.5)
    bias, rs = torch.randn(N, generator=g), torch.rand(N, generator=g) + 0.5
    res = torch.randn(M, N, generator=g)
    rowadd = torch.randn(T, N, generator=g)
    gadd, gidx = torch.randn(5, N, generator=g), torch.tensor([4, 0, 2] * 11)
    A, B = _dev(a).bfloat16(), _dev(b).bfloat16()
    base = a @ b.t()
    # bias + gelu, bf16 out  (one bf16 rounding: rel 2^-8 worst case)
    out = ops.gemm_nt(A, B, bias=_dev(bias), act="gelu", out_dtype=torch.bfloat16, tile_hint=hint)
    torch.testing.assert_close(out.float().cpu(), torch.nn.functional.gelu(base + bias), rtol=2**-7, atol=1e-3)
    # residual * scale + bias, in place on the residual buffer
    x = _dev(res.clone())
    ops.gemm_nt(A, B, bias=_dev(bias), res=x, res_scale=_dev(rs), out=x, tile_hint=hint)
    torch.testing.assert_close(x.cpu(), base + bias + res * rs, rtol=1e-5, atol=1e-4)
    # alpha + row bias alone (the voxel head's operators), f32 and bf16 out
    rb = torch.randn(M, generator=g)
    out = ops.gemm_nt(A, B, alpha=0.25, bias=_dev(rb), bias_row=True, tile_hint=hint)
    torch.testing.assert_close(out.cpu(), 0.25 * base + rb[:, None], rtol=1e-5, atol=1e-4)
    out = ops.gemm_nt(A, B, res=_dev(res), out_dtype=torch.bfloat16, tile_hint=hint)
    torch.testing.assert_close(out.float().cpu(), base + res, rtol=2**-7, atol=1e-3)
    # column bias + periodic row add alone (the projector's operators: pos_embed[t] rides in the residual's LDS slot on interior tiles)
    out = ops.gemm_nt(A, B, bias=_dev(bias), rowadd=_dev(rowadd), rowadd_period=T, tile_hint=hint)
    torch.testing.assert_close(out.cpu(), base + bias + rowadd[torch.arange(M) % T], rtol=1e-5, atol=1e-4)
    # alpha, row bias, rowadd (period T) and gathered add (div T)
    out = ops.gemm_nt(A, B, alpha=0.25, bias=_dev(rb), bias_row=True, rowadd=_dev(rowadd), rowadd_period=T, gadd=_dev(gadd),
                      gadd_index=_dev(gidx), gadd_div=T, tile_hint=hint)
    m = torch.arange(M)
    want = 0.25 * base + rb[:, None] + rowadd[m % T] + gadd[gidx[m // T]]
    torch.testing.assert_close(out.cpu(), want, rtol=1e-5, atol=1e-4)
    # batched
    a3, b3 = bf(torch.randn(3, 70, 64, generator=g)), bf(torch.randn(3, 40, 64, generator=g))
    out = ops.gemm_nt(_dev(a3).bfloat16(), _dev(b3).bfloat16())
    torch.testing.assert_close(out.cpu(), torch.einsum("zmk,znk->zmn", a3, b3), rtol=1e-5, atol=1e-4)
    with pytest.raises(ValueError):
        ops.gemm_nt(_dev(torch.zeros(8, 48)).bfloat16(), _dev(torch.zeros(8, 48)).bfloat16())  # K % 64


def test_gemm_epilogue_fuzz(ops):
    """Random shapes x random operator sets x both tile sizes against an f64 reference: the wait-free interior epilogue, the
    generic edge epilogue and the masked bottom rows must agree on every combination (96 seeded cases over the four tile kernels)."""
    rng = np.random.default_rng(2025)
    for case in range(96):
        hint = int(rng.integers(1, 3)) if case < 48 else int(rng.integers(3, 5))   # (the first 48 cases are round 2's, unchanged)
        tile = 128 if hint in (1, 3) else 256
        tile_n = {1: 128, 2: 256, 3: 128, 4: 192}[hint]
        M = int(rng.choice([int(rng.integers(1, 3 * tile)), tile, 2 * tile + 3]))
        N = int(rng.choice([tile_n, 2 * tile_n, int(rng.integers(1, 2 * tile_n)) // 4 * 4 + 4]))
        K = 64 * int(rng.integers(1, 5))
        g = torch.Generator().manual_seed(case)
        a, b = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) / K**0.5)
        use_bias = int(rng.integers(0, 3))          # 0 none, 1 column, 2 row
        use_res, use_rs, use_gelu, use_rowadd = (bool(rng.integers(0, 2)) for _ in range(4))
        out_bf16 = bool(rng.integers(0, 2))
        alpha = float(rng.choice([1.0, 0.5]))
        kw = {"alpha": alpha, "tile_hint": hint, "out_dtype": torch.bfloat16 if out_bf16 else torch.float32}
        want = alpha * (a.double() @ b.double().t())
        if use_bias == 1:
            bias = torch.randn(N, generator=g)
            kw["bias"] = _dev(bias)
            want = want + bias.double()
        elif use_bias == 2:
            bias = torch.randn(M, generator=g)
            kw["bias"], kw["bias_row"] = _dev(bias), True
            want = want + bias.double()[:, None]
        if use_gelu:
            kw["act"] = "gelu"
            want = torch.nn.functional.gelu(want)
        if use_res:
            res = torch.randn(M, N, generator=g)
            kw["res"] = _dev(res)
            if use_rs:
                rs = torch.rand(N, generator=g) + 0.5
                kw["res_scale"] = _dev(rs)
                want = want + res.double() * rs.double()
            else:
                want = want + res.double()
        if use_rowadd:
            T = int(rng.integers(1, M + 1))
            ra = torch.randn(T, N, generator=g)
            kw["rowadd"], kw["rowadd_period"] = _dev(ra), T
            want = want + ra.double()[torch.arange(M) % T]
        got = ops.gemm_nt(_dev(a).bfloat16(), _dev(b).bfloat16(), **kw).double().cpu()
        tol = 2**-7 if out_bf16 else 1e-4
        err = (got - want).abs().max() / max(1.0, float(want.abs().max()))
        assert err < tol, (case, M, N, K, hint, use_bias, use_res, use_rs, use_gelu, use_rowadd, out_bf16, float(err))


@pytest.mark.parametrize("hint", [0, 1, 2, 3, 4])
def test_gemm_fused_norm_operands(ops, hint):
    """The ScaleNorm operands of the epilogue (c_bf16 copy, per-row partial sums of squares, row_scale on the consumer side) on every
    tile kernel: the number of slots per row follows the tile (tribe_gemm_sumsq_slots), their sum is the row's sum of squares."""
    import ctypes as C

    from tribe_hip import _lib

    M, N, K = 300, 768, 128
    g = torch.Generator().manual_seed(11 + hint)
    a, b = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) / K**0.5)
    res, rs, bias = torch.randn(M, N, generator=g), torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    A, B, x = _dev(a).bfloat16(), _dev(b).bfloat16(), _dev(res.clone())
    rsd, biasd = _dev(rs), _dev(bias)
    xb = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.batch1, d.batch0 = M, N, K, 1, 1
    d.A, d.lda, d.B, d.ldb = A.data_ptr(), K, B.data_ptr(), K
    d.C, d.ldc, d.c_dtype, d.alpha, d.tile_hint = x.data_ptr(), N, _lib.F32, 1.0, hint
    d.bias, d.bias_mode = biasd.data_ptr(), _lib.BIAS_COL
    d.res, d.ldres, d.res_scale = x.data_ptr(), N, rsd.data_ptr()
    d.c_bf16, d.ld_c_bf16 = xb.data_ptr(), N
    ssq = torch.full((M, N // 32), float("nan"), device="cuda")
    d.row_sumsq = ssq.data_ptr()
    slots = _lib.lib().tribe_gemm_sumsq_slots(C.byref(d))
    assert slots == {0: N // 64, 1: N // 64, 2: N // 64, 3: N // 64, 4: N // 48}[hint]
    d.ld_row_sumsq = slots
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib().tribe_gemm_bf16(C.byref(d), s), "gemm")
    want = a.double() @ b.double().t() + bias.double() + res.double() * rs.double()
    torch.testing.assert_close(x.cpu().double(), want, rtol=1e-5, atol=1e-4)
    assert torch.equal(xb.cpu(), x.cpu().bfloat16())
    part = ssq.flatten()[: M * slots].view(M, slots).cpu().double()
    torch.testing.assert_close(part.sum(1), (x.cpu().double() ** 2).sum(1), rtol=1e-5, atol=1e-4)
    # consumer side: accumulator rows scaled before bias and activation
    scale = torch.rand(M, generator=g) + 0.5
    sd = _dev(scale)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    c = _lib.GemmDesc()
    c.M, c.N, c.K, c.batch1, c.batch0 = M, N, K, 1, 1
    c.A, c.lda, c.B, c.ldb = A.data_ptr(), K, B.data_ptr(), K
    c.C, c.ldc, c.c_dtype, c.alpha, c.tile_hint = out.data_ptr(), N, _lib.BF16, 1.0, hint
    c.bias, c.bias_mode, c.act, c.row_scale = biasd.data_ptr(), _lib.BIAS_COL, _lib.ACT_GELU, sd.data_ptr()
    _lib.check(_lib.lib().tribe_gemm_bf16(C.byref(c), s), "gemm")
    want = torch.nn.functional.gelu((a.double() @ b.double().t()) * scale.double()[:, None] + bias.double())
    torch.testing.assert_close(out.cpu().double(), want, rtol=2**-7, atol=2e-3)


@pytest.mark.parametrize("role", ["qkv", "ff1", "out_proj", "ff2"])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 512, 128), (1001, 768, 192), (512, 256, 1024), (70, 512, 256)])
def test_gemm_one_wave_per_simd_roles(ops, role, M, N, K):
    """The 256 x 256 one-wave-per-SIMD kernel (tile_hint 5) exists for the four encoder GEMMs, each with its compile-time operator set
    (QKV: [row scale] -> bf16; FF1: [row scale] + bias + GELU -> bf16; out-proj / FF2: [bias] + scaled residual in place [+ bf16 copy + row
    sums of squares]).  K = 64 ... 1024 covers the K loop's one-, two- and many-tile paths; M = 70 / 300 / 1001 end inside a tile.  The same
    descriptor through the 8-wave kernel (tile_hint 2) must agree to f32 summation order, and repeated launches bit for bit."""
    import ctypes as C

    from tribe_hip import _lib

    g = torch.Generator().manual_seed(M + 3 * N + K)
    a, b = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) / K**0.5)
    bias, rs, scale = torch.randn(N, generator=g), torch.rand(N, generator=g) + 0.5, torch.rand(M, generator=g) + 0.5
    res = torch.randn(M, N, generator=g)
    A, B, biasd, rsd, sd = _dev(a).bfloat16(), _dev(b).bfloat16(), _dev(bias), _dev(rs), _dev(scale)
    res_role = role in ("out_proj", "ff2")
    base = a.double() @ b.double().t()
    s = torch.cuda.current_stream().cuda_stream
    for variant in range(2):   # 0: without the optional operands, 1: with them
        outs = {}
        for hint in (2, 5):
            d = _lib.GemmDesc()
            d.M, d.N, d.K, d.batch1, d.batch0 = M, N, K, 1, 1
            d.A, d.lda, d.B, d.ldb = A.data_ptr(), K, B.data_ptr(), K
            d.alpha, d.tile_hint, d.role = 1.0, hint, _lib.ROLES.index(role)
            if res_role:
                x = _dev(res.clone())
                xb = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
                ssq = torch.zeros(M, N // 32, device="cuda")
                d.C, d.ldc, d.c_dtype = x.data_ptr(), N, _lib.F32
                d.res, d.ldres = x.data_ptr(), N
                if role == "ff2":
                    d.bias, d.bias_mode = biasd.data_ptr(), _lib.BIAS_COL
                if variant:
                    d.res_scale = rsd.data_ptr()
                    d.c_bf16, d.ld_c_bf16, d.row_sumsq = xb.data_ptr(), N, ssq.data_ptr()
                    d.ld_row_sumsq = _lib.lib().tribe_gemm_sumsq_slots(C.byref(d))
                    assert d.ld_row_sumsq == N // 64
                _lib.check(_lib.lib().tribe_gemm_bf16(C.byref(d), s), "gemm")
                want = base + (bias.double() if role == "ff2" else 0.0) + res.double() * (rs.double() if variant else 1.0)
                torch.testing.assert_close(x.cpu().double(), want, rtol=1e-5, atol=1e-4)
                if variant:
                    assert torch.equal(xb.cpu(), x.cpu().bfloat16())
                    part = ssq.flatten()[: M * (N // 64)].view(M, N // 64).cpu().double()   # (the kernel's row pitch is the slot count)
                    torch.testing.assert_close(part.sum(1), (x.cpu().double() ** 2).sum(1), rtol=1e-5, atol=1e-4)
                outs[hint] = x.cpu()
            else:
                out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
                d.C, d.ldc, d.c_dtype = out.data_ptr(), N, _lib.BF16
                if role == "ff1":
                    d.bias, d.bias_mode, d.act = biasd.data_ptr(), _lib.BIAS_COL, _lib.ACT_GELU
                if variant:
                    d.row_scale = sd.data_ptr()
                _lib.check(_lib.lib().tribe_gemm_bf16(C.byref(d), s), "gemm")
                want = base * (scale.double()[:, None] if variant else 1.0)
                if role == "ff1":
                    want = torch.nn.functional.gelu(want + bias.double())
                torch.testing.assert_close(out.cpu().double(), want, rtol=2**-7, atol=2e-3)
                first = out.clone()
                for _ in range(5):
                    _lib.check(_lib.lib().tribe_gemm_bf16(C.byref(d), s), "gemm")
                    assert torch.equal(out, first)
                outs[hint] = out.cpu().float()
        torch.testing.assert_close(outs[5], outs[2], rtol=2**-7, atol=2e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float64])
@pytest.mark.parametrize("layer_mean", [False, True])
def test_pack_features(ops, dtype, layer_mean):
    g = torch.Generator().manual_seed(5)
    B, L, D, T = 3, 2, 37, 70
    feat = torch.randn(B, L, D, T, generator=g).to(dtype)
    out = ops.pack_features(_dev(feat), layer_mean).float().cpu()
    want = tribe_ref.prepare_modality(feat, "mean" if layer_mean else "cat")  # [B, T, K] fp32 (model.py:146-155)
    K = want.shape[-1]
    assert out.shape == (B * T, (K + 63) // 64 * 64)
    torch.testing.assert_close(out[:, :K], bf(want.reshape(B * T, K)), rtol=0, atol=0)
    assert out[:, K:].abs().max() == 0
    out3 = ops.pack_features(_dev(feat[:, 0].contiguous()), False).float().cpu()  # [B, D, T] input (model.py:148-149)
    torch.testing.assert_close(out3[:, :D], bf(feat[:, 0].float().transpose(1, 2).reshape(B * T, D)), rtol=0, atol=0)


def test_pack_weights(ops):
    g = torch.Generator().manual_seed(6)
    w = torch.randn(50, 100, generator=g)
    p = ops.pack_weight(_dev(w)).float().cpu()
    assert p.shape == (50, 128)
    torch.testing.assert_close(p[:, :100], bf(w), rtol=0, atol=0)
    assert p[:, 100:].abs().max() == 0
    for rows, cols in ((384, 3072), (33, 64), (7, 128)):   # unpadded + contiguous: the flat cast (32 per thread) and its fallbacks
        wu = torch.randn(rows, cols, generator=g)
        torch.testing.assert_close(ops.pack_weight(_dev(wu)).float().cpu(), bf(wu), rtol=0, atol=0)
    sw = torch.randn(3, 70, 45, generator=g)
    ps = ops.pack_subject_weights(_dev(sw)).float().cpu()
    assert ps.shape == (3, 128, 128)
    torch.testing.assert_close(ps[:, :45, :70], bf(sw.transpose(1, 2)), rtol=0, atol=0)
    assert ps[:, 45:].abs().max() == 0 and ps[:, :, 70:].abs().max() == 0


@pytest.mark.parametrize("legacy", [False, True])
def test_scalenorm(ops, legacy):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(37, 768, generator=g) * 3
    x[5] = 0  # eps path
    ref = xt_encoder.ScaleNorm(768, legacy=legacy)
    with torch.no_grad():
        ref.g.mul_(1.3)
        want = ref(x)
    gain_scale = 1.0 if legacy else 768**0.5
    y = ops.scalenorm(_dev(x), _dev(ref.g.detach()), gain_scale, ref.eps, torch.float32)
    torch.testing.assert_close(y.cpu(), want, rtol=1e-5, atol=1e-6)
    yb = ops.scalenorm(_dev(x), _dev(ref.g.detach()), gain_scale, ref.eps, torch.bfloat16)
    torch.testing.assert_close(yb.float().cpu(), want, rtol=2**-8, atol=1e-6)


@pytest.mark.parametrize("interleaved", [True, False])
def test_rotary(ops, interleaved):
    g = torch.Generator().manual_seed(8)
    B, T, h, d = 2, 19, 4, 192
    rot = max(d // 2, 32)
    qkv = bf(torch.randn(B * T, 3 * h * d, generator=g))
    rope = xt_encoder.RotaryEmbedding(rot, interleaved=interleaved)
    freqs = rope(T)
    half = torch.einsum("i,j->ij", torch.arange(T).float(), rope.inv_freq)
    out = ops.rotary_(_dev(qkv).bfloat16(), T, h, d, rot, _dev(half.cos().contiguous()), _dev(half.sin().contiguous()),
                      interleaved).float().cpu()
    q, k, v = qkv.view(B, T, 3, h, d).unbind(2)
    want_q = xt_encoder.apply_rotary_pos_emb(q.transpose(1, 2), freqs, interleaved).transpose(1, 2)
    want_k = xt_encoder.apply_rotary_pos_emb(k.transpose(1, 2), freqs, interleaved).transpose(1, 2)
    got = out.view(B, T, 3, h, d)
    torch.testing.assert_close(got[:, :, 0], want_q, rtol=2**-8, atol=2**-8)
    torch.testing.assert_close(got[:, :, 1], want_k, rtol=2**-8, atol=2**-8)
    torch.testing.assert_close(got[:, :, 2], v, rtol=0, atol=0)


# 0 = fused flash-style kernels (dim_head 384: 32 query rows per wave, one wave per SIMD), 1 = materialised scores (3 kernels),
# 2 = fused with dim_head 384 on the 16-row kernel, 3 = dim_head 384 on the key-split kernel (two waves per SIMD, each half of
# the head dimension).  Sequence lengths cover whole tiles, a ragged last tile (70, 298, 33, 1000),
# fewer keys than one tile (7) and the deferred-max rescale late in the key loop (1024).
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("B,T,h,d", [(2, 70, 4, 64), (1, 298, 2, 192), (3, 128, 8, 384), (2, 1024, 2, 384), (1, 33, 1, 128), (2, 70, 3, 384),
                                     (1, 298, 8, 384), (1, 7, 2, 384), (1, 1000, 1, 384)])
def test_attention(ops, B, T, h, d, mode):
    g = torch.Generator().manual_seed(9)
    qkv = bf(torch.randn(B * T, 3 * h * d, generator=g))
    if T == 1024:  # force the deferred-max rescale branch late in the key loop: one huge score at key 900 for query 5
        qkv.view(B, T, 3, h, d)[0, 5, 0, 0] *= 6.0
        qkv.view(B, T, 3, h, d)[0, 900, 1, 0] = bf(qkv.view(B, T, 3, h, d)[0, 5, 0, 0] * 0.5)
    scale = d**-0.5
    ops.attention_set_mode(mode)
    try:
        out = ops.attention(_dev(qkv).bfloat16(), B, T, h, d, scale).float().cpu()
    finally:
        ops.attention_set_mode(0)
    q, k, v = (t.transpose(1, 2) for t in qkv.view(B, T, 3, h, d).unbind(2))
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * scale
    want = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v).transpose(1, 2).reshape(B * T, h * d)
    # P and the output are rounded to bf16 once each: 2 * 2^-9 relative on O(1/sqrt(T))-sized values
    torch.testing.assert_close(out, want, rtol=2**-6, atol=3e-3)


@pytest.mark.parametrize("B,T,h", [(2, 70, 3), (1, 298, 2), (2, 1024, 2), (1, 7, 1)])
def test_attention_log_sum_exp_output(ops, B, T, h):
    """tribe_attention_desc.lse (dim_head 384, one-wave-per-SIMD kernel): lse2[b, h, q] = log2 sum_j 2^(q.k_j scale log2e), i.e.
    torch.logsumexp(scores) / ln 2, and the output is the same as without it.  Other head sizes / modes refuse the request."""
    d = 384
    g = torch.Generator().manual_seed(21)
    qkv = bf(torch.randn(B * T, 3 * h * d, generator=g))
    if T == 1024:
        qkv.view(B, T, 3, h, d)[0, 5, 0, 0] *= 6.0
        qkv.view(B, T, 3, h, d)[0, 900, 1, 0] = bf(qkv.view(B, T, 3, h, d)[0, 5, 0, 0] * 0.5)
        qkv = bf(qkv)   # (x 6 leaves the bf16 grid)
    scale = d**-0.5
    dev_qkv = _dev(qkv).bfloat16()
    out, lse = ops.attention_with_lse(dev_qkv, B, T, h, d, scale)
    assert torch.equal(out, ops.attention(dev_qkv, B, T, h, d, scale))
    q, k, _ = (t.transpose(1, 2) for t in qkv.view(B, T, 3, h, d).unbind(2))
    sim = torch.einsum("bhid,bhjd->bhij", q.double(), k.double()) * scale
    want = torch.logsumexp(sim, dim=-1) / np.log(2.0)
    torch.testing.assert_close(lse.cpu().double(), want, rtol=0, atol=2e-3)   # bf16 P in the running sum: ~2^-9 relative on the sum
    assert not ops.attention_lse_supported(64) and not ops.attention_lse_supported(128)
    ops.attention_set_mode(3)
    try:
        assert not ops.attention_lse_supported(384)
        with pytest.raises(ValueError, match="log-sum-exp"):
            ops.attention_with_lse(dev_qkv, B, T, h, d, scale)
    finally:
        ops.attention_set_mode(0)


@pytest.mark.parametrize("nb,h,T,K", [(2, 3, 256, 128), (1, 2, 70, 64), (2, 2, 298, 384), (1, 1, 1024, 384), (3, 1, 300, 64)])
def test_gemm_exp2_and_mul_aux_epilogues(ops, nb, h, T, K):
    """The two epilogue operators of the attention backward, batched over (sequence, head) with a per-(sequence, head) ROW bias
    (tribe_gemm_desc.sBias0): P = exp2(alpha a.b + bias[row]) and dS = (alpha a.b + bias[row]) * aux, aux addressed with C's batch strides.
    T = 256 / 1024 run the wait-free whole-wave epilogue, 70 / 298 / 300 the per-row path (tiles that cross N, pad columns left untouched)."""
    from modeling_utils import autograd as ag
    from tribe_hip import _lib

    g = torch.Generator().manual_seed(nb * 100 + T)
    Tp = ops.round_up(T, 64)
    a, b = bf(torch.randn(nb, h, T, K, generator=g)), bf(torch.randn(nb, h, T, K, generator=g))
    bias = torch.randn(nb, h, T, generator=g)
    alpha = 0.37 * K**-0.5
    lin = alpha * torch.einsum("zhik,zhjk->zhij", a, b) + bias[..., None]
    da, db, dbias = _dev(a).bfloat16().contiguous(), _dev(b).bfloat16().contiguous(), _dev(bias)
    kw = dict(lda=K, ldb=K, ldc=Tp, M=T, N=T, K=K, alpha=alpha, batch1=nb, batch0=h, sA=(h * T * K, T * K), sB=(h * T * K, T * K),
              sC=(h * T * Tp, T * Tp), row_bias=dbias, sBias=(h * T, T))
    P = torch.full((nb, h, T, Tp), 7.0, dtype=torch.bfloat16, device="cuda")
    ag._gemm(da, db, P, act=_lib.ACT_EXP2, **kw)
    torch.testing.assert_close(P[..., :T].float().cpu(), torch.exp2(lin), rtol=2**-7, atol=1e-6)
    assert (P[..., T:] == 7.0).all()
    dS = torch.full((nb, h, T, Tp), 7.0, dtype=torch.bfloat16, device="cuda")
    ag._gemm(da, db, dS, act=_lib.ACT_MUL_AUX, aux=P, ld_aux=Tp, **kw)
    torch.testing.assert_close(dS[..., :T].float().cpu(), lin * P[..., :T].float().cpu(), rtol=2**-7, atol=1e-5)
    assert (dS[..., T:] == 7.0).all()
    with pytest.raises(ValueError, match="MUL_AUX"):
        ag._gemm(da, db, dS, act=_lib.ACT_MUL_AUX, **kw)                                 # no aux
    with pytest.raises(ValueError, match="row bias"):
        ag._gemm(da, db, P, act=_lib.ACT_EXP2, bias=_dev(torch.zeros(T)), **{k: v for k, v in kw.items() if k not in ("row_bias", "sBias")})


@pytest.mark.parametrize("B,T,h,d", [(2, 70, 3, 384), (1, 1024, 8, 384), (3, 33, 2, 64), (1, 5, 1, 1032)])
def test_rowdot_heads(ops, B, T, h, d):
    g = torch.Generator().manual_seed(5)
    a, b = bf(torch.randn(B * T, h * d, generator=g)), bf(torch.randn(B * T, h * d, generator=g))
    got = ops.rowdot_heads(_dev(a).bfloat16(), _dev(b).bfloat16(), B, T, h, d, -0.25).cpu()
    want = -0.25 * (a.view(B, T, h, d).double() * b.view(B, T, h, d).double()).sum(-1).permute(0, 2, 1)
    torch.testing.assert_close(got.double(), want, rtol=1e-5, atol=1e-4)


# Transposed-operand GEMM (tribe_gemm_desc.trans_ab): C = At^T Bt, the weight-gradient form.  Whole tiles, ragged M / N (multiples of 8
# only), one K-tile, a long reduction, bf16 output + bias.
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 1024), (3072, 1024, 4096), (264, 520, 128), (8, 16, 64), (1000, 296, 192)])
def test_gemm_transposed_operands(ops, M, N, K):
    g = torch.Generator().manual_seed(12)
    at, bt = bf(torch.randn(K, M, generator=g)), bf(torch.randn(K, N, generator=g))
    bias = torch.randn(N, generator=g)
    want = at.t() @ bt
    got = ops.gemm_tn(_dev(at).bfloat16(), _dev(bt).bfloat16()).cpu()
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-3 * K**0.5)
    got = ops.gemm_tn(_dev(at).bfloat16(), _dev(bt).bfloat16(), out_dtype=torch.bfloat16, alpha=0.5, bias=_dev(bias)).float().cpu()
    torch.testing.assert_close(got, bf(0.5 * want + bias), rtol=2**-7, atol=2e-2 * K**0.5)
    with pytest.raises(ValueError, match="multiples of 8"):
        ops.gemm_tn(_dev(at[:, : M - 3].contiguous()).bfloat16(), _dev(bt).bfloat16())   # M not a multiple of 8


# Split-K of small NT grids (tribe_gemm_desc.stream_k without trans_ab): the four GEMMs of an encoder layer at M = 128 rows (BASELINE config 1)
# with their model epilogues -- QKV: row_scale -> bf16; FF1: row_scale + bias + GELU -> bf16; out-proj / FF2: [bias +] scaled f32 residual in
# place + bf16 copy + row sums of squares -- against the same launch without the split; plain shapes through ops.gemm_nt (ragged M, rowadd).
@pytest.mark.parametrize("role,M,N,K", [("qkv", 128, 9216, 3072), ("ff1", 128, 12288, 3072), ("out_proj", 128, 3072, 3072), ("ff2", 128, 3072, 12288),
                                        ("ff2", 200, 768, 4096), ("qkv", 100, 1152, 1024)])
def test_gemm_split_k_with_model_epilogues(ops, role, M, N, K):
    import ctypes as C

    from tribe_hip import _lib

    g = torch.Generator().manual_seed(N + K)
    a, b = _dev(torch.randn(M, K, generator=g)).bfloat16(), _dev(torch.randn(N, K, generator=g) / K**0.5).bfloat16()
    bias, rs, scale = _dev(torch.randn(N, generator=g)), _dev(torch.rand(N, generator=g) + 0.5), _dev(torch.rand(M, generator=g) + 0.5)
    x0 = _dev(torch.randn(M, N, generator=g))
    res_role = role in ("out_proj", "ff2")
    s = torch.cuda.current_stream().cuda_stream
    got = {}
    for split in (False, True):
        x = x0.clone()
        out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        ssq = torch.full((M, N // 32 + 1), float("nan"), device="cuda")
        d = _lib.GemmDesc()
        d.M, d.N, d.K, d.batch1, d.batch0 = M, N, K, 1, 1
        d.A, d.lda, d.B, d.ldb, d.alpha = a.data_ptr(), K, b.data_ptr(), K, 1.0
        d.role = _lib.ROLES.index(role)
        if res_role:
            d.C, d.ldc, d.c_dtype, d.res, d.ldres, d.res_scale = x.data_ptr(), N, _lib.F32, x.data_ptr(), N, rs.data_ptr()
            d.c_bf16, d.ld_c_bf16, d.row_sumsq = out.data_ptr(), N, ssq.data_ptr()
            if role == "ff2":
                d.bias, d.bias_mode = bias.data_ptr(), _lib.BIAS_COL
        else:
            d.C, d.ldc, d.c_dtype, d.row_scale = out.data_ptr(), N, _lib.BF16, scale.data_ptr()
            if role == "ff1":
                d.bias, d.bias_mode, d.act = bias.data_ptr(), _lib.BIAS_COL, _lib.ACT_GELU
        nbytes = 0
        if split:
            d.stream_k = 1
            nbytes = _lib.lib().tribe_gemm_stream_k_workspace_bytes(C.byref(d))
            assert nbytes > 0, "these shapes are far below one round of tiles: the planner must split them"
            ws = ops.workspace(nbytes, x.device, tag="streamk")
            d.stream_k_ws, d.stream_k_ws_bytes = ws.data_ptr(), ws.numel()
        slots = 0
        if res_role:
            slots = _lib.lib().tribe_gemm_sumsq_slots(C.byref(d))
            d.ld_row_sumsq = slots
        _lib.check(_lib.lib().tribe_gemm_bf16(C.byref(d), s), "tribe_gemm_bf16")
        got[split] = (x.cpu(), out.float().cpu(), ssq.flatten()[: M * slots].view(M, slots).sum(1).cpu() if res_role else None)
    if res_role:
        torch.testing.assert_close(got[True][0], got[False][0], rtol=2e-5, atol=2e-5 * K**0.5)
        assert torch.equal(got[True][1], got[True][0].bfloat16().float())                       # the bf16 copy is the copy of what was stored
        torch.testing.assert_close(got[True][2], (got[True][0].double() ** 2).sum(1).float(), rtol=1e-5, atol=1e-3)
    else:
        torch.testing.assert_close(got[True][1], got[False][1], rtol=2**-7, atol=2e-3)       # bf16 outputs: one rounding step apart at most
    if split and not res_role:
        with pytest.raises(ValueError, match="split over K"):                                   # split planned, no workspace handed over
            d.stream_k_ws = None
            _lib.check(_lib.lib().tribe_gemm_bf16(C.byref(d), s), "tribe_gemm_bf16")


def test_gemm_split_k_plain_shapes(ops):
    g = torch.Generator().manual_seed(4)
    for M, N, K, kw in [(77, 1000, 4096, {}), (128, 1024, 4096, {"rowadd": True}), (60, 256, 8192, {"out_dtype": torch.bfloat16, "bias": True, "act": "gelu"})]:
        a, b = _dev(torch.randn(M, K, generator=g)).bfloat16(), _dev(torch.randn(N, K, generator=g) / K**0.5).bfloat16()
        args = {"out_dtype": kw.get("out_dtype", torch.float32), "act": kw.get("act")}
        if kw.get("bias"):
            args["bias"] = _dev(torch.randn(N, generator=g))
        if kw.get("rowadd"):
            args["rowadd"], args["rowadd_period"] = _dev(torch.randn(32, N, generator=g)), 32
        whole = ops.gemm_nt(a, b, **args)
        split = ops.gemm_nt(a, b, split_k=True, **args)
        assert ops.gemm_nt.last_split
        tol = dict(rtol=2**-7, atol=2e-3) if args["out_dtype"] == torch.bfloat16 else dict(rtol=2e-5, atol=2e-5 * K**0.5)
        torch.testing.assert_close(split.float(), whole.float(), **tol)
        assert torch.equal(ops.gemm_nt(a, b, split_k=True, **args), split)       # shares summed in a fixed order
    big = ops.gemm_nt(_dev(torch.randn(4096, 256, generator=g)).bfloat16(), _dev(torch.randn(3072, 256, generator=g)).bfloat16(), split_k=True)
    assert not ops.gemm_nt.last_split and big.shape == (4096, 3072)


# Stream-K schedule of the transposed-operand GEMM (tribe_gemm_desc.stream_k): shapes whose tile count leaves at most half a round on 256 CUs --
# 6 tiles (ragged M / N), 120 tiles of 136 K-steps (runs of 64 K-steps: most cross a tile boundary), 272 tiles (one whole round stored plainly
# + 16 tiles cut into 256 runs) -- and two that are left alone (4 tiles x 2 K-steps; 144 tiles = more than half a round).
@pytest.mark.parametrize("M,N,K,splits", [(264, 520, 22016, True), (2560, 3072, 8704, True), (4352, 4096, 8192, True), (512, 512, 128, False),
                                          (3072, 3072, 4096, False)])
def test_gemm_transposed_operands_stream_k(ops, M, N, K, splits):
    g = torch.Generator().manual_seed(M + K)
    at, bt = _dev(torch.randn(K, M, generator=g)).bfloat16(), _dev(torch.randn(K, N, generator=g)).bfloat16()
    whole = ops.gemm_tn(at, bt, alpha=0.5)
    split = ops.gemm_tn(at, bt, alpha=0.5, stream_k=True)
    assert ops.gemm_tn.last_split == splits
    # same products, f32 accumulation in a different order: K^0.5 * 2^-24-sized differences on K^0.5-sized sums
    torch.testing.assert_close(split, whole, rtol=2e-5, atol=2e-5 * K**0.5)
    if not splits:
        assert torch.equal(split, whole)
    for _ in range(3):   # the parts of a split tile are summed in a fixed order: bit-reproducible
        assert torch.equal(ops.gemm_tn(at, bt, alpha=0.5, stream_k=True), split)
    ref = 0.5 * (at[:, :64].float().t() @ bt[:, :96].float())
    torch.testing.assert_close(split[:64, :96], ref, rtol=1e-4, atol=1e-3 * K**0.5)
    with pytest.raises(ValueError, match="stream_k"):
        ops.gemm_tn(at, bt, stream_k=True, bias=_dev(torch.zeros(N)))


# dim_head 64 (the extractors) on the 64-row-per-wave kernels (mode 4: four waves; mode 5: anti-phase wave pairs; mode 0 picks by
# grid size) and on the 16-row kernel (mode 2): several query blocks, ragged last key tile and last row tile, one sub-tile only
# (T = 20), an odd number of sub-tiles (T = 1000: 32 sub-tiles, T = 3000: 94, T = 257: 9), a late deferred-max rescale.
@pytest.mark.parametrize("mode", [0, 2, 4, 5])
@pytest.mark.parametrize("B,T,h", [(1, 20, 2), (2, 257, 3), (1, 1000, 5), (1, 3000, 2), (3, 64, 1)])
def test_attention_dim_head_64(ops, B, T, h, mode):
    d = 64
    g = torch.Generator().manual_seed(10)
    qkv = bf(torch.randn(B * T, 3 * h * d, generator=g))
    if T >= 1000:
        qkv.view(B, T, 3, h, d)[0, 37, 0, 0] *= 6.0
        qkv.view(B, T, 3, h, d)[0, 900, 1, 0] = bf(qkv.view(B, T, 3, h, d)[0, 37, 0, 0] * 0.5)
    ops.attention_set_mode(mode)
    try:
        dev_qkv = _dev(qkv).bfloat16()
        out = ops.attention(dev_qkv, B, T, h, d, d**-0.5).float().cpu()
        # bit-identical from launch to launch: a read of an accumulator before its MFMA has written it (no hardware interlock) shows up here
        for _ in range(3):
            assert torch.equal(ops.attention(dev_qkv, B, T, h, d, d**-0.5).float().cpu(), out)
    finally:
        ops.attention_set_mode(0)
    q, k, v = (t.transpose(1, 2) for t in qkv.view(B, T, 3, h, d).unbind(2))
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * d**-0.5
    want = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v).transpose(1, 2).reshape(B * T, h * d)
    torch.testing.assert_close(out, want, rtol=2**-6, atol=3e-3)


# relative_key bias of Wav2Vec-BERT (HF modeling_wav2vec2_bert.py:308-320: distance clamped to [-left, right], q . E[distance] added
# before the scale).  The 64-row kernel adds the two out-of-band constants per row and gathers only inside the band; T = 3000 is the
# extractor's 60 s chunk, (left, right) = (64, 8) its geometry, (5, 3) / (0, 0) / (100, 90) move the band edges across sub-tiles.
@pytest.mark.parametrize("mode", [0, 2, 4, 5])
@pytest.mark.parametrize("B,T,h,left,right", [(1, 3000, 2, 64, 8), (2, 300, 3, 64, 8), (1, 70, 2, 5, 3), (1, 500, 1, 0, 0), (1, 400, 2, 100, 90)])
def test_attention_relative_key(ops, B, T, h, left, right, mode):
    d = 64
    g = torch.Generator().manual_seed(11)
    qkv = bf(torch.randn(B * T, 3 * h * d, generator=g))
    npos = left + right + 1
    stride = (npos + 7) // 8 * 8
    emb = torch.randn(npos, d, generator=g) * 0.5
    q, k, v = (t.transpose(1, 2) for t in qkv.view(B, T, 3, h, d).unbind(2))           # [B, h, T, d]
    qe = torch.einsum("bhid,pd->bhip", q, emb)                                            # [B, h, T, npos]
    qe_dev = torch.zeros(B * T, h, stride)
    qe_dev[:, :, :npos] = qe.permute(0, 2, 1, 3).reshape(B * T, h, npos)
    ops.attention_set_mode(mode)
    try:
        out = ops.attention_relative_key(_dev(qkv).bfloat16(), B, T, h, d, d**-0.5, _dev(qe_dev), left, right).float().cpu()
    finally:
        ops.attention_set_mode(0)
    dist = (torch.arange(T)[None, :] - torch.arange(T)[:, None]).clamp(-left, right) + left     # [i, j]
    bias = torch.gather(qe, 3, dist[None, None].expand(B, h, T, T))
    sim = (torch.einsum("bhid,bhjd->bhij", q, k) + bias) * d**-0.5
    want = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v).transpose(1, 2).reshape(B * T, h * d)
    torch.testing.assert_close(out, want, rtol=2**-6, atol=3e-3)


@pytest.mark.parametrize("interleaved,legacy", [(True, False), (False, True)])
def test_encoder_vs_oracle(ops, interleaved, legacy):
    """2-layer dim-768 encoder: HIP (bf16 operands, fp32 accumulate / residual) vs the fp32 oracle."""
    from modeling_utils.models.transformer import TransformerEncoderConfig

    torch.manual_seed(0)
    dim, depth, heads, B, T = 768, 2, 4, 2, 50
    ref = xt_encoder.Encoder(dim=dim, depth=depth, heads=heads, attn_dim_head=dim // heads, rotary_interleaved=interleaved,
                             legacy_scalenorm=legacy).eval()
    with torch.no_grad():
        tribe_ref.fill_params_(ref, seed=1)
    enc = TransformerEncoderConfig(depth=depth, heads=heads, attn_dropout=0.0, rotary_interleaved=interleaved,
                                   legacy_scalenorm=legacy).build(dim)
    enc.load_state_dict(ref.state_dict())
    enc = enc.cuda()
    x = torch.randn(B, T, dim)
    with torch.no_grad():
        want = ref(x)
    got = enc(_dev(x)).cpu()
    err = (got - want).norm() / want.norm()
    assert err < 6e-3, f"relative L2 error {err:.2e}"  # ~2^-9 per bf16 operand rounding, 8 GEMMs + 2 attention products deep


def test_voxel_head_golden(ops, golden_dir):
    """G1 (reference SubjectLayers executed on CPU) vs tribe_voxel_head_fwd through the module API."""
    from modeling_utils.models import SubjectLayers

    g = np.load(golden_dir / "g1_subject_layers.npz")
    x, w, b = (torch.from_numpy(g[k]) for k in ("x", "w", "b"))
    subj = torch.from_numpy(g["subj"])
    sl = SubjectLayers(48, 13, 4, bias=True)
    with torch.no_grad():
        sl.weights.copy_(w)
        sl.bias.copy_(b)
    sl = sl.cuda()
    y = sl(_dev(x), _dev(subj)).cpu()
    # operands are rounded to bf16 (2^-9 rel each) before an exact 48-term fp32 dot product of O(1) values
    torch.testing.assert_close(y, torch.from_numpy(g["y"]), rtol=0, atol=3e-2)
    want_bf = tribe_ref.subject_layers_fwd(bf(x), bf(w), b, subj)
    torch.testing.assert_close(y, want_bf, rtol=1e-5, atol=1e-5)  # same rounded operands -> fp32-exact
    with pytest.raises(AssertionError):
        sl(_dev(x), _dev(subj + 2))  # common.py:53-55
    sl_nb = SubjectLayers(48, 13, 4, bias=False)
    with torch.no_grad():
        sl_nb.weights.copy_(w)
    y = sl_nb.cuda()(_dev(x), _dev(subj.flatten())).cpu()
    torch.testing.assert_close(y, tribe_ref.subject_layers_fwd(bf(x), bf(w), None, subj), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("t_in,t_out", [(298, 100), (14, 5), (1024, 1024), (100, 7)])
def test_adaptive_pool(ops, t_in, t_out):
    x = torch.randn(3, 11, t_in)
    torch.testing.assert_close(ops.adaptive_avg_pool(_dev(x), t_out).cpu(), torch.nn.AdaptiveAvgPool1d(t_out)(x), rtol=1e-5, atol=1e-6)


def test_losses_and_pearson(ops, golden_dir):
    from modeling_utils.losses import MSELoss, PearsonLoss
    from modeling_utils.metrics import GroupedMetric, MultidimPearsonCorrCoef

    g4 = np.load(golden_dir / "g4_pearson_loss.npz")
    x, y = torch.from_numpy(g4["x"]), torch.from_numpy(g4["y"])
    np.testing.assert_allclose(PearsonLoss("mean")(_dev(x), _dev(y)).cpu().numpy(), g4["mean"], rtol=1e-5)
    np.testing.assert_allclose(PearsonLoss("sum")(_dev(x), _dev(y)).cpu().numpy(), g4["sum"], rtol=1e-5)
    with pytest.raises(ValueError):
        PearsonLoss("none")(_dev(x), _dev(y))
    g = torch.Generator().manual_seed(11)
    B, V, T = 5, 33, 100
    pred = torch.randn(B, V, T, generator=g)
    true = 0.3 * pred + torch.randn(B, V, T, generator=g)
    sid = torch.tensor([[2], [0], [1], [2], [0]])
    loss, p_flat, t_flat, groups = tribe_ref.run_step(pred, true, sid)
    np.testing.assert_allclose(MSELoss()(_dev(pred), _dev(true)).cpu().numpy(), loss.numpy(), rtol=1e-6)
    np.testing.assert_allclose(PearsonLoss().forward_bvt(_dev(pred), _dev(true)).cpu().numpy(),
                               tribe_ref.pearson_loss(p_flat, t_flat).numpy(), rtol=1e-5)
    np.testing.assert_allclose(PearsonLoss()(_dev(p_flat), _dev(t_flat)).cpu().numpy(),
                               tribe_ref.pearson_loss(p_flat, t_flat).numpy(), rtol=1e-5)
    # streaming metric over two updates vs scipy on the concatenation (main.py:474-477)
    ref = tribe_ref.scipy_pearson_columns(p_flat.numpy(), t_flat.numpy())
    metric = MultidimPearsonCorrCoef(num_outputs=V)
    metric.update(_dev(pred[:2]), _dev(true[:2]))
    metric.update(_dev(flat := tribe_ref.flatten_bt(pred[2:])), _dev(tribe_ref.flatten_bt(true[2:])))  # [N, V] form
    np.testing.assert_allclose(metric.per_output()[0].cpu().numpy(), ref, atol=2e-6)
    np.testing.assert_allclose(float(metric.compute()), ref.mean(), atol=2e-6)
    _ = flat
    # grouped by subject (metrics/base.py:52-78): keys in first-seen order, value = mean r of that subject's rows
    gm = GroupedMetric("MultidimPearsonCorrCoef", {"num_outputs": V})
    gm.update(_dev(pred), _dev(true), groups=_dev(sid))
    out = gm.compute()
    assert list(out.keys()) == ["2", "0", "1"]
    for key in out:
        rows = groups.flatten() == int(key)
        want = tribe_ref.scipy_pearson_columns(p_flat[rows].numpy(), t_flat[rows].numpy()).mean()
        assert abs(out[key] - want) < 2e-6
    gm2 = GroupedMetric("MultidimPearsonCorrCoef", {"num_outputs": V})
    gm2.update(_dev(p_flat), _dev(t_flat), groups=_dev(groups))  # reference calling convention ([N, V] + per-row groups)
    for key in out:
        assert abs(gm2.compute()[key] - out[key]) < 1e-6
