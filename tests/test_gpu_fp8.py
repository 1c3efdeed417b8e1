"""fp8 (OCP e4m3) GEMM and quantisation kernels (csrc/gemm_fp8.hip through the C ABI).

Oracle: torch's own `float8_e4m3fn` conversion on the CPU (bit-exact for the quantiser, both round to nearest even and the
kernel clamps to +-448 first) and an f64 matmul over the DEquantised operands for the GEMM (the products of two e4m3 values
are exact in f32, so only the f32 accumulation differs: tolerance 1e-4 of the output scale)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def deq(u8: torch.Tensor) -> torch.Tensor:
    return u8.cpu().view(torch.float8_e4m3fn).float()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_quantize_matches_torch_e4m3(dtype):
    from tribe_hip import ops

    torch.manual_seed(0)
    x = (torch.randn(37, 200) * 3).to(dtype)
    x[0, :8] = torch.tensor([0.0, 1e-9, -1e-9, 500.0, -1e4, 448.0, 0.0019, -0.0009765625]).to(dtype)
    scale = 0.37
    q = ops.quantize_fp8(x.cuda(), scale)
    assert q.shape == (37, 256) and not q[:, 200:].any()
    want = (x.float() * np.float32(1.0 / scale)).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    got = q[:, :200].cpu()
    # +0 / -0 may differ in the sign bit after flushing tiny negatives; compare values
    assert torch.equal(deq(got), deq(want))
    amax = ops.absmax(x.cuda())
    assert float(amax) == float(x.float().abs().max())


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 520, 384), (1024, 3072, 3072), (16, 16, 256)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_gemm_fp8_vs_dequantised_reference(M, N, K, out_dtype):
    from tribe_hip import ops

    torch.manual_seed(M + N)
    a, b = torch.randn(M, K), torch.randn(N, K) * 0.05
    sa, sb = float(a.abs().max()) / 448, float(b.abs().max()) / 448
    qa, qb = ops.quantize_fp8(a.cuda(), sa), ops.quantize_fp8(b.cuda(), sb)
    bias = torch.randn(N)
    res = torch.randn(M, N)
    got = ops.gemm_fp8_nt(qa, qb, sa * sb, bias=bias.cuda(), res=res.cuda(), out_dtype=out_dtype).float().cpu()
    want = (deq(qa).double() @ deq(qb).double().T * (np.float32(sa * sb)) + bias.double() + res.double()).float()
    tol = 1e-4 if out_dtype == torch.float32 else 8e-3   # f32 accumulation over up to 3072 terms vs f64
    assert (got - want).abs().max() <= tol * want.abs().max()
    # and the quantised product tracks the fp32 product to the precision e4m3 allows (3 mantissa bits, two operands)
    exact = a @ b.T + bias + res
    assert ((got - exact).norm() / exact.norm()) < 0.05


def test_gemm_fp8_swiglu_epilogue_and_errors():
    from tribe_hip import _lib, ops

    torch.manual_seed(3)
    a, b = torch.randn(512, 256), torch.randn(1024, 256)
    sa, sb = float(a.abs().max()) / 448, float(b.abs().max()) / 448
    qa, qb = ops.quantize_fp8(a.cuda(), sa), ops.quantize_fp8(b.cuda(), sb)
    got = ops.gemm_fp8_nt(qa, qb, sa * sb, act="swiglu", out_dtype=torch.bfloat16).float().cpu()
    y = (deq(qa).double() @ deq(qb).double().T * np.float32(sa * sb)).float()
    want = torch.nn.functional.silu(y[:, 0::2]) * y[:, 1::2]
    assert got.shape == (512, 512) and (got - want).abs().max() <= 1e-2 * want.abs().max()
    with pytest.raises(ValueError):
        ops.gemm_fp8_nt(qa[:, :192].contiguous(), qb[:, :192].contiguous(), 1.0)       # K must be a multiple of 128
    with pytest.raises(TypeError):
        ops.gemm_fp8_nt(qa.float(), qb, 1.0)
    assert _lib.lib().tribe_gemm_fp8(None, None) < 0


@pytest.mark.parametrize("layernorm", [False, True])
def test_norm_quantize_fused_is_bit_identical_to_two_passes(layernorm):
    """tribe_norm_quantize_fp8_fwd == tribe_quantize_fp8_fwd(bf16 norm): same bytes, including saturated and sub-normal values."""
    from tribe_hip import ops

    g = torch.Generator().manual_seed(5)
    rows, dim = 1027, 3072
    x = (torch.randn(rows, dim, generator=g) * torch.logspace(-3, 2, rows).unsqueeze(1)).cuda()
    x[5, 7] = 1e4                                                # one outlier row: the rest of it quantises to (near) zero
    w = (1.0 + 0.2 * torch.randn(dim, generator=g)).cuda()
    b = (0.1 * torch.randn(dim, generator=g)).cuda() if layernorm else None
    for scale in (0.004, 0.02):                                  # the first one saturates the tail
        two = ops.quantize_fp8(ops.layernorm(x, w, b, 1e-6) if layernorm else ops.rmsnorm(x, w, 1e-5), scale, K_pad=dim)
        one = ops.norm_quantize_fp8(x, w, b, 1e-6 if layernorm else 1e-5, scale, layernorm)
        assert one.shape == two.shape == (rows, dim)
        assert torch.equal(one, two)
    with pytest.raises(Exception):
        ops.norm_quantize_fp8(x[:, :3064], w[:3064], None, 1e-5, 0.02, False)     # dim % 16
