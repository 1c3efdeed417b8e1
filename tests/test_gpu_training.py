"""GPU: the backward pass (pl_module.training_step -> loss.backward()) against the CPU oracle's autograd.
Every forward and backward op of the training path is a HIP kernel; torch.autograd only orders them.
Tolerance: bf16 GEMM operands (activations, weights AND incoming gradients are rounded to bf16 before each MFMA
product, f32 accumulation) vs an fp32 CPU graph -> per-tensor relative L2 error of a few 1e-2 at most."""

import queue
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tribe_ref  # noqa: E402


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_autograd_blocks_vs_torch():
    """Each autograd function alone vs a plain torch fp32 graph fed the same bf16-rounded inputs."""
    from modeling_utils import autograd as ag

    g = torch.Generator().manual_seed(0)
    dev = "cuda"
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    # --- Linear with scaled residual
    M, K, N = 200, 128, 192
    x = bf(torch.randn(M, K, generator=g)); w = bf(torch.randn(N, K, generator=g) / K**0.5); b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g); rs = torch.rand(N, generator=g) + 0.5; dy = bf(torch.randn(M, N, generator=g))
    wt, bt, rt, rst = (t.clone().requires_grad_() for t in (w, b, res, rs)); xt = x.clone().requires_grad_()
    (xt @ wt.t() + bt + rt * rst).backward(dy)
    xg = x.to(dev).bfloat16().requires_grad_(); wg, bg, rg, rsg = (t.to(dev).requires_grad_() for t in (w, b, res, rs))
    y = ag.Linear.apply(xg, wg, bg, rg, rsg, True)
    y.backward(dy.to(dev))
    for name, got, want in (("dx", xg.grad.float(), xt.grad), ("dw", wg.grad, wt.grad), ("db", bg.grad, bt.grad), ("dres", rg.grad, rt.grad),
                            ("drs", rsg.grad, rst.grad)):
        assert _rel(got.cpu(), want) < 2e-2, name
    # --- FeedForward
    D, Fh = 128, 256
    x = bf(torch.randn(M, D, generator=g)); w1 = bf(torch.randn(Fh, D, generator=g) / D**0.5); b1 = torch.randn(Fh, generator=g) * 0.1
    w2 = bf(torch.randn(D, Fh, generator=g) / Fh**0.5); b2 = torch.randn(D, generator=g) * 0.1
    res = torch.randn(M, D, generator=g); rs = torch.rand(D, generator=g) + 0.5; dy = bf(torch.randn(M, D, generator=g))
    ts = [t.clone().requires_grad_() for t in (x, w1, b1, w2, b2, res, rs)]
    (torch.nn.functional.gelu(ts[0] @ ts[1].t() + ts[2]) @ ts[3].t() + ts[4] + ts[5] * ts[6]).backward(dy)
    gs = [x.to(dev).bfloat16().requires_grad_()] + [t.to(dev).requires_grad_() for t in (w1, b1, w2, b2, res, rs)]
    ag.FeedForward.apply(*gs).backward(dy.to(dev))
    for name, got, want in zip(("dx", "dw1", "db1", "dw2", "db2", "dres", "drs"), gs, ts):
        assert _rel(got.grad.float().cpu(), want.grad) < 3e-2, name
    # --- ScaleNorm
    x = torch.randn(50, 256, generator=g); gpar = torch.tensor([1.3]); dy = torch.randn(50, 256, generator=g)
    xt, gt = x.clone().requires_grad_(), gpar.clone().requires_grad_()
    (xt / xt.norm(dim=-1, keepdim=True).clamp(min=1e-12) * 16.0 * gt).backward(dy)
    xg, gg = x.to(dev).requires_grad_(), gpar.to(dev).requires_grad_()
    ag.ScaleNorm.apply(xg, gg, 16.0, 1e-12, True).backward(dy.to(dev))
    assert _rel(xg.grad.cpu(), xt.grad) < 1e-4 and _rel(gg.grad.cpu(), gt.grad) < 1e-4
    # --- Attention (+ rotary): grads wrt the fused qkv buffer
    B, T, h, d = 2, 70, 2, 64
    qkv = bf(torch.randn(B * T, 3 * h * d, generator=g)); dout = bf(torch.randn(B * T, h * d, generator=g))
    qt = qkv.clone().requires_grad_()
    q, k, v = (t.transpose(1, 2) for t in qt.view(B, T, 3, h, d).unbind(2))
    att = (torch.einsum("bhid,bhjd->bhij", q, k) * d**-0.5).softmax(-1)
    torch.einsum("bhij,bhjd->bhid", att, v).transpose(1, 2).reshape(B * T, h * d).backward(dout)
    qg = qkv.to(dev).bfloat16().requires_grad_()
    ag.Attention.apply(qg, B, T, h, d, d**-0.5).backward(dout.to(dev).bfloat16())
    assert _rel(qg.grad.float().cpu(), qt.grad) < 3e-2
    # --- adaptive pool and MSE
    x = torch.randn(3, 5, 14, generator=g); xt = x.clone().requires_grad_()
    dy = torch.randn(3, 5, 5, generator=g)
    torch.nn.functional.adaptive_avg_pool1d(xt, 5).backward(dy)
    xg = x.to(dev).requires_grad_()
    ag.AdaptivePool.apply(xg, 5).backward(dy.to(dev))
    torch.testing.assert_close(xg.grad.cpu(), xt.grad, rtol=1e-5, atol=1e-6)
    p, t = torch.randn(4, 6, 9, generator=g), torch.randn(4, 6, 9, generator=g)
    pt = p.clone().requires_grad_(); (((pt - t) ** 2).mean() * 3.0).backward()
    pg = p.to(dev).requires_grad_(); (ag.MSE.apply(pg, t.to(dev)) * 3.0).backward()
    torch.testing.assert_close(pg.grad.cpu(), pt.grad, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("B,T,h,d", [(2, 128, 2, 64), (1, 192, 3, 384), (2, 70, 2, 64), (1, 256, 2, 64), (1, 1024, 1, 128)])
def test_attention_backward_with_and_without_explicit_transposes(B, T, h, d):
    """T % 64 == 0 sends dV = P^T dO and dK = dS^T Q through the transposed-operand GEMM (batched over (sequence, head), strided head
    slices as operands); T = 70 keeps the explicit transposes.  Gradients of the fused qkv buffer vs torch."""
    from modeling_utils import autograd as ag

    g = torch.Generator().manual_seed(14)
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    qkv = bf(torch.randn(B * T, 3 * h * d, generator=g)); dout = bf(torch.randn(B * T, h * d, generator=g))
    qt = qkv.clone().requires_grad_()
    q, k, v = (t.transpose(1, 2) for t in qt.view(B, T, 3, h, d).unbind(2))
    att = (torch.einsum("bhid,bhjd->bhij", q, k) * d**-0.5).softmax(-1)
    torch.einsum("bhij,bhjd->bhid", att, v).transpose(1, 2).reshape(B * T, h * d).backward(dout)
    qg = qkv.cuda().bfloat16().requires_grad_()
    ag.Attention.apply(qg, B, T, h, d, d**-0.5).backward(dout.cuda().bfloat16())
    got, want = qg.grad.float().cpu().view(B, T, 3, h, d), qt.grad.view(B, T, 3, h, d)
    for i, name in enumerate(("dq", "dk", "dv")):
        assert _rel(got[:, :, i], want[:, :, i]) < 3e-2, name


@pytest.mark.parametrize("B,T,h,chunk_seqs", [(2, 70, 2, 8), (3, 256, 3, 1), (2, 298, 2, 8), (3, 1024, 2, 2)])
def test_attention_backward_without_f32_score_tensors(B, T, h, chunk_seqs):
    """dim_head 384: the forward kernel hands over its base-2 log-sum-exp, the backward rebuilds P in the score GEMM's epilogue (ACT_EXP2) and
    dS in the dO V^T GEMM's epilogue (ACT_MUL_AUX, D = rowsum(dO * O) as the row bias) -- no f32 [B h, T, T] tensor, no softmax kernel.
    Against torch, and against the materialised path (FUSED_SOFTMAX = False) on the same inputs; T = 70 / 298 exercise the padded score
    columns (T_pad 128 / 320) and the scalar epilogue path, chunk_seqs < B several chunks with their row-bias offsets."""
    from modeling_utils import autograd as ag
    from tribe_hip import ops

    d = 384
    assert ops.attention_lse_supported(d)
    g = torch.Generator().manual_seed(15)
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    qkv = bf(torch.randn(B * T, 3 * h * d, generator=g) * 0.7); dout = bf(torch.randn(B * T, h * d, generator=g))
    qt = qkv.clone().requires_grad_()
    q, k, v = (t.transpose(1, 2) for t in qt.view(B, T, 3, h, d).unbind(2))
    att = (torch.einsum("bhid,bhjd->bhij", q, k) * d**-0.5).softmax(-1)
    torch.einsum("bhij,bhjd->bhid", att, v).transpose(1, 2).reshape(B * T, h * d).backward(dout)
    grads = {}
    keep = ag.Attention.CHUNK_BYTES, ag.Attention.CHUNK_BYTES_FUSED, ag.Attention.FUSED_SOFTMAX
    try:
        ag.Attention.CHUNK_BYTES = ag.Attention.CHUNK_BYTES_FUSED = chunk_seqs * h * T * ops.round_up(T, 64) * 4
        for fused in (True, False):
            ag.Attention.FUSED_SOFTMAX = fused
            qg = qkv.cuda().bfloat16().requires_grad_()
            out = ag.Attention.apply(qg, B, T, h, d, d**-0.5)
            assert (len(out.grad_fn.saved_tensors) == 3) == fused
            out.backward(dout.cuda().bfloat16())
            grads[fused] = qg.grad.float().cpu().view(B, T, 3, h, d)
    finally:
        ag.Attention.CHUNK_BYTES, ag.Attention.CHUNK_BYTES_FUSED, ag.Attention.FUSED_SOFTMAX = keep
    want = qt.grad.view(B, T, 3, h, d)
    for i, name in enumerate(("dq", "dk", "dv")):
        assert _rel(grads[True][:, :, i], want[:, :, i]) < 3e-2, name
        assert _rel(grads[True][:, :, i], grads[False][:, :, i]) < 2e-2, name
    assert torch.isfinite(grads[True]).all()


@pytest.mark.parametrize("M,K,N", [(256, 128, 192), (320, 256, 136), (192, 128, 100)])
def test_linear_weight_gradient_through_the_transposed_operand_gemm(M, K, N):
    """M % 64 == 0 with N, K >= 128 sends dW = dY^T X through tribe_gemm_desc.trans_ab (no explicit transposes; N = 136 pads dY's row
    stride to 192, N = 100 stays on the transpose route): same gradients as torch either way."""
    from modeling_utils import autograd as ag

    g = torch.Generator().manual_seed(13)
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    x = bf(torch.randn(M, K, generator=g)); w = bf(torch.randn(N, K, generator=g) / K**0.5); b = torch.randn(N, generator=g)
    dy = bf(torch.randn(M, N, generator=g))
    xt, wt, bt = (t.clone().requires_grad_() for t in (x, w, b))
    (xt @ wt.t() + bt).backward(dy)
    xg = x.cuda().bfloat16().requires_grad_(); wg, bg = (t.cuda().requires_grad_() for t in (w, b))
    ag.Linear.apply(xg, wg, bg, None, None, True).backward(dy.cuda())
    for name, got, want in (("dx", xg.grad.float(), xt.grad), ("dw", wg.grad, wt.grad), ("db", bg.grad, bt.grad)):
        assert _rel(got.cpu(), want) < 2e-2, name


def test_gelu_backward_epilogue():
    """act = GELU_BWD: out = (A B^T) * gelu'(pre) with the derivative from the forward's Phi polynomial + one exp2 (gemm_common.h gelu_grad2).
    A = identity rows pick single B entries, so the f32 output isolates the factor: |gelu' error| <= 1e-4 over pre in [-6, 6]."""
    from modeling_utils import autograd as ag
    from tribe_hip import _lib

    M = N = K = 256
    a = torch.eye(M, K).cuda().bfloat16()
    b = torch.ones(N, K).cuda().bfloat16()            # (A B^T)[m, n] = 1
    pre = torch.linspace(-6, 6, M * N).view(M, N)
    pre_bf = pre.to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    ag._gemm(a, b, out, lda=K, ldb=K, ldc=N, M=M, N=N, K=K, act=_lib.ACT_GELU_BWD, aux=pre_bf.cuda())
    x = pre_bf.double()
    want = 0.5 * (1 + torch.erf(x / 2**0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
    assert float((out.cpu().double() - want).abs().max()) < 1e-4


def test_scalenorm_fork_sums_both_branches_in_its_backward():
    """ScaleNormFork + a residual consumer with raw_res_grad=True == the plain pre-norm residual block y = Linear(norm(x)) + x * rs."""
    from modeling_utils import autograd as ag

    g = torch.Generator().manual_seed(9)
    M, D = 70, 768
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    x = torch.randn(M, D, generator=g); gpar = torch.tensor([1.1]); rs = torch.rand(D, generator=g) + 0.5
    w = bf(torch.randn(D, D, generator=g) / D**0.5); dy = torch.randn(M, D, generator=g)
    xt, gt, rst, wt = (t.clone().requires_grad_() for t in (x, gpar, rs, w))
    xn = xt / xt.norm(dim=-1, keepdim=True).clamp(min=1e-12) * D**0.5 * gt
    (xn.to(torch.bfloat16).float() @ wt.t() + xt * rst).backward(dy)
    xg, gg, rsg, wg = (t.cuda().requires_grad_() for t in (x, gpar, rs, w))
    yn, xr = ag.ScaleNormFork.apply(xg, gg, D**0.5, 1e-12, rsg)
    ag.Linear.apply(yn, wg, None, xr, rsg, True, True).backward(dy.cuda())
    for name, got, want in (("dx", xg.grad, xt.grad), ("dg", gg.grad, gt.grad), ("drs", rsg.grad, rst.grad), ("dw", wg.grad, wt.grad)):
        assert _rel(got.cpu(), want) < 2e-2, name


@pytest.mark.parametrize("V", [20, 23])   # C * V % 4 == 0: one batched GEMM + ordered device scatter sum; otherwise the per-sample chain
def test_subject_head_gradients_with_repeated_subjects(V):
    """SubjectLayers (modeling_utils/models/common.py:60-76 in the reference) backward when several samples share a subject: dW[s] is
    the sum over those samples, bias likewise; launch to launch the result is bit-identical (fixed summation order, no atomics)."""
    from modeling_utils import autograd as ag

    g = torch.Generator().manual_seed(5)
    B, T, C, S = 7, 40, 64, 3
    subjects = torch.tensor([2, 0, 2, 1, 2, 0, 2])
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    x = bf(torch.randn(B, T, C, generator=g)); w = bf(torch.randn(S, C, V, generator=g) / C**0.5); bias = torch.randn(S, V, generator=g)
    dy = bf(torch.randn(B, V, T, generator=g))
    xt, wt, bt = (t.clone().requires_grad_() for t in (x, w, bias))
    (torch.einsum("btc,bcv->bvt", xt, wt[subjects]) + bt[subjects][:, :, None]).backward(dy)
    grads = []
    for _ in range(2):
        xg = x.cuda().bfloat16().requires_grad_(); wg, bg = (t.cuda().requires_grad_() for t in (w, bias))
        y = ag.VoxelHead.apply(xg, wg, bg, subjects.cuda())
        y.backward(dy.cuda())
        grads.append((xg.grad.float().cpu(), wg.grad.cpu(), bg.grad.cpu()))
    for name, got, want in zip(("dx", "dw", "db"), grads[0], (xt.grad, wt.grad, bt.grad)):
        assert _rel(got, want) < 2e-2, name
    assert torch.equal(grads[0][1], grads[1][1])


@pytest.mark.parametrize("M,N,dtype,with_b", [(16384, 3072, torch.float32, False), (4099, 768, torch.float32, True), (1000, 12288, torch.bfloat16, False),
                                              (77, 1001, torch.float32, True), (5, 8, torch.bfloat16, False)])
def test_column_sums(M, N, dtype, with_b):
    """tribe_colsum_fwd (bias gradients, res_scale gradients): the four-columns-per-lane kernel and, for N % 4 != 0, the scalar one."""
    from modeling_utils.autograd import colsum

    g = torch.Generator().manual_seed(8)
    a = torch.randn(M, N, generator=g).to(dtype)
    b = torch.randn(M, N, generator=g) if with_b else None
    want = (a.double() * (b.double() if with_b else 1.0)).sum(0)
    got = colsum(a.cuda(), M, N, b=b.cuda() if with_b else None).cpu().double()
    torch.testing.assert_close(got, want, rtol=1e-4, atol=2e-3 * M**0.5)


@pytest.mark.parametrize("M,N,with_b", [(16384, 3072, True), (1000, 768, False), (77, 100, True)])
def test_gradient_sums_and_cast_in_one_pass(M, N, with_b):
    """tribe_colsum_cast_fwd: bias gradient, res_scale gradient and the bf16 GEMM operand from one pass over an f32 gradient."""
    from modeling_utils.autograd import grad_sums_and_cast

    g = torch.Generator().manual_seed(15)
    dy = torch.randn(M, N, generator=g)
    res = torch.randn(M, N, generator=g) if with_b else None
    sa, sab, bf = grad_sums_and_cast(dy.cuda(), M, N, res=res.cuda() if with_b else None, want_sum=True)
    torch.testing.assert_close(sa.cpu().double(), dy.double().sum(0), rtol=1e-4, atol=2e-3 * M**0.5)
    if with_b:
        torch.testing.assert_close(sab.cpu().double(), (dy.double() * res.double()).sum(0), rtol=1e-4, atol=2e-3 * M**0.5)
    else:
        assert sab is None
    assert torch.equal(bf.cpu(), dy.bfloat16())


def test_slab_scatter_sum_is_the_ordered_sum():
    """tribe_slab_scatter_sum: dst[idx[b]] += src[b] for b in order == the same loop on the host, bit for bit."""
    from tribe_hip._lib import check, lib

    g = torch.Generator().manual_seed(6)
    B, n, S = 9, 4 * 331, 4
    src = torch.randn(B, n, generator=g) * torch.logspace(-3, 3, B)[:, None]      # magnitudes that make the order matter
    idx = torch.tensor([1, 3, 1, 1, 0, 3, 1, 0, 1])
    want = torch.zeros(S, n)
    for b in range(B):
        want[idx[b]] += src[b]
    dst, src_d, idx_d = torch.zeros(S, n, device="cuda"), src.cuda(), idx.cuda()
    check(lib().tribe_slab_scatter_sum(src_d.data_ptr(), B, n, idx_d.data_ptr(), dst.data_ptr(), torch.cuda.current_stream().cuda_stream),
          "tribe_slab_scatter_sum")
    assert torch.equal(dst.cpu(), want)
    assert lib().tribe_slab_scatter_sum(src_d.data_ptr(), B, 6, idx_d.data_ptr(), dst.data_ptr(), 0) != 0   # n % 4 != 0 is refused


@pytest.mark.parametrize("cfg_kw", [{}, {"layer_aggregation": "mean", "subject_embedding": True}, {"feature_aggregation": "sum"}])
def test_training_step_gradients_vs_oracle(cfg_kw):
    """BrainModule.training_step + loss.backward(): loss and every parameter gradient vs the fp32 CPU oracle graph."""
    from algonauts2025.model import FmriEncoderConfig
    from algonauts2025.pl_module import BrainModule
    from data_utils.dataloader import SegmentData
    from modeling_utils.losses import TorchLossConfig

    fdims = {"text": (2, 40), "audio": (2, 24), "video": (2, 33)}
    V, Tout, S, B, T = 50, 10, 3, 4, 31
    dims = tribe_ref.EncoderDims(hidden=768, depth=2, heads=4)
    ref = tribe_ref.FmriEncoderRef(fdims, V, Tout, S, layer_aggregation=cfg_kw.get("layer_aggregation", "cat"),
                                   feature_aggregation=cfg_kw.get("feature_aggregation", "cat"),
                                   subject_embedding=cfg_kw.get("subject_embedding", False), dims=dims).train()
    with torch.no_grad():
        tribe_ref.fill_params_(ref, seed=2)
    model = FmriEncoderConfig(n_subjects=S, hidden=768, depth=2, heads=4, **cfg_kw).build(fdims, V, Tout)
    model.load_state_dict(ref.state_dict())
    model = model.cuda().train()
    data = tribe_ref.synthetic_batch(B, T, fdims, S, seed=4)
    fmri = torch.randn(B, V, Tout, generator=torch.Generator().manual_seed(9))
    # oracle
    loss_ref, *_ = tribe_ref.run_step(ref(data), fmri, data["subject_id"])
    loss_ref.backward()
    # HIP
    bm = BrainModule(model, TorchLossConfig(name="MSELoss").build(), None, {})
    batch = SegmentData(data={**{k: v.cuda() for k, v in data.items()}, "fmri": fmri.cuda()}, segments=[None] * B)
    loss = bm.training_step(batch, 0)
    assert loss.requires_grad and abs(float(loss) - float(loss_ref)) < 2e-3 * max(1.0, float(loss_ref))
    loss.backward()
    ref_grads = dict(ref.named_parameters())
    worst = {}
    for name, p in model.named_parameters():
        want = ref_grads[name].grad
        if want is None:
            continue
        assert p.grad is not None, f"no gradient for {name}"
        worst[name] = _rel(p.grad.cpu(), want)
    bad = {k: v for k, v in worst.items() if v > 6e-2}
    assert not bad, f"gradient mismatch: {bad}"
    print("max grad rel err", max(worst.values()), "over", len(worst), "tensors")


def test_pearson_loss_and_infonce_gradients():
    from modeling_utils import autograd as ag
    from modeling_utils.losses import PearsonLoss

    g = torch.Generator().manual_seed(3)
    B, V, T = 3, 17, 40
    pred = torch.randn(B, V, T, generator=g); true = 0.4 * pred + torch.randn(B, V, T, generator=g)
    for reduction in ("mean", "sum"):
        pt = pred.clone().requires_grad_()
        tribe_ref.pearson_loss(tribe_ref.flatten_bt(pt), tribe_ref.flatten_bt(true), reduction).backward()
        pg = pred.cuda().requires_grad_()
        loss = PearsonLoss(reduction).forward_bvt(pg, true.cuda())
        loss.backward()
        torch.testing.assert_close(pg.grad.cpu(), pt.grad, rtol=2e-4, atol=1e-6)
    # InfoNCE (model.py:208-221) incl. the row normalisation
    N, H = 150, 128
    q = torch.randn(N, H, generator=g); k = 0.7 * q + torch.randn(N, H, generator=g)
    qt, kt = q.clone().requires_grad_(), k.clone().requires_grad_()
    want = tribe_ref.info_nce(qt[None], kt[None], 0.07)
    want.backward()
    qg, kg = q.cuda().requires_grad_(), k.cuda().requires_grad_()
    got = ag.InfoNCE.apply(qg, kg, 0.07)
    got.backward()
    assert abs(float(got.detach()) - float(want.detach())) < 2e-2 * abs(float(want.detach()))  # bf16-rounded unit vectors, logits / 0.07
    assert _rel(qg.grad.cpu(), qt.grad) < 5e-2 and _rel(kg.grad.cpu(), kt.grad) < 5e-2


def test_training_step_with_contrastive_branch():
    """defaults.py:101-105 enables the InfoNCE alignment with the video modality: loss = mse + 0.1 * info_nce."""
    from algonauts2025.model import FmriEncoderConfig
    from algonauts2025.pl_module import BrainModule
    from data_utils.dataloader import SegmentData
    from modeling_utils.losses import TorchLossConfig

    fdims = {"text": (2, 40), "audio": (2, 24), "video": (2, 33)}
    V, Tout, S, B, T = 50, 10, 3, 4, 31
    dims = tribe_ref.EncoderDims(hidden=768, depth=2, heads=4)
    ref = tribe_ref.FmriEncoderRef(fdims, V, Tout, S, contrastive_modalities=["video"], dims=dims).train()
    with torch.no_grad():
        tribe_ref.fill_params_(ref, seed=2)
    model = FmriEncoderConfig(n_subjects=S, hidden=768, depth=2, heads=4, contrastive_enabled=True).build(fdims, V, Tout)
    model.load_state_dict(ref.state_dict())
    model = model.cuda().train()
    data = tribe_ref.synthetic_batch(B, T, fdims, S, seed=4)
    fmri = torch.randn(B, V, Tout, generator=torch.Generator().manual_seed(9))
    mse, *_ = tribe_ref.run_step(ref(data), fmri, data["subject_id"])
    nce = ref.compute_contrastive_loss(data)["video"]
    loss_ref = mse + 0.1 * nce  # pl_module.py:58-77
    loss_ref.backward()
    bm = BrainModule(model, TorchLossConfig(name="MSELoss").build(), None, {})
    batch = SegmentData(data={**{k: v.cuda() for k, v in data.items()}, "fmri": fmri.cuda()}, segments=[None] * B)
    loss = bm.training_step(batch, 0)
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-2 * float(loss_ref.detach())
    loss.backward()
    ref_grads = dict(ref.named_parameters())
    worst = {n: _rel(p.grad.cpu(), ref_grads[n].grad) for n, p in model.named_parameters() if ref_grads[n].grad is not None}
    assert "contrastive_heads.video.weight" in worst
    bad = {k: v for k, v in worst.items() if v > 8e-2}
    assert not bad, f"gradient mismatch: {bad}"
    print("contrastive: max grad rel err", max(worst.values()))


def test_contrastive_pass_shares_the_prediction_latents():
    """with equal dropout draws the contrastive pass takes the prediction pass's latents instead of re-running the
    encoder (model.py:228).  The loss must be BIT-identical to the two-pass form, the gradients equal up to the order of one f32 sum, and
    a differing draw, changed inputs or stepped parameters must fall back to recomputation."""
    from algonauts2025.model import FmriEncoderConfig
    from algonauts2025.pl_module import BrainModule
    from data_utils.dataloader import SegmentData
    from modeling_utils.losses import TorchLossConfig

    fdims = {"text": (2, 40), "audio": (2, 24), "video": (2, 33)}
    V, Tout, S, B, T = 50, 10, 3, 4, 31
    data = tribe_ref.synthetic_batch(B, T, fdims, S, seed=4)
    fmri = torch.randn(B, V, Tout, generator=torch.Generator().manual_seed(9))
    batch = SegmentData(data={**{k: v.cuda() for k, v in data.items()}, "fmri": fmri.cuda()}, segments=[None] * B)

    def run(share: bool, dropout: float, seed: int):
        torch.manual_seed(0)
        model = FmriEncoderConfig(n_subjects=S, hidden=768, depth=2, heads=4, contrastive_enabled=True, modality_dropout=dropout,
                                  share_contrastive_latents=share).build(fdims, V, Tout).cuda().train()
        bm = BrainModule(model, TorchLossConfig(name="MSELoss").build(), None, {})
        torch.manual_seed(seed)
        np.random.seed(seed)
        loss = bm.training_step(batch, 0)
        loss.backward()
        return model, loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    m1, l1, g1 = run(True, 0.0, 1)
    m0, l0, g0 = run(False, 0.0, 1)
    assert getattr(m1, "shared_latent_hits", 0) == 1 and getattr(m0, "shared_latent_hits", 0) == 0
    assert torch.equal(l1, l0), (float(l1), float(l0))
    assert g1.keys() == g0.keys()
    worst = max(_rel(g1[k].cpu(), g0[k].cpu()) for k in g1)
    assert worst < 2e-2, worst   # one backward through the encoder for the summed gradient vs two summed at the parameters (bf16 dgrad operands)
    # dropout 0.5: find a seed whose two draws differ and one whose draws coincide; sharing must follow the draws, the loss never changes
    seen = set()
    for seed in range(2, 40):
        ms, ls, _ = run(True, 0.5, seed)
        mn, ln, _ = run(False, 0.5, seed)
        assert torch.equal(ls, ln), seed
        seen.add(getattr(ms, "shared_latent_hits", 0))
        if seen == {0, 1}:
            break
    assert seen == {0, 1}, seen
    # a parameter written between the two passes invalidates the kept latents
    torch.manual_seed(0)
    model = FmriEncoderConfig(n_subjects=S, hidden=768, depth=2, heads=4, contrastive_enabled=True).build(fdims, V, Tout).cuda().train()
    model(batch)
    with torch.no_grad():
        model.time_pos_embed.mul_(1.0)
    model.compute_contrastive_loss(batch)
    assert getattr(model, "shared_latent_hits", 0) == 0


def test_hip_adam_matches_torch_adam():
    """One launch per group vs torch.optim.Adam on the CPU (oracle), under a OneCycleLR schedule that moves lr AND beta1
    every step (grids/defaults.py:126-141); ragged sizes exercise the chunking and the unaligned tail."""
    from modeling_utils.optim import HipAdam

    torch.manual_seed(0)
    shapes = [(3072, 257), (5,), (1, 1024, 48), (16385,), (7, 3)]
    for wd, decoupled in [(0.0, False), (0.01, False), (0.01, True)]:
        ref = [torch.randn(s).requires_grad_() for s in shapes]
        mine = [p.detach().clone().cuda().requires_grad_() for p in ref]
        opt_ref = (torch.optim.AdamW if decoupled else torch.optim.Adam)(ref, lr=1e-3, weight_decay=wd)
        opt = HipAdam(mine, lr=1e-3, weight_decay=wd, decoupled_weight_decay=decoupled)
        sch_ref = torch.optim.lr_scheduler.OneCycleLR(opt_ref, max_lr=1e-2, total_steps=6, pct_start=0.3)
        sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-2, total_steps=6, pct_start=0.3)
        for step in range(5):
            for a, b in zip(ref, mine):
                g = torch.randn(a.shape) * (10.0 ** (step - 2))
                a.grad, b.grad = g.clone(), g.clone().cuda()
            opt_ref.step(), opt.step()
            sch_ref.step(), sch.step()
            assert opt.param_groups[0]["lr"] == opt_ref.param_groups[0]["lr"] and opt.param_groups[0]["betas"] == opt_ref.param_groups[0]["betas"]
            for a, b in zip(ref, mine):
                torch.testing.assert_close(b.detach().cpu(), a.detach(), rtol=2e-6, atol=2e-7)
        sd = opt.state_dict()["state"][0]
        assert set(sd) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["step"]) == 5.0
        torch.testing.assert_close(sd["exp_avg_sq"].cpu(), opt_ref.state_dict()["state"][0]["exp_avg_sq"], rtol=3e-5, atol=1e-12)
        assert all(p._version >= 5 for p in mine)                 # raw-pointer updates are visible to version-keyed caches
    cpu_param = torch.zeros(3, requires_grad=True)
    cpu_param.grad = torch.ones(3)
    with pytest.raises(Exception):
        HipAdam([cpu_param], lr=1e-3).step()                      # no CPU fallback


def test_hip_adam_refreshes_the_bf16_operand_copies_of_the_weights():
    """A weight that went through ag.Linear / ag.QKVLinear has a bf16 shadow (its packed GEMM operand); HipAdam's kernel writes it in the
    same pass as the f32 update, so after the step the cached pack IS bf16(w) -- same buffer, no cast pass -- and its transpose follows."""
    from modeling_utils import autograd as ag
    from modeling_utils.optim import HipAdam

    torch.manual_seed(3)
    lin = torch.nn.Linear(128, 192, bias=False).cuda()
    wq, wk, wv = (torch.nn.Parameter(torch.randn(64, 128, device="cuda") * 0.1) for _ in range(3))
    opt = HipAdam([lin.weight, wq, wk, wv], lr=1e-2)
    x = torch.randn(256, 128, device="cuda").bfloat16()
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        y = ag.Linear.apply(x, lin.weight, None, None, None, True).sum() + ag.QKVLinear.apply(x, wq, wk, wv).float().sum()
        y.backward()
        pack_before, fused_before = ag.PACKS.get(lin.weight)[0], ag.QKV_PACKS.get((wq, wk, wv))[0]
        opt.step()
        pack, pack_t = ag.PACKS.get(lin.weight)
        assert pack.data_ptr() == pack_before.data_ptr()                       # refreshed in place by the optimiser kernel, not re-packed
        assert torch.equal(pack, lin.weight.detach().bfloat16()) and torch.equal(pack_t[:, :192], pack.t())
        fused, fused_t = ag.QKV_PACKS.get((wq, wk, wv))
        assert fused.data_ptr() == fused_before.data_ptr()
        assert torch.equal(fused, torch.cat([wq, wk, wv]).detach().bfloat16()) and torch.equal(fused_t[:, :192], fused.t())
    with torch.no_grad():
        lin.weight.mul_(2.0)                                                     # any other in-place write: the pack is rebuilt from the f32 values
    assert torch.equal(ag.PACKS.get(lin.weight)[0], lin.weight.detach().bfloat16())


def test_hip_adam_late_starting_parameter_keeps_its_own_step_count():
    """torch.optim.Adam counts steps per parameter: one whose grad was None on earlier steps (a projector whose modality was
    dropped under `zero_grad(set_to_none=True)`, model.py:133-141) starts its bias corrections at 1 when it first gets a gradient."""
    from modeling_utils.optim import HipAdam

    torch.manual_seed(1)
    shapes = [(257, 33), (1000,), (40, 7)]
    ref = [torch.randn(s).requires_grad_() for s in shapes]
    mine = [p.detach().clone().cuda().requires_grad_() for p in ref]
    opt_ref, opt = torch.optim.Adam(ref, lr=1e-2), HipAdam(mine, lr=1e-2)
    present = [[0, 2], [0, 2], [0, 1, 2], [1], [0, 1, 2]]        # parameter 1 joins at step 3; 0 and 2 sit out step 4
    for active in present:
        opt_ref.zero_grad(set_to_none=True), opt.zero_grad(set_to_none=True)
        for i in active:
            g = torch.randn(shapes[i])
            ref[i].grad, mine[i].grad = g.clone(), g.clone().cuda()
        opt_ref.step(), opt.step()
        for a, b in zip(ref, mine):
            torch.testing.assert_close(b.detach().cpu(), a.detach(), rtol=2e-6, atol=2e-7)
    assert [float(opt.state[p]["step"]) for p in mine] == [4.0, 3.0, 4.0]


def test_fit_with_modality_dropout_runs_and_matches_torch_adam():
    """Trainer.fit WITHOUT a GradReducer at modality_dropout = 0.5 (the reference trains with 0.3, defaults.py): projectors whose
    modality is dropped get grad None under zero_grad(set_to_none=True) and re-join later; before the per-parameter step counts
    HipAdam aborted the fit there.  The gradients every step hands to the optimiser are recorded and replayed through
    torch.optim.Adam from the same initial weights: identical inputs, so the optimisers must agree to rounding (comparing two
    separate fits instead is chaotic: q / k gradients of a freshly initialised encoder are ~1e-9, the size of Adam's eps)."""
    import numpy as np

    from algonauts2025.model import FmriEncoderConfig
    from algonauts2025.pl_module import BrainModule
    from algonauts2025.trainer import Trainer
    from data_utils.dataloader import SegmentData
    from modeling_utils.losses.base import TorchLossConfig
    from modeling_utils.optim import HipAdam

    fdims = {"text": (2, 24), "audio": (2, 16), "video": (2, 20)}
    B, T, V, S = 4, 24, 30, 2

    class Recording(HipAdam):
        def __init__(self, params, **kw):
            params = list(params)
            super().__init__(params, **kw)
            self.watched = params
            self.start = [p.detach().clone() for p in params]
            self.trace = []

        def step(self, closure=None):
            self.trace.append([None if p.grad is None else p.grad.detach().clone() for p in self.watched])
            return super().step(closure)

    class _Opt:
        def build(self, params, total_steps=None):
            return Recording(params, lr=1e-3)

    torch.manual_seed(0)
    np.random.seed(0)
    model = FmriEncoderConfig(n_subjects=S, hidden=768, depth=1, heads=2, modality_dropout=0.5).build(fdims, V, T)
    g = torch.Generator().manual_seed(3)
    data = {m: torch.randn(B, l, d, T, generator=g) for m, (l, d) in fdims.items()}
    data["subject_id"] = (torch.arange(B) % S).view(B, 1)
    data["fmri"] = torch.randn(B, V, T, generator=g)
    batch = SegmentData(data=data, segments=[None] * B)
    module = BrainModule(model, TorchLossConfig(name="MSELoss").build(), _Opt(), {}, max_epochs=1)
    torch.manual_seed(11)                                     # the dropout draws of the steps
    trainer = Trainer(max_epochs=1, reduce_gradients=False)
    trainer.fit(module, [batch] * 8)                          # raised "parameters of one group must share their step count" before
    opt = trainer.optimizers[0]
    assert len(opt.trace) == 8
    absent = [sum(gr is None for gr in step) for step in opt.trace]
    assert 0 < sum(absent) and min(absent) == 0, f"the draws must drop a modality on some steps only, got {absent}"
    steps = {float(opt.state[p]["step"]) for p in opt.watched}
    assert len(steps) > 1, "some projector should have taken fewer steps than the encoder"

    ref = [p.clone().requires_grad_() for p in opt.start]
    opt_ref = torch.optim.Adam(ref, lr=1e-3)
    for grads in opt.trace:
        for p, gr in zip(ref, grads):
            p.grad = None if gr is None else gr.clone()
        opt_ref.step()
    for p, q in zip(opt.watched, ref):
        torch.testing.assert_close(p.detach(), q.detach(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("Z,R,Cc,pad_c", [(3, 70, 130, 0), (2, 1024, 384, 0), (1, 33, 5, 0), (4, 64, 64, 64), (2, 257, 36, 12)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_transpose_bf16_vector_and_scalar_paths(Z, R, Cc, pad_c, dtype):
    """out[z, c, r] = bf16(in[z, r, c]) on a strided view (row stride Cc + pad_c), rows padded to 64: aligned views take the
    4-element vector path, odd strides the element path; both must be exact."""
    from modeling_utils.autograd import transpose_bf16

    g = torch.Generator().manual_seed(R)
    full = torch.randn(Z, R, Cc + pad_c, generator=g).to(dtype).cuda()
    got = transpose_bf16(full, Z, R, Cc, R * (Cc + pad_c), Cc + pad_c)
    Rp = (R + 63) // 64 * 64
    assert got.shape == (Z, Cc, Rp)
    want = full[:, :, :Cc].to(torch.bfloat16).transpose(1, 2)
    assert torch.equal(got[:, :, :R], want) and not got[:, :, R:].any()


def test_swa_update_matches_averaged_model_rule():
    """tribe_swa_update against torch.optim.swa_utils.AveragedModel (CPU): ragged sizes, four updates, then the copy-back."""
    from torch.optim.swa_utils import AveragedModel

    from algonauts2025.callbacks import StochasticWeightAveraging

    class Bag(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(3)
            self.ps = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(s, generator=g)) for s in [(3072, 129), (5,), (16385,), (2, 1024, 9)]])

    ref, mine = Bag(), Bag().cuda()
    avg_ref = AveragedModel(ref)
    swa = StochasticWeightAveraging(swa_lrs=1e-5, swa_epoch_start=0.6, annealing_epochs=2)
    g = torch.Generator().manual_seed(4)
    for n in range(4):
        with torch.no_grad():
            for a, b in zip(ref.ps, mine.ps):
                d = torch.randn(a.shape, generator=g)
                a.add_(d), b.add_(d.cuda())
        avg_ref.update_parameters(ref)
        swa.update_average(mine)
        assert swa.n_averaged == n + 1 == int(avg_ref.n_averaged)
        for a, b in zip(avg_ref.module.ps, swa.averages):
            torch.testing.assert_close(b.cpu(), a.detach(), rtol=1e-6, atol=1e-6)
    with pytest.raises(Exception):
        StochasticWeightAveraging(swa_lrs=1e-5).update_average(Bag())            # CPU parameters: no fallback


def test_fit_loop_onecycle_then_swa():
    """Trainer.fit with the reference's optimiser block (defaults.py:126-141) and SWA settings (main.py:365-373) on a small model:
    the LR trajectory is torch's OneCycleLR per step, then SWALR per epoch from int(0.6 * n_epochs); the final weights are the mean
    of the weights seen at the start of the averaged epochs; the loss goes down."""
    from torch.optim.swa_utils import SWALR

    from algonauts2025.callbacks import StochasticWeightAveraging
    from algonauts2025.model import FmriEncoderConfig
    from algonauts2025.pl_module import BrainModule
    from algonauts2025.trainer import Trainer
    from data_utils.dataloader import SegmentData
    from modeling_utils.losses import TorchLossConfig
    from modeling_utils.optim import HipAdam
    from modeling_utils.optimizers import LightningOptimizerConfig

    fdims = {"text": (2, 40), "audio": (2, 24), "video": (2, 33)}
    V, Tout, S, B, T = 50, 10, 3, 4, 31
    n_epochs, n_batches = 5, 3
    model = FmriEncoderConfig(n_subjects=S, hidden=768, depth=1, heads=4, modality_dropout=0.0).build(fdims, V, Tout)
    optim = LightningOptimizerConfig(optimizer={"name": "Adam", "lr": 1e-4, "kwargs": {"weight_decay": 0.0}},
                                     scheduler={"name": "OneCycleLR", "kwargs": {"max_lr": 1e-3, "pct_start": 0.1}})
    bm = BrainModule(model, TorchLossConfig(name="MSELoss").build(), optim, {}, max_epochs=n_epochs)
    batches = []
    for i in range(n_batches):
        data = tribe_ref.synthetic_batch(B, T, fdims, S, seed=20 + i)
        fmri = torch.randn(B, V, Tout, generator=torch.Generator().manual_seed(30 + i))
        batches.append(SegmentData(data={**data, "fmri": fmri}, segments=[None] * B))

    class Probe:                                                 # after the SWA callback in the list: sees what it averaged
        def __init__(self):
            self.lrs, self.snaps = [], {}

        def on_train_epoch_start(self, trainer, module):
            self.lrs.append(trainer.optimizers[0].param_groups[0]["lr"])
            self.snaps[trainer.current_epoch] = [p.detach().clone() for p in module.parameters()]

    annealing = int(n_epochs * (1 - 0.6))
    swa = StochasticWeightAveraging(swa_epoch_start=0.6, annealing_epochs=annealing, swa_lrs=1e-5, annealing_strategy="cos")
    probe = Probe()
    trainer = Trainer(max_epochs=n_epochs, callbacks=[swa, probe])
    trainer.fit(bm, batches)
    assert isinstance(trainer.optimizers[0], HipAdam)
    assert (swa.swa_start, swa.swa_end, swa.n_averaged) == (3, 4, 2)

    # the same schedule on a dummy torch optimiser
    dummy = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
    sched = {"s": torch.optim.lr_scheduler.OneCycleLR(dummy, max_lr=1e-3, pct_start=0.1, total_steps=n_epochs * n_batches), "per": "step"}
    want_lrs = []
    for epoch in range(n_epochs):
        if epoch == 3:
            sched = {"s": SWALR(dummy, swa_lr=1e-5, anneal_epochs=annealing, anneal_strategy="cos"), "per": "epoch"}
        want_lrs.append(dummy.param_groups[0]["lr"])
        for _ in range(n_batches):
            dummy.step()
            if sched["per"] == "step":
                sched["s"].step()
        if sched["per"] == "epoch":
            sched["s"].step()
    assert probe.lrs == pytest.approx(want_lrs, rel=1e-12)
    for p, a, b in zip(bm.parameters(), probe.snaps[3], probe.snaps[4]):
        torch.testing.assert_close(p.detach(), (a + b) / 2, rtol=1e-6, atol=1e-7)
    losses = [h["train/loss"] for h in trainer.history]
    assert losses[2] < losses[0], losses
    # the averaged weights are what the next forward uses (packed-weight caches refreshed through the version counters)
    bm.eval()
    with torch.no_grad():
        again = bm(batches[0].to("cuda"))
    assert torch.isfinite(again).all()


def test_fit_with_bucketed_gradients_equals_plain_fit():
    """GradReducer on the GPU (world of one: no collective, but gradients live in the flat buckets, HipAdam steps from the views and
    zero_grad is a memset per bucket): same weights as the plain loop up to the atomics' summation order."""
    from algonauts2025.model import FmriEncoderConfig
    from algonauts2025.pl_module import BrainModule
    from algonauts2025.trainer import Trainer
    from data_utils.dataloader import SegmentData
    from modeling_utils.losses import TorchLossConfig

    fdims = {"text": (2, 40), "audio": (2, 24), "video": (2, 33)}
    V, Tout, S, B, T = 50, 10, 3, 4, 31
    batches = []
    for i in range(2):
        data = tribe_ref.synthetic_batch(B, T, fdims, S, seed=40 + i)
        batches.append(SegmentData(data={**data, "fmri": torch.randn(B, V, Tout, generator=torch.Generator().manual_seed(50 + i))}, segments=[None] * B))
    finals = []
    for bucketed in (False, True):
        torch.manual_seed(7)
        model = FmriEncoderConfig(n_subjects=S, hidden=768, depth=2, heads=4, contrastive_enabled=True).build(fdims, V, Tout)
        bm = BrainModule(model, TorchLossConfig(name="MSELoss").build(), None, {})
        trainer = Trainer(max_epochs=2, reduce_gradients=bucketed, bucket_bytes=8 << 20)
        trainer.fit(bm, batches)
        finals.append([p.detach().cpu() for p in bm.parameters()])
    for a, b in zip(*finals):
        assert _rel(b, a) < 1e-4


def _ddp_batches(rank: int | None, world: int):
    from data_utils.dataloader import SegmentData

    fdims = {"text": (2, 40), "audio": (2, 24), "video": (2, 33)}
    V, Tout, S, B, T = 50, 10, 3, 4, 31
    out = []
    for i in range(2):
        data = tribe_ref.synthetic_batch(B, T, fdims, S, seed=60 + i)
        data["fmri"] = torch.randn(B, V, Tout, generator=torch.Generator().manual_seed(70 + i))
        if rank is not None:                                     # rank r of G takes sequences r::G
            data = {k: v[rank::world].contiguous() for k, v in data.items()}
        out.append(SegmentData(data=data, segments=[None] * next(iter(data.values())).shape[0]))
    return fdims, V, Tout, S, out


def _ddp_fit(rank: int | None, world: int):
    from algonauts2025.model import FmriEncoderConfig
    from algonauts2025.pl_module import BrainModule
    from algonauts2025.trainer import Trainer
    from modeling_utils.losses import TorchLossConfig
    from modeling_utils.optimizers import LightningOptimizerConfig

    fdims, V, Tout, S, batches = _ddp_batches(rank, world)
    torch.manual_seed(11)                                        # same initial weights on every rank
    model = FmriEncoderConfig(n_subjects=S, hidden=768, depth=2, heads=4).build(fdims, V, Tout)
    optim = LightningOptimizerConfig(optimizer={"name": "SGD", "lr": 0.05})    # linear in the gradient: a clean equivalence check
    bm = BrainModule(model, TorchLossConfig(name="MSELoss").build(), optim, {})
    trainer = Trainer(max_epochs=1, bucket_bytes=4 << 20)
    trainer.fit(bm, batches)
    return trainer, {n: p.detach().float().cpu() for n, p in bm.named_parameters()}


def _ddp_worker(rank: int, world: int, port: int, q):
    import os

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)                                     # rehearsal: both ranks share the one GPU, gloo carries the exchange
    import datetime

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        trainer, weights = _ddp_fit(rank, world)
        q.put((rank, {k: v.numpy() for k, v in weights.items()}))
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_fit_matches_single_process():
    """One process per rank, sequences r::2, GradReducer averaging the bucketed gradients (gloo between two processes on this one GPU;
    RCCL on a real node): after two steps every rank holds the weights of the single-process run on the whole batch."""
    import socket

    import torch.multiprocessing as mp

    _, want = _ddp_fit(None, 1)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        results = {}
        deadline = time.monotonic() + 300
        while len(results) < len(procs):
            dead = [p for p in procs if p.exitcode not in (None, 0)]
            assert not dead, f"rank process exited with code {dead[0].exitcode}"
            assert time.monotonic() < deadline, "two-rank fit timed out"
            try:
                rank, weights = q.get(timeout=1.0)
                results[rank] = weights
            except queue.Empty:
                pass
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:                                                      # never leave an orphan holding the GPU inside a gloo collective
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
                if p.is_alive():
                    p.kill()
    assert sorted(results) == [0, 1]
    for rank, got in results.items():
        worst = max(_rel(torch.from_numpy(got[n]), want[n]) for n in want)
        assert worst < 2e-4, (rank, worst)
