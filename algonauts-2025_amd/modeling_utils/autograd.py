"""Autograd functions of the training path: every forward AND backward is a HIP kernel launch (MFMA GEMMs for the
dense gradients, streaming kernels for the rest); torch.autograd only orders the calls and owns the saved tensors.

Conventions: activations that feed a GEMM are bf16 `[M, K]`; the residual stream and all parameter gradients are f32.
For Y = X W^T (X [M, K], W [N, K]):
    dX = dY W          -> NT GEMM with A = dY [M, N],   B = W^T [K, N]   (W^T packed once per weight version)
    dW = dY^T X        -> the same kernel with transposed operands (desc.trans_ab: dY and X read as they lie; `wgrad` below), or, for
                          shapes outside it, NT GEMM with A = dY^T [N, M], B = X^T [K, M] after tribe_transpose_bf16
    db = column sums of dY.
Reference semantics followed: pl_module.py:126-128 (training_step = _run_step loss -> Lightning backward),
model.py:113-174 (forward graph), x_transformers encoder (oracle/xt_encoder.py).
"""

from __future__ import annotations

import ctypes as C
import typing as tp

import torch

from tribe_hip import _lib, ops
from tribe_hip._lib import BF16, F32, GemmDesc, check, lib

_DT = {torch.float32: F32, torch.bfloat16: BF16}


_LOG2E = 1.4426950408889634


def _s() -> int:
    return torch.cuda.current_stream().cuda_stream


def _gemm(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, *, lda=None, ldb=None, ldc=None, M=None, N=None, K=None, alpha=1.0,
          bias=None, act=_lib.ACT_NONE, aux=None, res=None, ldres=None, res_scale=None, batch1=1, batch0=1, sA=(0, 0), sB=(0, 0), sC=(0, 0),
          gather1=None, gather_a=False, gather_b=False, a_off=0, b_off=0, c_off=0, role=0, trans_ab=False, row_bias=None, row_bias_off=0,
          sBias=(0, 0), ld_aux=None, stream_k=False) -> None:
    """Thin positional wrapper over tribe_gemm_bf16 (element offsets allow strided views without copies).
    stream_k (trans_ab only): let the launcher cut the last partial round of tiles over all CUs (workspace taken from ops.workspace)."""
    d = GemmDesc()
    d.trans_ab = int(trans_ab)
    d.stream_k = int(stream_k)
    d.M, d.N, d.K, d.batch1, d.batch0 = M, N, K, batch1, batch0
    d.A, d.lda, d.sA1, d.sA0 = a.data_ptr() + 2 * a_off, lda, sA[0], sA[1]
    d.B, d.ldb, d.sB1, d.sB0 = b.data_ptr() + 2 * b_off, ldb, sB[0], sB[1]
    d.C, d.ldc, d.sC1, d.sC0 = out.data_ptr() + out.element_size() * c_off, ldc, sC[0], sC[1]
    d.c_dtype, d.alpha, d.act, d.role = _DT[out.dtype], alpha, act, role
    if bias is not None:
        d.bias, d.bias_mode = bias.data_ptr(), _lib.BIAS_COL
    if row_bias is not None:   # f32 per output row, batch strides sBias (elements)
        d.bias, d.bias_mode, d.sBias1, d.sBias0 = row_bias.data_ptr() + 4 * row_bias_off, _lib.BIAS_ROW, sBias[0], sBias[1]
    if aux is not None:
        d.aux, d.ld_aux = aux.data_ptr(), (ld_aux if ld_aux is not None else N)
    if res is not None:
        d.res, d.ldres = res.data_ptr() + 4 * c_off, (ldres if ldres is not None else ldc)
        d.sRes1, d.sRes0 = sC
    if res_scale is not None:
        d.res_scale = res_scale.data_ptr()
    if gather1 is not None:
        d.gather1, d.gather_a, d.gather_b = gather1.data_ptr(), int(gather_a), int(gather_b)
    if stream_k:
        nbytes = lib().tribe_gemm_stream_k_workspace_bytes(C.byref(d))
        if nbytes > 0:   # parts of split tiles travel through the workspace; a second launch sums them in order
            ws = ops.workspace(nbytes, out.device, tag="streamk")
            d.stream_k_ws, d.stream_k_ws_bytes = ws.data_ptr(), ws.numel() * ws.element_size()
    check(lib().tribe_gemm_bf16(C.byref(d), _s()), "tribe_gemm_bf16")


def transpose_bf16(x: torch.Tensor, Z: int, R: int, Cc: int, s_z: int, s_r: int, off: int = 0) -> torch.Tensor:
    """out[z, c, r] = x[z, r, c] as bf16 [Z, C, R_pad64] (x f32 or bf16, strided view given by element strides / offset)."""
    R_pad = ops.round_up(R, 64)
    out = torch.empty(Z, Cc, R_pad, dtype=torch.bfloat16, device=x.device)
    check(lib().tribe_transpose_bf16(x.data_ptr() + x.element_size() * off, _DT[x.dtype], Z, R, Cc, s_z, s_r, out.data_ptr(), Cc * R_pad, R_pad,
                                     _s()), "tribe_transpose_bf16")
    return out


def grad_sums_and_cast(dy: torch.Tensor, M: int, N: int, res: torch.Tensor | None = None, want_sum: bool = False,
                       want_bf16: bool = True) -> tuple[torch.Tensor | None, torch.Tensor | None, torch.Tensor | None]:
    """One pass over an incoming f32 gradient dy [M, N]: (column sums or None, column sums of dy * res or None, bf16 copy or None) --
    the bias gradient, the res_scale gradient and the GEMM operand of a Linear / FeedForward backward (tribe_colsum_cast_fwd)."""
    dev = dy.device
    if dy.dtype != torch.float32 or N % 4:
        return (colsum(dy, M, N) if want_sum else None, colsum(dy, M, N, b=res) if res is not None else None,
                cast_bf16(dy) if want_bf16 else None)
    sa = torch.empty(N, dtype=torch.float32, device=dev) if want_sum else None
    sab = torch.empty(N, dtype=torch.float32, device=dev) if res is not None else None
    bf = torch.empty(M, N, dtype=torch.bfloat16, device=dev) if want_bf16 else None
    check(lib().tribe_colsum_cast_fwd(dy.data_ptr(), ops._p(res), M, N, N, ops._p(sa), ops._p(sab), ops._p(bf), N, _s()), "tribe_colsum_cast_fwd")
    return sa, sab, bf


# weight gradients through the stream-K schedule of the transposed-operand GEMM (tribe_gemm_desc.stream_k): dW of a 3072 x 3072 layer is 144
# tiles on 256 CUs, of a 12288 x 3072 one 576 = 2.25 rounds.  Split tiles are summed in a fixed order: bit-reproducible.
STREAM_K_WGRAD = True


def wgrad(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, M: int, N: int, K: int, ld_dy: int, ld_x: int) -> None:
    """dw[n, k] = sum_m dy[m, n] x[m, k]  (dy bf16 [M, ld_dy >= N], x bf16 [M, ld_x >= K], dw f32 [N, K]): the weight gradient of a Linear.
    Shapes the 256^2 kernel covers go through its transposed-operand form (desc.trans_ab: dy and x are read as they lie, the LDS reads
    transpose); the rest keep the two explicit bf16 transposes in front of the NT GEMM."""
    if M % 64 == 0 and N % 8 == 0 and K % 8 == 0 and N >= 128 and K >= 128 and ld_dy % 8 == 0 and ld_x % 8 == 0:
        _gemm(dy, x, dw, lda=ld_dy, ldb=ld_x, ldc=K, M=N, N=K, K=M, trans_ab=True, stream_k=STREAM_K_WGRAD)
        return
    dy_t = transpose_bf16(dy, 1, M, N, 0, ld_dy)[0]   # [N, M_pad]
    x_t = transpose_bf16(x, 1, M, K, 0, ld_x)[0]      # [K, M_pad]
    Mp = dy_t.shape[1]
    _gemm(dy_t, x_t, dw, lda=Mp, ldb=Mp, ldc=K, M=N, N=K, K=Mp)


def colsum(a: torch.Tensor, M: int, N: int, b: torch.Tensor | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    acc = out is not None
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=a.device)
    check(lib().tribe_colsum_fwd(a.data_ptr(), _DT[a.dtype], ops._p(b), M, N, N, out.data_ptr(), int(acc), _s()), "tribe_colsum_fwd")
    return out


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    if x.dtype == torch.bfloat16:
        return x.contiguous()
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib().tribe_cast_bf16_fwd(x.data_ptr(), x.numel(), y.data_ptr(), _s()), "tribe_cast_bf16_fwd")
    return y


class _WeightPacks:
    """bf16 W [N, K_pad] and W^T [K, N_pad] per (tensor object, version) of an f32 weight; entries die with the tensor
    (temporaries are rebuilt every step, parameters only when they were updated).  An unpadded pack (K % 64 == 0: a plain element-wise
    cast) is registered as the parameter's bf16 shadow (modeling_utils/shadow.py): HipAdam's kernel then writes it while it updates the
    parameter and `_refreshed` records the new version -- no cast pass after the step; W^T is re-derived from the bf16 pack."""

    def __init__(self) -> None:
        self._c: dict[int, list] = {}     # id(w) -> [weakref, sig, pack, packT]

    def _refreshed(self, key: int, w_ref: tp.Any, version: int) -> None:
        hit, w = self._c.get(key), w_ref()
        if hit is None or w is None or hit[0]() is not w:
            return
        hit[1] = (w.data_ptr(), version, tuple(w.shape))
        N, K = w.shape
        hit[3] = transpose_bf16(hit[2], 1, N, K, 0, hit[2].shape[1])[0]     # [K, N_pad64] from the fresh bf16 pack

    def get(self, w: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        import weakref

        from . import shadow

        key = id(w)
        sig = (w.data_ptr(), w._version, tuple(w.shape))
        hit = self._c.get(key)
        if hit is None or hit[0]() is not w or hit[1] != sig:
            wd = w.detach().contiguous()
            N, K = wd.shape
            ref = weakref.ref(w, lambda _r, k=key: self._c.pop(k, None))
            hit = [ref, sig, ops.pack_weight(wd), transpose_bf16(wd, 1, N, K, 0, K)[0]]
            self._c[key] = hit
            if hit[2].shape == wd.shape and w.is_contiguous() and isinstance(w, torch.nn.Parameter):
                shadow.register(w, hit[2], lambda version, k=key, r=ref: self._refreshed(k, r, version))
        return hit[2], hit[3]


PACKS = _WeightPacks()


# ------------------------------------------------------------------------------------------------------------
class Linear(torch.autograd.Function):
    """y = x W^T + b (+ res * res_scale).  x bf16 [M, K]; W f32 [N, K]; y bf16 or f32 [M, N]."""

    @staticmethod
    def forward(ctx, x, w, b, res, res_scale, out_f32: bool, raw_res_grad: bool = False):
        ctx.raw_res_grad = raw_res_grad   # the residual came from ScaleNormFork, which applies res_scale to its gradient itself
        M, K = x.shape
        N = w.shape[0]
        wp, _ = PACKS.get(w)
        Kp = wp.shape[1]
        if Kp != K:
            raise ValueError(f"Linear: activation width {K} must equal the packed K {Kp} (pad activations to 64)")
        y = torch.empty(M, N, dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
        _gemm(x, wp, y, lda=K, ldb=Kp, ldc=N, M=M, N=N, K=K, bias=b, res=res, res_scale=res_scale)
        ctx.save_for_backward(x, w, res, res_scale)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, res, res_scale = ctx.saved_tensors
        M, K = x.shape
        N = w.shape[0]
        dy = dy.contiguous()
        dres = drs = None
        if res is not None:
            if ctx.raw_res_grad:
                dres = dy
            else:
                dres = torch.empty_like(res)
                check(lib().tribe_scale_cols_fwd(dy.data_ptr(), ops._p(res_scale), M, N, dres.data_ptr(), _s()), "tribe_scale_cols_fwd")
        # bias gradient, res_scale gradient and the bf16 GEMM operand from ONE pass over dy
        db, drs, dpre = grad_sums_and_cast(dy, M, N, res=res if (res is not None and res_scale is not None) else None, want_sum=ctx.has_b)
        _, wt = PACKS.get(w)  # [K, N_pad64]
        Np = wt.shape[1]
        if Np != N:
            dpre = torch.nn.functional.pad(dpre, (0, Np - N))  # zero K-padding for the dgrad GEMM (N % 64 != 0 only)
        dx = torch.empty(M, K, dtype=x.dtype, device=x.device)   # x is bf16: autograd would cast an f32 gradient in a pass of its own
        _gemm(dpre, wt, dx, lda=Np, ldb=Np, ldc=K, M=M, N=K, K=Np)
        # dW[n, k] = sum_m dpre[m, n] x[m, k]
        dw = torch.empty(N, K, dtype=torch.float32, device=x.device)
        wgrad(dpre, x, dw, M, N, K, Np, K)
        dw = dw[:, : w.shape[1]] if w.shape[1] != K else dw
        return dx, dw, db, dres, drs, None, None


class _FusedQKVPacks:
    """bf16 [3N, K] (q | k | v rows) and its transpose [K, 3N] for one attention block, rebuilt only when one of the three f32
    weights was written: each weight is packed STRAIGHT into its row slice -- no f32 `torch.cat` of the three matrices per layer and
    step (113 MB written + read back, plus the split in backward), which model.py:228 of round 1 did.  The three row slices are the
    parameters' bf16 shadows (see _WeightPacks): after a HipAdam step only the transpose is re-derived."""

    def __init__(self) -> None:
        self._c: dict[int, list] = {}     # id(wq) -> [weakref, sig, fused, fusedT, [weak refs of the three weights]]

    def _refreshed(self, key: int, index: int, version: int) -> None:
        """HipAdam wrote the bf16 rows of weight `index` together with its f32 value, which now stands at `version`.  Only THAT
        weight's signature entry moves: a sibling written outside the optimiser (load_state_dict, copy_) that got no gradient in this
        step keeps its recorded version, still mismatches in get() and is repacked there."""
        hit = self._c.get(key)
        if hit is None:
            return
        ws = [r() for r in hit[4]]
        if any(w is None for w in ws) or hit[0]() is not ws[0]:
            return
        sig = list(hit[1])
        sig[index] = (ws[index].data_ptr(), version, tuple(ws[index].shape))
        hit[1] = tuple(sig)
        hit[3] = None                                                            # the transpose is re-derived on the next get()

    def get(self, ws: tuple[torch.Tensor, torch.Tensor, torch.Tensor]) -> tuple[torch.Tensor, torch.Tensor]:
        import weakref

        from . import shadow

        key = id(ws[0])
        sig = tuple((w.data_ptr(), w._version, tuple(w.shape)) for w in ws)
        hit = self._c.get(key)
        if hit is None or hit[0]() is not ws[0] or hit[1] != sig:
            N, K = ws[0].shape
            if any(tuple(w.shape) != (N, K) for w in ws) or K % 64:
                raise ValueError("fused q|k|v projection: the three weights must share one [N, K] shape with K % 64 == 0")
            fused = hit[2] if (hit is not None and hit[0]() is ws[0] and hit[2].shape == (3 * N, K)) else \
                torch.empty(3 * N, K, dtype=torch.bfloat16, device=ws[0].device)
            for i, w in enumerate(ws):
                wd = w.detach().contiguous()
                check(lib().tribe_pack_weight_bf16(wd.data_ptr(), N, K, K, fused[i * N:(i + 1) * N].data_ptr(), N, K, _s()), "tribe_pack_weight_bf16")
            hit = [weakref.ref(ws[0], lambda _r, k=key: self._c.pop(k, None)), sig, fused, None, [weakref.ref(w) for w in ws]]
            self._c[key] = hit
            for i, w in enumerate(ws):
                if isinstance(w, torch.nn.Parameter) and w.is_contiguous():
                    shadow.register(w, fused[i * N:(i + 1) * N], lambda version, k=key, i=i: self._refreshed(k, i, version))
        if hit[3] is None:
            N3, K = hit[2].shape
            hit[3] = transpose_bf16(hit[2], 1, N3, K, 0, K)[0]
        return hit[2], hit[3]


QKV_PACKS = _FusedQKVPacks()


class QKVLinear(torch.autograd.Function):
    """qkv = x [Wq; Wk; Wv]^T as ONE GEMM (x bf16 [M, K] -> bf16 [M, 3N]); backward: dx = dqkv W and ONE wgrad GEMM whose
    [3N, K] result is handed back as three row slices (views: no copy)."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv):
        M, K = x.shape
        N = wq.shape[0]
        wp, _ = QKV_PACKS.get((wq, wk, wv))
        y = torch.empty(M, 3 * N, dtype=torch.bfloat16, device=x.device)
        _gemm(x, wp, y, lda=K, ldb=K, ldc=3 * N, M=M, N=3 * N, K=K, role=2)
        ctx.save_for_backward(x, wq, wk, wv)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wq, wk, wv = ctx.saved_tensors
        M, K = x.shape
        N = wq.shape[0]
        _, wt = QKV_PACKS.get((wq, wk, wv))                     # [K, 3N_pad64]
        dpre = cast_bf16(dy.contiguous())
        N3, Np = 3 * N, wt.shape[1]
        if Np != N3:
            dpre = torch.nn.functional.pad(dpre, (0, Np - N3))
        dx = torch.empty(M, K, dtype=x.dtype, device=x.device)   # x is bf16: autograd would cast an f32 gradient in a pass of its own
        _gemm(dpre, wt, dx, lda=Np, ldb=Np, ldc=K, M=M, N=K, K=Np)
        dw = torch.empty(N3, K, dtype=torch.float32, device=x.device)
        wgrad(dpre, x, dw, M, N3, K, Np, K)
        return dx, dw[:N], dw[N:2 * N], dw[2 * N:]


class FeedForward(torch.autograd.Function):
    """out = gelu(x W1^T + b1) W2^T + b2 + res * res_scale  (x_transformers FeedForward + scaled residual).
    x bf16 [M, D]; res f32 [M, D] (the block input); out f32 [M, D]."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, res, res_scale, raw_res_grad: bool = False):
        ctx.raw_res_grad = raw_res_grad   # see Linear
        M, D = x.shape
        Fh = w1.shape[0]
        w1p, _ = PACKS.get(w1)
        w2p, _ = PACKS.get(w2)
        pre = torch.empty(M, Fh, dtype=torch.bfloat16, device=x.device)
        h = torch.empty(M, Fh, dtype=torch.bfloat16, device=x.device)
        _gemm(x, w1p, h, lda=D, ldb=D, ldc=Fh, M=M, N=Fh, K=D, bias=b1, act=_lib.ACT_GELU, aux=pre, role=6)
        out = torch.empty(M, D, dtype=torch.float32, device=x.device)
        _gemm(h, w2p, out, lda=Fh, ldb=Fh, ldc=D, M=M, N=D, K=Fh, bias=b2, res=res, res_scale=res_scale, role=7)
        ctx.save_for_backward(x, w1, w2, pre, h, res, res_scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w1, w2, pre, h, res, res_scale = ctx.saved_tensors
        M, D = x.shape
        Fh = w1.shape[0]
        dout = dout.contiguous()
        if ctx.raw_res_grad:
            dres = dout
        else:
            dres = torch.empty_like(res)
            check(lib().tribe_scale_cols_fwd(dout.data_ptr(), ops._p(res_scale), M, D, dres.data_ptr(), _s()), "tribe_scale_cols_fwd")
        db2, drs, dob = grad_sums_and_cast(dout, M, D, res=res if res_scale is not None else None, want_sum=True)   # one pass over dout
        _, w2t = PACKS.get(w2)                                  # [Fh, D]
        dpre = torch.empty(M, Fh, dtype=torch.bfloat16, device=x.device)
        _gemm(dob, w2t, dpre, lda=D, ldb=D, ldc=Fh, M=M, N=Fh, K=D, act=_lib.ACT_GELU_BWD, aux=pre)   # dh * gelu'(pre)
        dw2 = torch.empty(D, Fh, dtype=torch.float32, device=x.device)
        wgrad(dob, h, dw2, M, D, Fh, D, Fh)
        db1 = colsum(dpre, M, Fh)
        _, w1t = PACKS.get(w1)                                  # [D, Fh]
        dx = torch.empty(M, D, dtype=x.dtype, device=x.device)   # as in Linear.backward
        _gemm(dpre, w1t, dx, lda=Fh, ldb=Fh, ldc=D, M=M, N=D, K=Fh)
        dw1 = torch.empty(Fh, D, dtype=torch.float32, device=x.device)
        wgrad(dpre, x, dw1, M, Fh, D, Fh, D)
        return dx, dw1, db1, dw2, db2, dres, drs, None


class ScaleNorm(torch.autograd.Function):
    """y = x / max(||x||, eps) * gain_scale * g  (bf16 out; x f32 [M, D])."""

    @staticmethod
    def forward(ctx, x, g, gain_scale: float, eps: float, out_f32: bool = False):
        y = ops.scalenorm(x, g, gain_scale, eps, torch.float32 if out_f32 else torch.bfloat16)
        ctx.save_for_backward(x, g)
        ctx.gs, ctx.eps = gain_scale, eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        dy = dy.contiguous()
        M, D = x.shape
        dx = torch.empty_like(x)
        dg = torch.zeros(1, dtype=torch.float32, device=x.device)
        check(lib().tribe_scalenorm_bwd(x.data_ptr(), dy.data_ptr(), _DT[dy.dtype], g.data_ptr(), ctx.gs, ctx.eps, M, D, None, None,
                                        dx.data_ptr(), dg.data_ptr(), _s()), "tribe_scalenorm_bwd")
        return dx, dg, None, None, None


class ScaleNormFork(torch.autograd.Function):
    """(xn, xr) = (ScaleNorm(x), x): the input of a pre-norm block forked into its normed branch and its residual branch (x_transformers
    `Residual(scale_residual=True)`: block(norm(x)) + x * res_scale).  The backward adds the two incoming gradients INSIDE the ScaleNorm
    backward kernel, dx = d_norm + d_res * res_scale: the consumer of `xr` (Linear / FeedForward with raw_res_grad=True) hands its output
    gradient back as it is, so the per-block column scaling pass and autograd's add of the two branches (two f32 [M, D] passes each) go."""

    @staticmethod
    def forward(ctx, x, g, gain_scale: float, eps: float, res_scale):
        y = ops.scalenorm(x, g, gain_scale, eps, torch.bfloat16)
        ctx.save_for_backward(x, g, res_scale)
        ctx.gs, ctx.eps = gain_scale, eps
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dxr):
        x, g, rs = ctx.saved_tensors
        M, D = x.shape
        dx = torch.empty_like(x)
        dg = torch.zeros(1, dtype=torch.float32, device=x.device)
        if dy is None:    # the normed branch was not used: only the residual gradient flows
            dx = dxr if rs is None else dxr * rs
            return dx, None, None, None, None
        dy = dy.contiguous()
        dres = None if dxr is None else dxr.float().contiguous()
        check(lib().tribe_scalenorm_bwd(x.data_ptr(), dy.data_ptr(), _DT[dy.dtype], g.data_ptr(), ctx.gs, ctx.eps, M, D, ops._p(dres),
                                        None if dres is None else ops._p(rs), dx.data_ptr(), dg.data_ptr(), _s()), "tribe_scalenorm_bwd")
        return dx, dg, None, None, None


class Rotary(torch.autograd.Function):
    """In-place partial rotary on the q and k heads of a fused qkv buffer (returns the same storage)."""

    @staticmethod
    def forward(ctx, qkv, cos, sin, T: int, heads: int, dim_head: int, rot_dim: int, interleaved: bool):
        ctx.mark_dirty(qkv)
        ops.rotary_(qkv, T, heads, dim_head, rot_dim, cos, sin, interleaved)
        ctx.save_for_backward(cos, sin)
        ctx.meta = (T, heads, dim_head, rot_dim, interleaved)
        return qkv

    @staticmethod
    def backward(ctx, dqkv):
        cos, sin = ctx.saved_tensors
        T, heads, dim_head, rot_dim, interleaved = ctx.meta
        d = dqkv.contiguous().clone() if not dqkv.is_contiguous() else dqkv.clone()
        ops.rotary_(d, T, heads, dim_head, rot_dim, cos, (-sin).contiguous(), interleaved)  # rotation by -theta = transpose
        return d, None, None, None, None, None, None, None


class Attention(torch.autograd.Function):
    """softmax(q k^T * scale) v per (batch, head) on a fused bf16 qkv [B*T, 3*inner]; forward = fused flash kernel,
    backward = five MFMA GEMM products per head on P / dS recomputed per batch chunk.  Where the forward kernel can hand over its
    log-sum-exp (dim_head 384) the backward keeps no f32 [B h, T, T] tensor and runs no softmax kernel: the score GEMM's epilogue writes
    P = exp2(scale log2e q.k - lse2[row]) (ACT_EXP2), the dO V^T GEMM's epilogue writes dS = scale (dP - D[row]) * P (ACT_MUL_AUX) with
    D = rowsum(dO * O) (FlashAttention-2's identity for rowsum(P * dP)); 12 bytes of HBM traffic per score instead of 28.  Other head
    sizes (and FUSED_SOFTMAX = False) take the materialised S / dP + softmax kernels."""

    # f32 score bytes per chunk of sequences.  Same box, B = 16 (scripts/train_bench.py attn-chunk=N): 7 sequences (256 MiB) 112.0 ms,
    # 4 sequences 110.9, 2 sequences 117.4, 1 sequence 122.5 -- smaller chunks keep more of S / P / dP / dS in the Infinity Cache but
    # leave the batched GEMMs under one round of tiles
    CHUNK_BYTES = 144 << 20
    # With the fused-softmax backward only P and dS (bf16) exist and fewer, larger batched GEMMs win: B = 16 in one chunk 102.8 ms, 8 sequences
    # 103.7, 4 sequences 104.5 (materialised path, 4 sequences: 105.8; same box, profiles/r03_z5_train.txt)
    CHUNK_BYTES_FUSED = 1 << 30
    FUSED_SOFTMAX = True

    @staticmethod
    def _fwd(ctx, qkv, B: int, T: int, heads: int, dim_head: int, scale: float):
        if Attention.FUSED_SOFTMAX and ops.attention_lse_supported(dim_head):
            out, lse = ops.attention_with_lse(qkv, B, T, heads, dim_head, scale)
            ctx.save_for_backward(qkv, out, lse)
        else:
            out = ops.attention(qkv, B, T, heads, dim_head, scale)
            ctx.save_for_backward(qkv)
        return out

    @staticmethod
    def forward(ctx, qkv, B: int, T: int, heads: int, dim_head: int, scale: float):
        out = Attention._fwd(ctx, qkv, B, T, heads, dim_head, scale)
        ctx.meta = (B, T, heads, dim_head, scale)
        ctx.rotary = None
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv = ctx.saved_tensors[0]
        fused = len(ctx.saved_tensors) == 3
        B, T, h, d, scale = ctx.meta
        inner, ld = h * d, 3 * h * d
        dev = qkv.device
        dout = cast_bf16(dout)
        dqkv = torch.empty_like(qkv)
        Tp = ops.round_up(T, 64)
        chunk = max(1, min(B, (Attention.CHUNK_BYTES_FUSED if fused else Attention.CHUNK_BYTES) // (h * T * Tp * 4)))
        if fused:
            _, out, lse = ctx.saved_tensors
            neg_lse = lse.neg()                                             # the row biases of the two epilogues, [B, h, T] f32
            neg_d = ops.rowdot_heads(dout, out, B, T, h, d, -scale)
            # (the GEMMs write columns < T only: the pad columns the K = T_pad products read stay zero)
            P = (torch.zeros if Tp != T else torch.empty)(chunk * h, T, Tp, dtype=torch.bfloat16, device=dev)
            dS = (torch.zeros if Tp != T else torch.empty)(chunk * h, T, Tp, dtype=torch.bfloat16, device=dev)
        else:
            S = torch.empty(chunk * h, T, Tp, dtype=torch.float32, device=dev)
            P = torch.empty(chunk * h, T, Tp, dtype=torch.bfloat16, device=dev)
            dP = torch.empty(chunk * h, T, Tp, dtype=torch.float32, device=dev)
            dS = torch.empty(chunk * h, T, Tp, dtype=torch.bfloat16, device=dev)
        st = _s()
        for b0 in range(0, B, chunk):
            nb = min(chunk, B - b0)
            Z = nb * h
            row0 = b0 * T
            q_off, k_off, v_off = row0 * ld, row0 * ld + inner, row0 * ld + 2 * inner
            if fused:
                # P = exp2(scale log2e Q K^T - lse2) ;  dS = (scale dO V^T - scale D) * P -- both straight out of the GEMM epilogues
                _gemm(qkv, qkv, P, lda=ld, ldb=ld, ldc=Tp, M=T, N=T, K=d, alpha=scale * _LOG2E, act=_lib.ACT_EXP2, batch1=nb, batch0=h,
                      sA=(T * ld, d), sB=(T * ld, d), sC=(h * T * Tp, T * Tp), a_off=q_off, b_off=k_off, row_bias=neg_lse, row_bias_off=b0 * h * T,
                      sBias=(h * T, T))
                _gemm(dout, qkv, dS, lda=inner, ldb=ld, ldc=Tp, M=T, N=T, K=d, alpha=scale, act=_lib.ACT_MUL_AUX, aux=P, ld_aux=Tp, batch1=nb,
                      batch0=h, sA=(T * inner, d), sB=(T * ld, d), sC=(h * T * Tp, T * Tp), a_off=row0 * inner, b_off=v_off, row_bias=neg_d,
                      row_bias_off=b0 * h * T, sBias=(h * T, T))
            else:
                # S = scale * Q K^T ;  P = softmax(S)
                _gemm(qkv, qkv, S, lda=ld, ldb=ld, ldc=Tp, M=T, N=T, K=d, alpha=scale, batch1=nb, batch0=h, sA=(T * ld, d), sB=(T * ld, d),
                      sC=(h * T * Tp, T * Tp), a_off=q_off, b_off=k_off)
                check(lib().tribe_softmax_fwd(S.data_ptr(), Z * T, T, Tp, P.data_ptr(), Tp, Tp, st), "tribe_softmax_fwd")
                # dP = dO V^T
                _gemm(dout, qkv, dP, lda=inner, ldb=ld, ldc=Tp, M=T, N=T, K=d, batch1=nb, batch0=h, sA=(T * inner, d), sB=(T * ld, d),
                      sC=(h * T * Tp, T * Tp), a_off=row0 * inner, b_off=v_off)
                # dS = P * (dP - rowsum(P dP)) * scale
                check(lib().tribe_softmax_bwd(P.data_ptr(), dP.data_ptr(), Z * T, T, Tp, Tp, Tp, scale, dS.data_ptr(), Tp, st), "tribe_softmax_bwd")
            # dQ[q, :] = sum_key dS[q, key] K[key, :]: the reduction runs along the rows of K -> per (b, h) transposed view [d, Tp]
            kT = _head_transpose(qkv, nb, h, T, d, ld, k_off)
            _gemm(dS, kT, dqkv, lda=Tp, ldb=Tp, ldc=ld, M=T, N=d, K=Tp, batch1=nb, batch0=h, sA=(h * T * Tp, T * Tp), sB=(h * d * Tp, d * Tp),
                  sC=(T * ld, d), c_off=q_off)
            if T % 64 == 0 and d % 8 == 0:
                # dV[key, :] = sum_q P[q, key] dO[q, :] and dK[key, :] = sum_q dS[q, key] Q[q, :]: both factors have the reduction index q on
                # their ROWS -- the transposed-operand form of the GEMM (desc.trans_ab) reads P / dS [T, Tp] and the dO / q head slices as
                # they lie: no P^T, dS^T ([B h, T, T] each), q^T, dO^T
                _gemm(P, dout, dqkv, lda=Tp, ldb=inner, ldc=ld, M=T, N=d, K=T, batch1=nb, batch0=h, sA=(h * T * Tp, T * Tp), sB=(T * inner, d),
                      sC=(T * ld, d), b_off=row0 * inner, c_off=v_off, trans_ab=True)
                _gemm(dS, qkv, dqkv, lda=Tp, ldb=ld, ldc=ld, M=T, N=d, K=T, batch1=nb, batch0=h, sA=(h * T * Tp, T * Tp), sB=(T * ld, d),
                      sC=(T * ld, d), b_off=q_off, c_off=k_off, trans_ab=True)
            else:
                qT = _head_transpose(qkv, nb, h, T, d, ld, q_off)
                doT = _head_transpose(dout, nb, h, T, d, inner, row0 * inner)
                PT = transpose_bf16(P, Z, T, T, T * Tp, Tp)
                dST = transpose_bf16(dS, Z, T, T, T * Tp, Tp)
                _gemm(PT, doT, dqkv, lda=Tp, ldb=Tp, ldc=ld, M=T, N=d, K=Tp, batch1=nb, batch0=h, sA=(h * T * Tp, T * Tp), sB=(h * d * Tp, d * Tp),
                      sC=(T * ld, d), c_off=v_off)
                _gemm(dST, qT, dqkv, lda=Tp, ldb=Tp, ldc=ld, M=T, N=d, K=Tp, batch1=nb, batch0=h, sA=(h * T * Tp, T * Tp), sB=(h * d * Tp, d * Tp),
                      sC=(T * ld, d), c_off=k_off)
        if ctx.rotary is not None:   # RotaryAttention: rotate dq, dk back in place -- dqkv is this function's own buffer
            cos, neg_sin, rot_dim, interleaved = ctx.rotary
            ops.rotary_(dqkv, T, h, d, rot_dim, cos, neg_sin, interleaved)
        return dqkv, None, None, None, None, None


class RotaryAttention(torch.autograd.Function):
    """Rotary (in place on the q and k heads of the fused qkv buffer) + attention as ONE node: the backward rotates dq, dk by -theta in the
    gradient buffer it allocated itself.  As two nodes (Rotary, Attention) the rotation had to work on a clone of the incoming gradient
    (302 MB copied per layer and step at B = 16) and negate the sine table every call."""

    @staticmethod
    def forward(ctx, qkv, cos, sin, neg_sin, B: int, T: int, heads: int, dim_head: int, scale: float, rot_dim: int, interleaved: bool):
        ctx.mark_dirty(qkv)
        ops.rotary_(qkv, T, heads, dim_head, rot_dim, cos, sin, interleaved)
        out = Attention._fwd(ctx, qkv, B, T, heads, dim_head, scale)
        ctx.meta = (B, T, heads, dim_head, scale)
        ctx.rotary = (cos, neg_sin, rot_dim, interleaved)
        return out, qkv

    @staticmethod
    def backward(ctx, dout, _dqkv_alias):
        return (Attention.backward(ctx, dout)[0],) + (None,) * 10


def _head_transpose(x: torch.Tensor, nb: int, h: int, T: int, d: int, ld: int, off: int) -> torch.Tensor:
    """[nb, T, (h, d)] strided view -> bf16 [nb*h, d, T_pad]: one launch, batch levels (sequence, head)."""
    Tp = ops.round_up(T, 64)
    out = torch.empty(nb * h, d, Tp, dtype=torch.bfloat16, device=x.device)
    check(lib().tribe_transpose_bf16_b2(x.data_ptr() + x.element_size() * off, _DT[x.dtype], nb, h, T, d, T * ld, d, ld, out.data_ptr(), d * Tp, Tp,
                                        _s()), "tribe_transpose_bf16_b2")
    return out


class VoxelHead(torch.autograd.Function):
    """SubjectLayers: y[b, v, t] = sum_c x[b, t, c] W[s_b, c, v] + bias[s_b, v].  x bf16 [B, T, C]; y f32 [B, V, T]."""

    @staticmethod
    def forward(ctx, x, w, bias, subjects):
        S, Cc, V = w.shape
        wp = ops.pack_subject_weights(w.detach().contiguous())
        y = ops.voxel_head(x, wp, None if bias is None else bias.detach().contiguous(), subjects, V)
        ctx.save_for_backward(x, w, subjects)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, subjects = ctx.saved_tensors
        B, T, Cp = x.shape
        S, Cc, V = w.shape
        dy = dy.contiguous()
        dev = x.device
        Vp, Tp = ops.round_up(V, 64), ops.round_up(T, 64)
        # dx[b, t, c] = sum_v dy[b, v, t] W[s_b, c, v]:  A = dy_b^T [T, Vp],  B = W[s_b] [C, Vp]
        dyT = ops.pack_features(dy, layer_mean=False, K_pad=Vp).view(B, T, Vp)            # "b v t -> b t v" + cast
        wv = ops.pack_weight(w.detach().reshape(S * Cc, V).contiguous(), cols_pad=Vp)     # [S*C, Vp]
        dx = torch.empty(B, T, Cp, dtype=torch.float32, device=dev)
        if Cp != Cc:
            dx.zero_()
        _gemm(dyT, wv, dx, lda=Vp, ldb=Vp, ldc=Cp, M=T, N=Cc, K=Vp, batch1=B, sA=(T * Vp, 0), sB=(Cc * Vp, 0), sC=(T * Cp, 0),
              gather1=subjects, gather_b=True)
        # dW[s, c, v] += sum_t x[b, t, c] dy[b, v, t]  for every sample b of subject s:  A = x_b^T [C, Tp], B = dy_b [V, Tp]
        xT = transpose_bf16(x, B, T, Cc, T * Cp, Cp)                                        # [B, C, Tp]
        dyb = ops.pack_weight(dy.reshape(B * V, T), cols_pad=Tp)                            # [B*V, Tp] bf16
        # one batched launch for the B per-sample products, then the samples of a subject summed in sample order on the device: no
        # device -> host read of `subjects` (it stalled the launch queue once per step) and B / 256-CU-filling tiles instead of B serial
        # 32-tile GEMMs chained through dW
        dw = torch.zeros(S, Cc, V, dtype=torch.float32, device=dev)
        if (Cc * V) % 4 == 0:
            dwb = torch.empty(B, Cc, V, dtype=torch.float32, device=dev)
            _gemm(xT, dyb, dwb, lda=Tp, ldb=Tp, ldc=V, M=Cc, N=V, K=Tp, batch1=B, sA=(Cc * Tp, 0), sB=(V * Tp, 0), sC=(Cc * V, 0))
            check(lib().tribe_slab_scatter_sum(dwb.data_ptr(), B, Cc * V, subjects.data_ptr(), dw.data_ptr(), _s()), "tribe_slab_scatter_sum")
        else:
            for b, s in enumerate(subjects.tolist()):
                _gemm(xT, dyb, dw, lda=Tp, ldb=Tp, ldc=V, M=Cc, N=V, K=Tp, a_off=b * Cc * Tp, b_off=b * V * Tp, c_off=s * Cc * V, res=dw,
                      ldres=V)
        db = None
        if ctx.has_bias:
            db = torch.zeros(S, V, dtype=torch.float32, device=dev)
            check(lib().tribe_rowsum_scatter(dy.data_ptr(), B, V, T, subjects.data_ptr(), db.data_ptr(), _s()), "tribe_rowsum_scatter")
        return dx, dw, db, None


class AdaptivePool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t_out: int):
        ctx.t_in, ctx.t_out = x.shape[-1], t_out
        return ops.adaptive_avg_pool(x.contiguous(), t_out)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        rows = dy.numel() // ctx.t_out
        dx = torch.empty(*dy.shape[:-1], ctx.t_in, dtype=torch.float32, device=dy.device)
        check(lib().tribe_adaptive_avg_pool_bwd(dy.data_ptr(), rows, ctx.t_in, ctx.t_out, dx.data_ptr(), _s()), "tribe_adaptive_avg_pool_bwd")
        return dx, None


class MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, true):
        pred, true = pred.contiguous(), true.contiguous()
        ctx.save_for_backward(pred, true)
        return ops.mse(pred, true)

    @staticmethod
    def backward(ctx, g):
        pred, true = ctx.saved_tensors
        dp = torch.empty_like(pred)
        g = g.reshape(1).to(torch.float32).contiguous()
        check(lib().tribe_mse_bwd(pred.data_ptr(), true.data_ptr(), pred.numel(), g.data_ptr(), dp.data_ptr(), _s()), "tribe_mse_bwd")
        return dp, None


class PearsonLossFn(torch.autograd.Function):
    """PearsonLoss over the '(b t) d' view of strided [B, V, T] tensors (losses.py:17-42)."""

    @staticmethod
    def forward(ctx, pred, true, reduction: str):
        out = ops.pearson_loss(pred, true, reduction)
        ctx.save_for_backward(pred, true)
        ctx.reduction = reduction
        return out

    @staticmethod
    def backward(ctx, g):
        pred, true = ctx.saved_tensors
        B, V, T = pred.shape
        sb, sv, st = pred.stride()
        stats = torch.zeros(1, V, 6, dtype=torch.float64, device=pred.device)
        ops.pearson_stats_update(stats, pred, true)
        dp = torch.empty(B, V, T, dtype=torch.float32, device=pred.device)
        g = g.reshape(1).to(torch.float32).contiguous()
        check(lib().tribe_pearson_loss_bwd(pred.data_ptr(), true.data_ptr(), B, V, T, sb, sv, st, stats.data_ptr(), int(ctx.reduction == "sum"),
                                           g.data_ptr(), dp.data_ptr(), _s()), "tribe_pearson_loss_bwd")
        if (sb, sv, st) != (V * T, T, 1):  # gradient in the layout of the (strided) input view
            dp = dp.as_strided((B, V, T), (V * T, T, 1))
            out = torch.empty_strided((B, V, T), (sb, sv, st), dtype=torch.float32, device=pred.device)
            out.copy_(dp)
            dp = out
        return dp, None, None


class InfoNCE(torch.autograd.Function):
    """Symmetric InfoNCE of model.py:208-221 on two [N, H] f32 latent matrices (rows are L2-normalised here)."""

    @staticmethod
    def forward(ctx, q, k, tau: float):
        N, H = q.shape
        dev = q.device
        one = torch.ones(1, dtype=torch.float32, device=dev)
        qn = ops.scalenorm(q.contiguous(), one, 1.0, 1e-12, torch.bfloat16)   # F.normalize(dim=-1)
        kn = ops.scalenorm(k.contiguous(), one, 1.0, 1e-12, torch.bfloat16)
        S = torch.empty(N, N, dtype=torch.float32, device=dev)
        St = torch.empty(N, N, dtype=torch.float32, device=dev)
        _gemm(qn, kn, S, lda=H, ldb=H, ldc=N, M=N, N=N, K=H, alpha=1.0 / tau)
        _gemm(kn, qn, St, lda=H, ldb=H, ldc=N, M=N, N=N, K=H, alpha=1.0 / tau)   # logits^T: its row LSE = column LSE of logits
        lse_r, lse_c, diag = (torch.empty(N, dtype=torch.float32, device=dev) for _ in range(3))
        check(lib().tribe_lse_rows_fwd(S.data_ptr(), N, N, lse_r.data_ptr(), diag.data_ptr(), _s()), "tribe_lse_rows_fwd")
        check(lib().tribe_lse_rows_fwd(St.data_ptr(), N, N, lse_c.data_ptr(), None, _s()), "tribe_lse_rows_fwd")
        del St
        loss = 0.5 * ((lse_r - diag).mean() + (lse_c - diag).mean())  # two length-N vector reductions (host glue)
        ctx.save_for_backward(q, k, qn, kn, S, lse_r, lse_c)
        ctx.tau = tau
        return loss

    @staticmethod
    def backward(ctx, g):
        q, k, qn, kn, S, lse_r, lse_c = ctx.saved_tensors
        N, H = q.shape
        dev = q.device
        Np = ops.round_up(N, 64)
        gs = (g.reshape(1).to(torch.float32) / ctx.tau).contiguous()   # d logits / d (q.k) = 1 / tau
        dL = torch.empty(N, Np, dtype=torch.bfloat16, device=dev)
        check(lib().tribe_infonce_dlogits(S.data_ptr(), N, N, lse_r.data_ptr(), lse_c.data_ptr(), gs.data_ptr(), dL.data_ptr(), Np, _s()),
              "tribe_infonce_dlogits")
        knT = transpose_bf16(kn, 1, N, H, 0, H)[0]   # [H, Np]
        qnT = transpose_bf16(qn, 1, N, H, 0, H)[0]
        dLT = transpose_bf16(dL, 1, N, N, 0, Np)[0]  # [N, Np]
        dqn = torch.empty(N, H, dtype=torch.float32, device=dev)
        dkn = torch.empty(N, H, dtype=torch.float32, device=dev)
        _gemm(dL, knT, dqn, lda=Np, ldb=Np, ldc=H, M=N, N=H, K=Np)    # dq^ = dL k^
        _gemm(dLT, qnT, dkn, lda=Np, ldb=Np, ldc=H, M=N, N=H, K=Np)   # dk^ = dL^T q^
        one = torch.ones(1, dtype=torch.float32, device=dev)
        dq, dk = torch.empty_like(q), torch.empty_like(k)
        for x, dy, dx in ((q, dqn, dq), (k, dkn, dk)):
            check(lib().tribe_scalenorm_bwd(x.data_ptr(), dy.data_ptr(), F32, one.data_ptr(), 1.0, 1e-12, N, H, None, None, dx.data_ptr(), None,
                                            _s()), "tribe_scalenorm_bwd")
        return dq, dk, None


class ProjectorFuse(torch.autograd.Function):
    """One modality's projector writing its column slice of the fused stream (+ positional / subject embeddings are
    added by EmbedAdd).  feat bf16 [M, Kp]; W f32 [N, K]; out slice f32 [M, N] (a fresh tensor; concat is done by cat)."""

    @staticmethod
    def forward(ctx, feat, w, b):
        M, Kp = feat.shape
        N = w.shape[0]
        wp, _ = PACKS.get(w)
        y = torch.empty(M, N, dtype=torch.float32, device=feat.device)
        _gemm(feat, wp, y, lda=Kp, ldb=Kp, ldc=N, M=M, N=N, K=Kp, bias=b, role=1)
        ctx.save_for_backward(feat, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        feat, w = ctx.saved_tensors
        M, Kp = feat.shape
        N, K = w.shape
        dy = dy.contiguous()
        dyb = cast_bf16(dy)
        dw = torch.empty(N, Kp, dtype=torch.float32, device=feat.device)
        wgrad(dyb, feat, dw, M, N, Kp, N, Kp)
        return None, dw[:, :K].contiguous() if Kp != K else dw, colsum(dy, M, N)


_ = tp
