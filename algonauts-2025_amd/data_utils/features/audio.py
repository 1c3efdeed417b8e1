"""Wav2Vec-BERT 2.0 audio feature extractor on MI355X HIP kernels.

Mirror of the reference plugin `Wav2VecBert` (/root/reference/data_utils/data_utils/features/audio.py:27-263): a 30-60 s
waveform chunk is resampled to 16 kHz, z-scored and turned into 160-dim filterbank features by the HF SeamlessM4T
feature extractor (audio.py:222-234, host side, out of scope like the reference's julius / soundfile IO);
`Wav2Vec2BertModel(features, output_hidden_states=True)` (audio.py:253-263) yields 25 hidden states `[T@50Hz, 1024]`,
which are resampled to 2 Hz by nearest-neighbour `F.interpolate` (audio.py:163-171) -> `[25, 1024, 2*duration]`.

Here the conformer forward and the resampling run in one C call (`tribe_w2vbert_fwd`): LayerNorm kernels, bf16 MFMA GEMMs
with fused swish / GLU / half-step-residual epilogues, flash attention with the "relative_key" position bias, a fused
causal-depthwise-conv + LayerNorm + swish kernel, and a row gather for the nearest-neighbour resampling.
"""

from __future__ import annotations

import ctypes as C
import typing as tp

import torch

from tribe_hip import ops
from tribe_hip._lib import ConformerLayer, W2vBertDesc, check, lib

# facebook/w2v-bert-2.0 hyper-parameters (public model card; configuration input, not verifiable offline)
W2V_BERT_2 = dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                  feature_projection_input_dim=160, hidden_act="swish", position_embeddings_type="relative_key",
                  left_max_position_embeddings=64, right_max_position_embeddings=8, conv_depthwise_kernel_size=31,
                  layer_norm_eps=1e-5)


def nearest_index(t_in: int, t_out: int) -> torch.Tensor:
    """Source index of F.interpolate(mode='nearest') along the last axis: floor(i * t_in / t_out) (audio.py:163-171)."""
    scale = torch.tensor(t_in / t_out, dtype=torch.float32)  # ATen computes the scale in float
    return torch.clamp((torch.arange(t_out, dtype=torch.float32) * scale).floor().to(torch.int64), max=t_in - 1)


class HipWav2Vec2Bert:
    def __init__(self, config: tp.Any, state_dict: dict[str, torch.Tensor], device: str | torch.device = "cuda"):
        g = (lambda k: config[k]) if isinstance(config, dict) else (lambda k: getattr(config, k))
        self.dim, self.depth, self.heads, self.inter = g("hidden_size"), g("num_hidden_layers"), g("num_attention_heads"), g("intermediate_size")
        self.feat_dim, self.kernel = g("feature_projection_input_dim"), g("conv_depthwise_kernel_size")
        self.left, self.right = g("left_max_position_embeddings"), g("right_max_position_embeddings")
        self.eps = float(g("layer_norm_eps"))
        if g("position_embeddings_type") != "relative_key" or g("hidden_act") != "swish":
            raise NotImplementedError("only the w2v-bert-2.0 configuration (relative_key positions, swish) is built")
        self.dim_head = self.dim // self.heads
        self.feat_pad = ops.round_up(self.feat_dim, 64)
        self.device = dev = torch.device(device)
        sd = state_dict

        def f32(name: str) -> torch.Tensor:
            return sd[name].detach().to(device=dev, dtype=torch.float32).contiguous()

        self.keep: list[torch.Tensor] = []

        def own(t: torch.Tensor) -> int:
            self.keep.append(t)
            return t.data_ptr()

        self.fp_ln = (f32("feature_projection.layer_norm.weight"), f32("feature_projection.layer_norm.bias"))
        self.w_fp = ops.pack_weight(f32("feature_projection.projection.weight"), cols_pad=self.feat_pad)
        self.b_fp = f32("feature_projection.projection.bias")
        self.layers = (ConformerLayer * max(self.depth, 1))()
        H = self.dim
        for i in range(self.depth):
            p = f"encoder.layers.{i}."
            L = self.layers[i]
            for tag in ("ffn1", "ffn2"):
                setattr(L, f"{tag}_ln_w", own(f32(p + f"{tag}_layer_norm.weight")))
                setattr(L, f"{tag}_ln_b", own(f32(p + f"{tag}_layer_norm.bias")))
                setattr(L, f"w_{tag}_in", own(ops.pack_weight(f32(p + f"{tag}.intermediate_dense.weight"))))
                setattr(L, f"b_{tag}_in", own(f32(p + f"{tag}.intermediate_dense.bias")))
                setattr(L, f"w_{tag}_out", own(ops.pack_weight(f32(p + f"{tag}.output_dense.weight"))))
                setattr(L, f"b_{tag}_out_half", own(0.5 * f32(p + f"{tag}.output_dense.bias")))  # x + 0.5 * ffn(x)
            L.attn_ln_w, L.attn_ln_b = own(f32(p + "self_attn_layer_norm.weight")), own(f32(p + "self_attn_layer_norm.bias"))
            a = p + "self_attn."
            L.w_qkv = own(ops.pack_weight(torch.cat([f32(a + "linear_q.weight"), f32(a + "linear_k.weight"), f32(a + "linear_v.weight")])))
            L.b_qkv = own(torch.cat([f32(a + "linear_q.bias"), f32(a + "linear_k.bias"), f32(a + "linear_v.bias")]))
            L.dist_emb = own(ops.pack_weight(f32(a + "distance_embedding.weight")))  # [left+right+1, 64] bf16
            L.w_attn_out, L.b_attn_out = own(ops.pack_weight(f32(a + "linear_out.weight"))), own(f32(a + "linear_out.bias"))
            c = p + "conv_module."
            L.conv_ln_w, L.conv_ln_b = own(f32(c + "layer_norm.weight")), own(f32(c + "layer_norm.bias"))
            pw1 = f32(c + "pointwise_conv1.weight").squeeze(-1)  # [2H, H]: GLU halves a = rows [:H], b = rows [H:]
            L.w_pw1 = own(ops.pack_weight(torch.stack([pw1[:H], pw1[H:]], dim=1).reshape(2 * H, H).contiguous()))  # a0, b0, a1, b1, ...
            L.w_dw_kc = own(f32(c + "depthwise_conv.weight").squeeze(1).t().contiguous())  # [K, H] tap-major
            L.dw_ln_w, L.dw_ln_b = own(f32(c + "depthwise_layer_norm.weight")), own(f32(c + "depthwise_layer_norm.bias"))
            L.w_pw2 = own(ops.pack_weight(f32(c + "pointwise_conv2.weight").squeeze(-1)))
            L.final_ln_w, L.final_ln_b = own(f32(p + "final_layer_norm.weight")), own(f32(p + "final_layer_norm.bias"))

    def hidden_states_resampled(self, input_features: torch.Tensor, n_out: int) -> torch.Tensor:
        """input_features f32 [B, T, feat_dim] (unpadded chunks) -> f32 [B, depth + 1, dim, n_out]: every hidden state
        transposed to channels-first and nearest-resampled to n_out time points (audio.py:253-263, 163-171)."""
        feats = input_features.to(device=self.device, dtype=torch.float32).contiguous()
        B, T, Fd = feats.shape
        if Fd != self.feat_dim:
            raise ValueError(f"expected {self.feat_dim}-dim features, got {Fd}")
        idx = nearest_index(T, n_out).to(self.device)
        d = W2vBertDesc()
        d.B, d.T, d.feat_dim, d.feat_pad = B, T, self.feat_dim, self.feat_pad
        d.dim, d.depth, d.heads, d.dim_head, d.inter, d.conv_kernel = self.dim, self.depth, self.heads, self.dim_head, self.inter, self.kernel
        d.rel_left, d.rel_right, d.ln_eps = self.left, self.right, self.eps
        d.fp_ln_w, d.fp_ln_b = self.fp_ln[0].data_ptr(), self.fp_ln[1].data_ptr()
        d.w_fp, d.b_fp = self.w_fp.data_ptr(), self.b_fp.data_ptr()
        d.layers_host = C.cast(self.layers, C.POINTER(ConformerLayer))
        d.features, d.out_index, d.n_out = feats.data_ptr(), idx.data_ptr(), n_out
        states = torch.empty(self.depth + 1, B, n_out, self.dim, dtype=torch.float32, device=self.device)
        ws = ops.workspace(lib().tribe_w2vbert_workspace_bytes(C.byref(d)), self.device, "extractor")
        check(lib().tribe_w2vbert_fwd(C.byref(d), states.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
              "tribe_w2vbert_fwd")
        return states.permute(1, 0, 3, 2).contiguous()  # [B, n_states, dim, n_out]
