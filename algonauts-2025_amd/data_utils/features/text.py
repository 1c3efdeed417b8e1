"""Llama-3.2 text feature extractor on MI355X HIP kernels.

Mirror of the reference plugin `LLAMA3p2` (/root/reference/data_utils/data_utils/features/text.py:42-256): for
every word, its left context is tokenised (right padding, left truncation, text.py:166-168,226-232), run through
`LlamaModel(..., output_hidden_states=True)` (text.py:236-240) and every hidden state is averaged over the last
`len(word)` non-pad positions (text.py:245-254) -> `[n_layers + 1, hidden]` per word; layer groups are then formed by
`_aggregate_layers` (text.py:129-149).

Here the model forward AND the pooling run in one C call (`tribe_llama_fwd`: RMSNorm, fused q|k|v MFMA GEMM, rotary,
causal grouped-query flash attention, SwiGLU GEMM epilogue, f32 residual stream); only `[B, n_states, hidden]` floats
leave the GPU, where the reference copies every hidden state to the host (text.py:240).

Weights come from any `transformers` LlamaModel / state_dict (names `embed_tokens.weight`,
`layers.N.self_attn.{q,k,v,o}_proj.weight`, `layers.N.mlp.{gate,up,down}_proj.weight`, `layers.N.*layernorm.weight`,
`norm.weight`).  The reference fetches `meta-llama/Llama-3.2-3B` by name; that checkpoint is not available offline, so
parity is tested against the installed `transformers` implementation with random weights (tests/test_gpu_extractors.py).
"""

from __future__ import annotations

import ctypes as C
import math
import typing as tp

import numpy as np
import pydantic
import torch

from tribe_hip import _lib, ops
from tribe_hip._lib import LlamaDesc, LlamaLayer, check, lib

from .plugin import HbmFeaturePlugin

# Llama-3.2-3B hyper-parameters (public model card; not verifiable offline -> configuration input)
LLAMA_3P2_3B = dict(
    vocab_size=128256, hidden_size=3072, intermediate_size=8192, num_hidden_layers=28, num_attention_heads=24,
    num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-5, max_position_embeddings=131072,
    rope_parameters={"rope_type": "llama3", "rope_theta": 500000.0, "factor": 32.0, "low_freq_factor": 1.0,
                     "high_freq_factor": 4.0, "original_max_position_embeddings": 8192},
)


def rope_inv_freq(head_dim: int, rope: dict[str, tp.Any]) -> torch.Tensor:
    """Inverse frequencies of HF's LlamaRotaryEmbedding: 'default' and 'llama3' (modeling_rope_utils llama3 rule:
    wavelengths above original_ctx / low_freq_factor are divided by `factor`, the band down to
    original_ctx / high_freq_factor is interpolated smoothly)."""
    base = float(rope.get("rope_theta", 10000.0))
    inv = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    kind = rope.get("rope_type", "default")
    if kind == "default":
        return inv
    if kind != "llama3":
        raise NotImplementedError(f"rope_type {kind!r} is not used by Llama-3.2")
    factor, lo, hi = float(rope["factor"]), float(rope["low_freq_factor"]), float(rope["high_freq_factor"])
    old = float(rope["original_max_position_embeddings"])
    wavelen = 2 * math.pi / inv
    scaled = torch.where(wavelen > old / lo, inv / factor, inv)
    smooth = (old / wavelen - lo) / (hi - lo)
    smoothed = (1 - smooth) * scaled / factor + smooth * scaled
    medium = ~(wavelen < old / hi) & ~(wavelen > old / lo)
    return torch.where(medium, smoothed, scaled)


class HipLlamaModel:
    """Packed bf16 weights of a LlamaModel + the forward-with-pooling launcher."""

    def __init__(self, config: tp.Any, state_dict: dict[str, torch.Tensor], device: str | torch.device = "cuda"):
        g = (lambda k: config[k]) if isinstance(config, dict) else (lambda k: getattr(config, k))
        self.dim, self.depth, self.inter = g("hidden_size"), g("num_hidden_layers"), g("intermediate_size")
        self.heads_q, self.heads_kv = g("num_attention_heads"), g("num_key_value_heads")
        try:
            self.dim_head = g("head_dim") or self.dim // self.heads_q
        except (KeyError, AttributeError):
            self.dim_head = self.dim // self.heads_q
        self.eps, self.vocab = float(g("rms_norm_eps")), g("vocab_size")
        try:
            rope = g("rope_parameters")
        except (KeyError, AttributeError):
            rope = None
        self.inv_freq = rope_inv_freq(self.dim_head, dict(rope) if rope else {"rope_type": "default", "rope_theta": 10000.0})
        self.device = torch.device(device)
        sd = {k.removeprefix("model."): v for k, v in state_dict.items()}
        dev = self.device

        def f32(name: str) -> torch.Tensor:
            return sd[name].detach().to(device=dev, dtype=torch.float32).contiguous()

        self.keep: list[torch.Tensor] = []

        def own(t: torch.Tensor) -> int:
            self.keep.append(t)
            return t.data_ptr()

        self.embed = f32("embed_tokens.weight").to(torch.bfloat16).contiguous()  # bf16 table: 0.79 GB for the 3B vocab
        self.layers = (LlamaLayer * max(self.depth, 1))()
        self.packs: list[tuple[torch.Tensor, ...]] = []      # bf16 (qkv, o, gate_up, down) per layer, the source of the fp8 packs
        self.fp8_layers = None                               # LlamaFp8Layer array once enable_fp8() has run
        for i in range(self.depth):
            p = f"layers.{i}."
            L = self.layers[i]
            wqkv = torch.cat([f32(p + "self_attn.q_proj.weight"), f32(p + "self_attn.k_proj.weight"), f32(p + "self_attn.v_proj.weight")])
            gate, up = f32(p + "mlp.gate_proj.weight"), f32(p + "mlp.up_proj.weight")
            gate_up = torch.stack([gate, up], dim=1).reshape(2 * self.inter, self.dim).contiguous()  # rows: g0, u0, g1, u1, ...
            packs = (ops.pack_weight(wqkv), ops.pack_weight(f32(p + "self_attn.o_proj.weight")), ops.pack_weight(gate_up),
                     ops.pack_weight(f32(p + "mlp.down_proj.weight")))
            self.packs.append(packs)
            L.input_norm_w = own(f32(p + "input_layernorm.weight"))
            L.w_qkv, L.w_o, L.w_gate_up, L.w_down = (t.data_ptr() for t in packs)
            L.post_norm_w = own(f32(p + "post_attention_layernorm.weight"))
            del wqkv, gate, up, gate_up
        self.final_norm = f32("norm.weight")
        self._tabs: dict[int, tuple[torch.Tensor, torch.Tensor]] = {}

    def _tables(self, T: int) -> tuple[torch.Tensor, torch.Tensor]:
        if T not in self._tabs:
            freqs = torch.outer(torch.arange(T, dtype=torch.float32), self.inv_freq)  # attention_scaling == 1 for llama3
            self._tabs[T] = (freqs.cos().to(self.device).contiguous(), freqs.sin().to(self.device).contiguous())
        return self._tabs[T]

    def enable_fp8(self, calibration_ids: torch.Tensor, margin: float = 1.0) -> torch.Tensor:
        """Switch the four Linear GEMMs of every layer to e4m3 (BASELINE config 5): per-tensor weight scales amax / 448 and
        static per-tensor input scales from one bf16 calibration pass over `calibration_ids` [B, T] (amax * margin / 448).
        Returns the calibration amax table f32 [depth, 4] (qkv in, o_proj in, gate_up in, down in)."""
        if any(v % 128 for v in (self.dim, self.heads_q * self.dim_head, self.inter)):
            raise ValueError("fp8 path: hidden_size, num_attention_heads * head_dim and intermediate_size must be multiples of 128")
        self.fp8_layers = None
        amax = torch.zeros(max(self.depth, 1), 4, dtype=torch.float32, device=self.device)
        B, T = calibration_ids.shape
        zeros, full = torch.zeros(B, dtype=torch.int64), torch.full((B,), T, dtype=torch.int64)
        self.forward_pooled(calibration_ids, zeros, full, _amax=amax)
        table = amax.cpu()                                    # one-time sync: the scales become launch constants
        if not bool((table[: self.depth] > 0).all()):
            raise ValueError("fp8 calibration saw an all-zero GEMM input")
        layers = (_lib.LlamaFp8Layer * max(self.depth, 1))()
        self.fp8_packs = []
        for i in range(self.depth):
            q = []
            for j, w in enumerate(self.packs[i]):
                w_scale = float(ops.absmax(w)) / ops.FP8_MAX
                q.append(ops.quantize_fp8(w, w_scale, K_pad=w.shape[1]))
                layers[i].w_scale[j] = w_scale
                layers[i].in_scale[j] = float(table[i, j]) * margin / ops.FP8_MAX
            layers[i].w_qkv, layers[i].w_o, layers[i].w_gate_up, layers[i].w_down = (t.data_ptr() for t in q)
            self.fp8_packs.append(q)
        self.fp8_layers = layers
        return table

    def forward_pooled(self, input_ids: torch.Tensor, pool_start: torch.Tensor, pool_len: torch.Tensor, fp8: bool | None = None,
                       _amax: torch.Tensor | None = None) -> torch.Tensor:
        """input_ids int64 [B, T] (right padded); returns f32 [n_layers + 1, B, dim]: every hidden state averaged over
        positions [pool_start[b], pool_start[b] + pool_len[b]).  fp8: None = use the e4m3 GEMMs when enable_fp8() has run."""
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        B, T = ids.shape
        if int(ids.min()) < 0 or int(ids.max()) >= self.vocab:
            raise ValueError("token id outside the vocabulary")
        cos, sin = self._tables(T)
        start = pool_start.to(device=self.device, dtype=torch.int64).contiguous()
        length = pool_len.to(device=self.device, dtype=torch.int64).contiguous()
        d = LlamaDesc()
        d.B, d.T = B, T
        d.dim, d.depth, d.heads_q, d.heads_kv, d.dim_head, d.inter = self.dim, self.depth, self.heads_q, self.heads_kv, self.dim_head, self.inter
        d.rms_eps = self.eps
        d.embed, d.embed_dtype, d.vocab = self.embed.data_ptr(), _lib.BF16, self.vocab
        d.layers_host = C.cast(self.layers, C.POINTER(LlamaLayer))
        d.final_norm_w = self.final_norm.data_ptr()
        d.cos_tab, d.sin_tab = cos.data_ptr(), sin.data_ptr()
        d.ids, d.pool_start, d.pool_len = ids.data_ptr(), start.data_ptr(), length.data_ptr()
        use_fp8 = (self.fp8_layers is not None) if fp8 is None else fp8
        if use_fp8 and _amax is None:
            if self.fp8_layers is None:
                raise ValueError("forward_pooled(fp8=True) before enable_fp8()")
            d.fp8_host = C.cast(self.fp8_layers, C.POINTER(_lib.LlamaFp8Layer))
        if _amax is not None:
            d.amax_out = _amax.data_ptr()
        states = torch.empty(self.depth + 1, B, self.dim, dtype=torch.float32, device=self.device)
        ws = ops.workspace(lib().tribe_llama_workspace_bytes(C.byref(d)), self.device, "extractor")
        check(lib().tribe_llama_fwd(C.byref(d), states.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
              "tribe_llama_fwd")
        return states


def word_pool_windows(input_ids: torch.Tensor, target_words: tp.Sequence[str], pad_id: int) -> tuple[torch.Tensor, torch.Tensor]:
    """text.py:245-252: n_pads = #tokens equal to pad_id; keep [:-n_pads]; average the last len(word) positions
    (python slicing semantics: a window longer than the sequence -- or len(word) == 0 -- takes everything)."""
    ids = input_ids.cpu()
    T = ids.shape[1]
    n_real = T - (ids == pad_id).sum(dim=1)
    k = torch.tensor([len(w) for w in target_words], dtype=torch.int64)
    k = torch.where((k == 0) | (k > n_real), n_real, k)
    return (n_real - k).to(torch.int64), k.to(torch.int64)


class LLAMA3p2(HbmFeaturePlugin):
    """The reference's text feature (text.py:42-256) on the HIP Llama forward: fields `name`, `layers`, `layer_aggregation`,
    `device`, `infra`; `prepare(events)`, `__call__(events, start, duration, trigger) -> Tensor[L, D, T]` and
    `_get_data(events) -> Iterator[np.ndarray [n_states, hidden]]` (data_utils/features/plugin.py).  One latent per Word
    event (item uid `text_context`, text.py:199-203), held for the word's duration."""

    name: tp.Literal["LLAMA3p2"] = "LLAMA3p2"
    batch_size: int = 8                                   # text.py:212 (DataLoader batch of contexts)
    pretrained: str = "meta-llama/Llama-3.2-3B"           # text.py:166-173; resolved from the local HF cache only
    _EVENT_TYPE: tp.ClassVar[str] = "Word"
    _KIND: tp.ClassVar[str] = "words"
    _model: tp.Any = pydantic.PrivateAttr(default=None)
    _tokenizer: tp.Any = pydantic.PrivateAttr(default=None)

    def attach(self, model: HipLlamaModel, tokenizer: tp.Any) -> "LLAMA3p2":
        """Provide weights + tokenizer explicitly (offline use: the reference fetches them by name)."""
        self._model, self._tokenizer = model, tokenizer
        return self

    @property
    def model(self) -> HipLlamaModel:
        if self._model is None:
            from transformers import AutoModel, AutoTokenizer  # local cache only: there is no network on the GPU boxes

            tok = AutoTokenizer.from_pretrained(self.pretrained, truncation_side="left", local_files_only=True)
            hf = AutoModel.from_pretrained(self.pretrained, local_files_only=True)
            if tok.pad_token is None:
                tok.pad_token = tok.eos_token
            self._model, self._tokenizer = HipLlamaModel(hf.config, hf.state_dict()), tok
        return self._model

    @property
    def tokenizer(self) -> tp.Any:
        self.model
        return self._tokenizer

    def _item_uid(self, event: tp.Any) -> str:
        return f"{event.text}_{event.context}"

    def _compute(self, events: list[tp.Any]) -> tp.Iterator[np.ndarray]:
        return self.extract([e.text for e in events], [e.context for e in events])

    def extract(self, target_words: tp.Sequence[str], contexts: tp.Sequence[str]) -> tp.Iterator[np.ndarray]:
        """The body of the reference's `_get_data` loop (text.py:204-256): yields [n_states, hidden] per word."""
        model, tok = self.model, self.tokenizer
        pad_id = tok.eos_token_id
        for i in range(0, len(contexts), self.batch_size):
            words, ctx = list(target_words[i:i + self.batch_size]), list(contexts[i:i + self.batch_size])
            enc = tok(ctx, add_special_tokens=False, return_tensors="pt", padding=True, truncation=True)
            start, length = word_pool_windows(enc["input_ids"], words, pad_id)
            states = model.forward_pooled(enc["input_ids"], start, length).cpu().numpy()  # [n_states, B, dim]
            for j in range(len(words)):
                yield states[:, j]

    def aggregate(self, latents: np.ndarray) -> np.ndarray:
        return self._aggregate_layers(latents)
