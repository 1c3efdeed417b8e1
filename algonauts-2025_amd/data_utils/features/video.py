"""V-JEPA2 (ViT-g) video feature extractor on MI355X HIP kernels.

Mirror of the reference plugin `VJEPA2` / `VideoModel` (/root/reference/data_utils/data_utils/features/video.py:56-274):
for every 0.5 s step a clip of 64 frames covering the previous 4 s (video.py:203-224) goes through the HF video
processor and `VJEPA2Model(..., output_hidden_states=True)` (video.py:247-268); the 41 hidden states are stacked and
averaged over the 8192 tokens (video.py:228) -> `[41, 1408]` per step.  Here the encoder forward and the token mean run
in one C call (`tribe_vjepa2_fwd`); the predictor head, whose output the reference discards, is not run.
Frame decoding and the HF processor (resize / rescale / normalise) stay on the host (SURVEY: moviepy decode is out of
scope); the boundary is the processor's `pixel_values_videos` f32 [B, frames, 3, H, W].
"""

from __future__ import annotations

import ctypes as C
import typing as tp

import numpy as np
import pydantic
import torch

from tribe_hip import ops
from tribe_hip._lib import Vjepa2Desc, VitLayer, check, lib

from .plugin import HbmFeaturePlugin

# facebook/vjepa2-vitg-fpc64-256 hyper-parameters (public model card; configuration input, not verifiable offline)
VJEPA2_VITG_FPC64_256 = dict(patch_size=16, crop_size=256, frames_per_clip=64, tubelet_size=2, hidden_size=1408, in_chans=3,
                             num_attention_heads=22, num_hidden_layers=40, mlp_ratio=48 / 11, layer_norm_eps=1e-6, qkv_bias=True)


def rope3d_tables(grid_depth: int, grid_size: int, dim_head: int) -> tuple[torch.Tensor, torch.Tensor]:
    """Per-element cos / sin tables [tokens, dim_head] of VJEPA2RopeAttention (modeling_vjepa2.py:180-294): the head is
    split into three segments of 2*((dim_head//3)//2) dims rotated by the frame / height / width index of the token;
    inside a segment of D dims element j uses frequency 10000^(-(j mod D/2)/(D/2)) (the library TILES its frequencies,
    `repeat(1,1,1,2)`, while pairing (2i, 2i+1)); remaining dims are passed through (cos 1, sin 0)."""
    seg = 2 * ((dim_head // 3) // 2)
    ids = torch.arange(grid_depth * grid_size * grid_size)
    tpf = grid_size * grid_size
    frame = ids // tpf
    height = (ids - tpf * frame) // grid_size
    width = (ids - tpf * frame) - grid_size * height
    omega = 1.0 / 10000 ** (torch.arange(seg // 2, dtype=torch.float32) / (seg / 2.0))
    cos = torch.ones(ids.numel(), dim_head)
    sin = torch.zeros(ids.numel(), dim_head)
    for k, pos in enumerate((frame, height, width)):
        freq = pos.float()[:, None] * omega[None, :]  # [tokens, seg/2]
        cos[:, k * seg:(k + 1) * seg] = freq.cos().repeat(1, 2)
        sin[:, k * seg:(k + 1) * seg] = freq.sin().repeat(1, 2)
    return cos.contiguous(), sin.contiguous()


class HipVJEPA2Encoder:
    def __init__(self, config: tp.Any, state_dict: dict[str, torch.Tensor], device: str | torch.device = "cuda"):
        g = (lambda k: config[k]) if isinstance(config, dict) else (lambda k: getattr(config, k))
        self.patch, self.crop, self.frames, self.tubelet = g("patch_size"), g("crop_size"), g("frames_per_clip"), g("tubelet_size")
        self.dim, self.chans, self.heads, self.depth = g("hidden_size"), g("in_chans"), g("num_attention_heads"), g("num_hidden_layers")
        self.mlp = int(self.dim * g("mlp_ratio"))
        self.eps = float(g("layer_norm_eps"))
        self.dim_head = self.dim // self.heads
        self.device = dev = torch.device(device)
        sd = {k.removeprefix("encoder."): v for k, v in state_dict.items() if not k.startswith("predictor.")}

        def f32(name: str) -> torch.Tensor | None:
            return sd[name].detach().to(device=dev, dtype=torch.float32).contiguous() if name in sd else None

        self.keep: list[torch.Tensor] = []

        def own(t: torch.Tensor | None) -> int | None:
            if t is None:
                return None
            self.keep.append(t)
            return t.data_ptr()

        wp = f32("embeddings.patch_embeddings.proj.weight")  # [dim, C, tub, p, p]
        self.w_patch = ops.pack_weight(wp.reshape(self.dim, -1).contiguous())
        self.b_patch = f32("embeddings.patch_embeddings.proj.bias")
        self.layers = (VitLayer * max(self.depth, 1))()
        self.packs: list[tuple[torch.Tensor, ...]] = []      # bf16 (qkv, proj, fc1, fc2) per layer, the source of the fp8 packs
        self.fp8_layers = None                               # VitFp8Layer array once enable_fp8() has run
        for i in range(self.depth):
            p = f"layer.{i}."
            L = self.layers[i]
            wqkv = torch.cat([f32(p + "attention.query.weight"), f32(p + "attention.key.weight"), f32(p + "attention.value.weight")])
            bq = f32(p + "attention.query.bias")
            L.norm1_w, L.norm1_b = own(f32(p + "norm1.weight")), own(f32(p + "norm1.bias"))
            packs = (ops.pack_weight(wqkv), ops.pack_weight(f32(p + "attention.proj.weight")), ops.pack_weight(f32(p + "mlp.fc1.weight")),
                     ops.pack_weight(f32(p + "mlp.fc2.weight")))
            self.packs.append(packs)
            L.w_qkv, L.w_proj, L.w_fc1, L.w_fc2 = (own(t) for t in packs)
            L.b_qkv = own(torch.cat([bq, f32(p + "attention.key.bias"), f32(p + "attention.value.bias")])) if bq is not None else None
            L.b_proj = own(f32(p + "attention.proj.bias"))
            L.norm2_w, L.norm2_b = own(f32(p + "norm2.weight")), own(f32(p + "norm2.bias"))
            L.b_fc1, L.b_fc2 = own(f32(p + "mlp.fc1.bias")), own(f32(p + "mlp.fc2.bias"))
        self._tabs: dict[tuple[int, int], tuple[torch.Tensor, torch.Tensor]] = {}

    def enable_fp8(self, calibration_clip: torch.Tensor, margin: float = 1.0) -> torch.Tensor:
        """e4m3 Linear GEMMs (BASELINE config 5), as HipLlamaModel.enable_fp8: per-tensor weight scales, static input scales
        from one bf16 pass over `calibration_clip` [B, frames, C, H, W].  Returns the amax table f32 [depth, 4]."""
        from tribe_hip._lib import VitFp8Layer

        if self.dim % 128 or self.mlp % 128:
            raise ValueError("fp8 path: hidden_size and the MLP width must be multiples of 128")
        self.fp8_layers = None
        amax = torch.zeros(max(self.depth, 1), 4, dtype=torch.float32, device=self.device)
        self.hidden_state_means(calibration_clip, _amax=amax)
        table = amax.cpu()
        if not bool((table[: self.depth] > 0).all()):
            raise ValueError("fp8 calibration saw an all-zero GEMM input")
        layers = (VitFp8Layer * max(self.depth, 1))()
        self.fp8_packs = []
        for i in range(self.depth):
            q = []
            for j, w in enumerate(self.packs[i]):
                w_scale = float(ops.absmax(w)) / ops.FP8_MAX
                q.append(ops.quantize_fp8(w, w_scale, K_pad=w.shape[1]))
                layers[i].w_scale[j] = w_scale
                layers[i].in_scale[j] = float(table[i, j]) * margin / ops.FP8_MAX
            layers[i].w_qkv, layers[i].w_proj, layers[i].w_fc1, layers[i].w_fc2 = (t.data_ptr() for t in q)
            self.fp8_packs.append(q)
        self.fp8_layers = layers
        return table

    def hidden_state_means(self, pixel_values_videos: torch.Tensor, fp8: bool | None = None, _amax: torch.Tensor | None = None) -> torch.Tensor:
        """pixel_values_videos f32 [B, frames, C, H, W] -> f32 [B, depth + 1, dim] (video.py:262-268 + :228).
        fp8: None = use the e4m3 GEMMs when enable_fp8() has run."""
        pix = pixel_values_videos.to(device=self.device, dtype=torch.float32).contiguous()
        B, F, Cc, H, W = pix.shape
        if Cc != self.chans or F % self.tubelet or H % self.patch or W % self.patch:
            raise ValueError(f"unexpected clip shape {tuple(pix.shape)}")
        key = (F // self.tubelet, H // self.patch)
        if H != W:
            raise ValueError("square frames expected (crop_size x crop_size)")
        if key not in self._tabs:
            cos, sin = rope3d_tables(key[0], key[1], self.dim_head)
            self._tabs[key] = (cos.to(self.device), sin.to(self.device))
        cos, sin = self._tabs[key]
        d = Vjepa2Desc()
        d.B, d.frames, d.chans, d.height, d.width, d.tubelet, d.patch = B, F, Cc, H, W, self.tubelet, self.patch
        d.dim, d.depth, d.heads, d.dim_head, d.mlp, d.ln_eps = self.dim, self.depth, self.heads, self.dim_head, self.mlp, self.eps
        d.w_patch, d.b_patch, d.K_pad = self.w_patch.data_ptr(), ops._p(self.b_patch), self.w_patch.shape[1]
        d.layers_host = C.cast(self.layers, C.POINTER(VitLayer))
        d.cos_tab, d.sin_tab, d.pixels = cos.data_ptr(), sin.data_ptr(), pix.data_ptr()
        use_fp8 = (self.fp8_layers is not None) if fp8 is None else fp8
        if use_fp8 and _amax is None:
            if self.fp8_layers is None:
                raise ValueError("hidden_state_means(fp8=True) before enable_fp8()")
            d.fp8_host = C.cast(self.fp8_layers, C.POINTER(type(self.fp8_layers[0])))
        if _amax is not None:
            d.amax_out = _amax.data_ptr()
        states = torch.empty(self.depth + 1, B, self.dim, dtype=torch.float32, device=self.device)
        ws = ops.workspace(lib().tribe_vjepa2_workspace_bytes(C.byref(d)), self.device, "extractor")
        check(lib().tribe_vjepa2_fwd(C.byref(d), states.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
              "tribe_vjepa2_fwd")
        return states.transpose(0, 1).contiguous()


def default_video_processor(frames: np.ndarray, crop_size: int = 256) -> torch.Tensor:
    """What the HF `VJEPA2VideoProcessor` does at its defaults (video_processing_vjepa2.py: resize shortest edge to
    int(crop * 256 / 224) bilinear, centre crop, rescale 1/255, ImageNet normalise) for uint8 frames [F, H, W, 3] ->
    f32 [1, F, 3, crop, crop].  The HF class needs torchvision (absent); its interpolation kernel's antialiasing is not
    reproduced bit for bit, so pixel parity is unpinned -- pass `processor=` to `VJEPA2.attach` to use another front end."""
    x = torch.from_numpy(np.ascontiguousarray(frames)).permute(0, 3, 1, 2).float()
    H, W = x.shape[-2:]
    short = int(crop_size * 256 / 224)
    scale = short / min(H, W)
    nh, nw = max(short, int(round(H * scale))), max(short, int(round(W * scale)))
    x = torch.nn.functional.interpolate(x, size=(nh, nw), mode="bilinear", align_corners=False, antialias=True)
    top, left = (nh - crop_size) // 2, (nw - crop_size) // 2
    x = x[:, :, top:top + crop_size, left:left + crop_size] / 255.0
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    return ((x - mean) / std)[None]


class VJEPA2(HbmFeaturePlugin):
    """The reference's video feature (video.py:56-236) on the HIP ViT-g forward: fields `name`, `layers`, `layer_aggregation`,
    `device`, `infra`; `prepare`, `__call__ -> Tensor[L, D, T]`, `_get_data -> [41, 1408, T_event@2Hz]` per Video event (item
    uid `filepath_offset_duration`, video.py:191-195).  Per 0.5 s step the 64 frames of the previous 4 s are decoded on the
    host (`event.read().get_frame(t)`, moviepy in the reference), processed, and the encoder forward + token mean of all
    41 states is one C call; `clips_per_launch` steps go through the encoder together."""

    name: tp.Literal["VJEPA2"] = "VJEPA2"
    pretrained: str = "facebook/vjepa2-vitg-fpc64-256"    # video.py:247-254; resolved from the local HF cache only
    clips_per_launch: int = 4   # 37.6 ms per clip at 4, 41.3 at 2, 44.9 at 1 (profiles/r02_p_config3_e2e.txt, r02_m_extractor_bench.txt)
    _EVENT_TYPE: tp.ClassVar[str] = "Video"
    _KIND: tp.ClassVar[str] = "sampled"
    _PASS_EVENT_DURATION: tp.ClassVar[bool] = True
    _model: tp.Any = pydantic.PrivateAttr(default=None)
    _processor: tp.Any = pydantic.PrivateAttr(default=None)

    def attach(self, model: HipVJEPA2Encoder, processor: tp.Callable[[np.ndarray], torch.Tensor] | None = None) -> "VJEPA2":
        self._model, self._processor = model, processor
        return self

    @property
    def model(self) -> HipVJEPA2Encoder:
        if self._model is None:
            from transformers import AutoModel

            hf = AutoModel.from_pretrained(self.pretrained, local_files_only=True)
            self._model = HipVJEPA2Encoder(hf.config, hf.state_dict())
        return self._model

    def _item_uid(self, event: tp.Any) -> str:
        return f"{event.filepath}_{event.offset:.2f}_{event.duration:.2f}"

    def _compute(self, events: list[tp.Any]) -> tp.Iterator[np.ndarray]:
        from ..base import Frequency

        model = self.model
        process = self._processor or (lambda fr: default_video_processor(fr, model.crop))
        n_frames = model.frames
        subtimes = [k / n_frames * 4.0 for k in reversed(range(n_frames))]                  # video.py:203-205
        for event in events:
            video = event.read()
            expect = Frequency(2.0).to_ind(event.duration)
            times = np.linspace(0, video.duration, expect + 1)[1:]                          # video.py:218
            out = np.zeros((len(times), model.depth + 1, model.dim))                        # f64, as np.zeros in video.py:230
            for k0 in range(0, len(times), self.clips_per_launch):
                clips = []
                for t in times[k0:k0 + self.clips_per_launch]:
                    frames = np.array([np.asarray(video.get_frame(max(0, t - t2))).astype("uint8") for t2 in subtimes])
                    clips.append(process(frames))
                means = model.hidden_state_means(torch.cat(clips, dim=0))                   # [clips, 41, dim]
                out[k0:k0 + len(clips)] = means.cpu().numpy()
            if hasattr(video, "close"):
                video.close()
            yield out.transpose(1, 2, 0)                                                    # [n_states, dim, T_event] (video.py:234)
