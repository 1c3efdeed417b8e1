"""BASELINE config 3 on one MI355X: trimodal encode with ON-THE-FLY extraction of one 149 s window (100 TRs, 298 feature
steps at 2 Hz) of one subject, full-size architectures with random weights (checkpoints are not fetchable):
  video  298 clips of 64 x 256 x 256 frames -> V-JEPA2 ViT-g  (one clip per 0.5 s, video.py:191-236)
  audio  3 chunks of <= 60 s (3000 + 3000 + 1450 fbank frames) -> Wav2Vec-BERT 2.0 (audio.py:253-263)
  text   ~370 words, each with a 1024-token context -> Llama-3.2-3B (text.py:204-256, batches of 8)
  -> HBM feature store -> segment loader -> FmriEncoder (3072 x 8 layers) -> [1, 1000, 100].
GPU box:  python scripts/e2e_bench.py [fp8] [clips=N]      ("fp8": e4m3 Linear GEMMs in Llama and V-JEPA2; N clips per V-JEPA2 launch, default 4)"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd"), str(ROOT / "scripts")]
import torch  # noqa: E402

import extractor_bench as xb  # noqa: E402  (random full-size state dicts)
from algonauts2025.model import FmriEncoderConfig  # noqa: E402
from data_utils.events import Fmri, Sound, Video, Word  # noqa: E402
from data_utils.gpu_loader import FeatureSpec, GpuSegmentLoader, HbmFeatureStore  # noqa: E402
from data_utils.segments import Segment  # noqa: E402

fp8 = "fp8" in sys.argv[1:]
CLIPS = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("clips=")), 4)   # V-JEPA2 clips per launch
dev = "cuda"
g = torch.Generator().manual_seed(0)


def clock(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


llama, vocab = xb.build_llama()
vj = xb.build_vjepa2()
w2v = xb.build_w2vbert()
if fp8:
    llama.enable_fp8(torch.randint(0, vocab, (4, 1024), generator=g))
    vj.enable_fp8(torch.randn(1, 64, 3, 256, 256, device=dev))
T_steps, n_words, ctx = 298, 368, 1024


def run_video():
    out = []
    for i in range(0, T_steps, CLIPS):
        clips = torch.randn(min(CLIPS, T_steps - i), 64, 3, 256, 256, device=dev)
        out.append(vj.hidden_state_means(clips))                    # [b, 41, 1408]
    return torch.cat(out).permute(1, 2, 0).contiguous()             # [41, 1408, 298]


def run_audio():
    parts = []
    for frames, steps in ((3000, 120), (3000, 120), (1450, 58)):
        parts.append(w2v.hidden_states_resampled(torch.randn(1, frames, 160, device=dev), steps)[0])
    return torch.cat(parts, dim=-1).contiguous()                    # [25, 1024, 298]


def run_text():
    out = []
    for i in range(0, n_words, 8):
        b = min(8, n_words - i)
        ids = torch.randint(0, vocab, (b, ctx), generator=g)
        out.append(llama.forward_pooled(ids, torch.full((b,), ctx - 3), torch.full((b,), 3)))   # [29, b, 3072]
    return torch.cat(out, dim=1).permute(1, 0, 2).contiguous()      # [n_words, 29, 3072]


for name, fn in (("warm-up", lambda: (vj.hidden_state_means(torch.randn(1, 64, 3, 256, 256, device=dev)), run_audio(),
                                        llama.forward_pooled(torch.randint(0, vocab, (8, ctx), generator=g), torch.full((8,), 0), torch.full((8,), 3)))),):
    clock(fn)
video, t_v = clock(run_video)
audio, t_a = clock(run_audio)
text, t_t = clock(run_text)

store = HbmFeatureStore(FeatureSpec.defaults())
snd, vid = Sound(start=0.0, duration=149.0, filepath="a.wav", timeline="t"), Video(start=0.0, duration=149.0, filepath="v.mkv", timeline="t")
rec = Fmri(start=4.47, duration=100 * 1.49, filepath="f.h5", frequency=1 / 1.49, subject="sub-01", timeline="t")
words = [Word(start=0.4 * i, duration=0.3, text="w", timeline="t") for i in range(n_words)]


def assemble_and_encode():
    store.put("video", vid, video)
    store.put("audio", snd, audio)
    store.put_words("text", words, text)
    store.put("fmri", rec, torch.randn(1000, 100))
    batch = GpuSegmentLoader(store, subject_index={"sub-01": 0}).batch([Segment(start=0.0, duration=149.0, ns_events=[rec, snd, vid] + words)])
    return model(batch)


fdims = {"text": (2, 3072), "audio": (2, 1024), "video": (2, 1408)}
model = FmriEncoderConfig(n_subjects=4).build(fdims, 1000, 100).to(dev).eval()
clock(assemble_and_encode)
y, t_e = clock(assemble_and_encode)
total = t_v + t_a + t_t + t_e
print(f"config 3, one 149 s window ({'fp8' if fp8 else 'bf16'} extractor GEMMs): video {t_v:.2f} s ({T_steps / t_v:.1f} clips/s), audio {t_a * 1e3:.0f} ms, "
      f"text {t_t:.2f} s ({n_words / t_t:.0f} words/s), store + load + encode {t_e * 1e3:.1f} ms -> {tuple(y.shape)}; "
      f"{100 / total:.2f} TRs/s end to end ({total:.2f} s per window; the reference caches extractor outputs once per stimulus)", flush=True)
