"""Segment-batch assembly from the HBM feature store vs the reference-style host path, at the reference's extractor
widths (text 2x3072, audio 2x1024, video 2x1408) and the synthetic sequence length (T = 1024 steps at 2 Hz).

GPU box:  python scripts/loader_bench.py [B]

Prints per-modality launch time (HIP events on the launch stream), algorithmic bytes (f32 source slice read + bf16
packed rows written) and the fraction of the 8 TB/s HBM peak, the wall time of `loader.batch()` with cold and cached
plans, and the host path the reference takes for ONE segment (numpy assembly with this build's host mirror of
TimedArray, data_utils/base.py, + H2D copy + tribe_pack_features), scaled to the batch."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "algonauts-2025_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from data_utils.events import Fmri, Segment, Sound, Video, Word  # noqa: E402
from data_utils.gpu_loader import FeatureSpec, GpuSegmentLoader, HbmFeatureStore  # noqa: E402
from data_utils.base import TimedArray  # noqa: E402
from data_utils.features.layers import aggregate_layers  # noqa: E402
from tribe_hip import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T, HBM_PEAK = 1024, 8.0e12
rng = np.random.default_rng(0)
T_ev = 2400                                   # a 20-minute movie chunk at 2 Hz
specs = FeatureSpec.defaults()
store = HbmFeatureStore(specs)
snd, vid = Sound(start=0.0, duration=T_ev / 2, filepath="a.wav"), Video(start=0.0, duration=T_ev / 2, filepath="v.mkv")
rec = Fmri(start=0.0, duration=805 * 1.49, filepath="f.h5", frequency=1 / 1.49, subject="sub-01")
audio = rng.standard_normal((25, 1024, T_ev), dtype=np.float32)
video = rng.standard_normal((41, 1408, T_ev), dtype=np.float32)
n_words = 3600
w_start = np.sort(rng.uniform(0, T_ev / 2, n_words)).round(3)
w_dur = rng.uniform(0.05, 0.6, n_words).round(3)
w_lat = rng.standard_normal((n_words, 29, 3072), dtype=np.float32)
words = [Word(start=float(s), duration=float(d), text=f"w{i}", timeline="t") for i, (s, d) in enumerate(zip(w_start, w_dur))]
t0 = time.perf_counter()
store.put("audio", snd, audio)
store.put("video", vid, video)
store.put("fmri", rec, rng.standard_normal((1000, 805), dtype=np.float32))
store.put_words("text", words, w_lat)
torch.cuda.synchronize()
print(f"store filled in {time.perf_counter() - t0:.2f} s: {store.nbytes() / 2**30:.2f} GiB resident "
      f"(raw states uploaded: {(audio.nbytes + video.nbytes + w_lat.nbytes) / 2**30:.2f} GiB)", flush=True)

dur = T / 2.0
starts = rng.uniform(0, T_ev / 2 - dur, B).round(2)
segs = [Segment(start=float(s), duration=dur, ns_events=[rec, snd, vid] + [w for w in words if w.start < s + dur and w.stop > s]) for s in starts]
loader = GpuSegmentLoader(store, subject_index={"sub-01": 0})
t0 = time.perf_counter()
batch = loader.batch(segs)
torch.cuda.synchronize()
cold = time.perf_counter() - t0
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    batch = loader.batch(segs)
torch.cuda.synchronize()
warm = (time.perf_counter() - t0) / reps
print(f"loader.batch(B={B}, T={T}): cold plans {cold * 1e3:.2f} ms, cached plans {warm * 1e3:.3f} ms wall per batch", flush=True)

by = {s.name: s for s in specs}
for name in ("text", "audio", "video", "fmri"):
    spec = by[name]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    loader.feature(spec, segs)
    torch.cuda.synchronize()
    # time only the kernel: rebuild the device tables once, then launch repeatedly on the current stream
    L, D = store.channels[name]
    C = L * D
    n_out = T if name != "fmri" else loader.plan(segs[0], spec).n_out
    if spec.kind == "words":
        ptr = np.zeros(B * T + 1, dtype=np.int64)
        idx = []
        for b, s in enumerate(segs):
            p = loader.plan(s, spec)
            order = np.argsort(p.steps, kind="stable")
            np.add.at(ptr, b * T + p.steps + 1, 1)
            idx.append(p.rows[order])
        np.cumsum(ptr, out=ptr)
        ptr_t = torch.from_numpy(ptr.astype(np.int32)).cuda()
        idx_t = torch.from_numpy(np.concatenate(idx).astype(np.int32)).cuda()
        table = store.word_table(name)
        run = lambda: ops.word_bag(table, ptr_t, idx_t, B * T)  # noqa: E731
        nbytes = idx_t.numel() * C * 4 + B * T * ((C + 63) // 64 * 64) * 2
    else:
        pieces = np.concatenate([loader.plan(s, spec).pieces for s in segs])
        seg_ptr = np.concatenate([[0], np.cumsum([len(loader.plan(s, spec).pieces) for s in segs])]).astype(np.int32)
        pieces_t = torch.from_numpy(pieces.view(np.uint8).reshape(-1)).cuda()
        seg_t = torch.from_numpy(seg_ptr).cuda()
        packed = spec.kind != "target"
        run = lambda: ops.segment_gather(pieces_t, seg_t, B, C, n_out, packed=packed)  # noqa: E731
        nbytes = B * C * n_out * 4 + (B * n_out * ((C + 63) // 64 * 64) * 2 if packed else B * C * n_out * 4)
    for _ in range(3):
        run()
    ev0.record()
    for _ in range(reps):
        run()
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    print(f"  {name:5s} C={C:5d}: {ms * 1e3:8.1f} us/launch  {nbytes / 2**20:8.1f} MiB algorithmic  {nbytes / (ms * 1e-3) / 1e9:8.0f} GB/s "
          f"= {nbytes / (ms * 1e-3) / HBM_PEAK * 100:5.1f} % of HBM peak", flush=True)

# ---- the reference's route for one segment: numpy on the host, H2D copy of fp32 [L, D, T], transpose/cast on the device
seg = segs[0]


def host_feature(pieces):
    out = TimedArray(aggregation="sum", start=seg.start, frequency=2.0, duration=seg.duration)
    for ta in pieces:
        out += ta
    return out.data


def sampled(ev, states, spec, **kw):
    sub = TimedArray(data=states, start=ev.start, frequency=2.0, **kw).overlap(seg.start, seg.duration)
    sub.data = aggregate_layers(sub.data, spec.layers, spec.layer_aggregation)
    return [sub]


row_of = {id(w): i for i, w in enumerate(words)}
t0 = time.perf_counter()
a = host_feature(sampled(snd, audio, by["audio"]))
v = host_feature(sampled(vid, video, by["video"], duration=vid.duration))
ws = [w for w in seg.ns_events if w.type == "Word"]
tx = host_feature(TimedArray(frequency=0, duration=w.duration, start=w.start, data=aggregate_layers(w_lat[row_of[id(w)]], by["text"].layers,
                                                                                                   by["text"].layer_aggregation))
                  for w in ws)
host = time.perf_counter() - t0
t0 = time.perf_counter()
for arr in (a, v, tx):
    ops.pack_features(torch.from_numpy(arr)[None].cuda(), layer_mean=False)
torch.cuda.synchronize()
h2d = time.perf_counter() - t0
print(f"host route, ONE segment on one core: numpy assembly {host * 1e3:.1f} ms + H2D/pack {h2d * 1e3:.1f} ms "
      f"-> {B} segments = {(host + h2d) * B * 1e3:.0f} ms per batch (vs {warm * 1e3:.3f} ms)", flush=True)
