"""Golden vectors for the time-axis arithmetic of the feature pipeline.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_timeline.py

EXECUTES the reference's own `data_utils/data_utils/base.py` (loaded by path; it needs numpy / pydantic /
yaml only) and the reference's `_aggregate_layers` methods (compiled out of text.py / audio.py / video.py
with `ast`, as make_golden.py does), then replays the assembly loop of the feature `__call__`s
(text.py:85-124, audio.py:78-120, video.py:172-189, neuro.py:60-106: `out = TimedArray(...)`;
`for ta in tarrays: out += ta`) on small synthetic timelines.  base.py creates `~/.cache/data_utils` on
import; `Path.home` is pointed into this repository's scratch directory while it loads so that nothing
outside /root/repo is touched.

Writes only data:
  g10_overlap_slices.npz   table of TimedArray._overlap_slice decisions
  g11_segment_assembly.npz inputs + expected [L, D, T] tensors of dense / word / fmri features
"""

from __future__ import annotations

import importlib.util
import sys
from pathlib import Path
from unittest import mock

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(HERE))

from make_golden import _extract_method  # noqa: E402


def load_base():
    scratch = ROOT / "gpurun_out" / "_home"
    scratch.mkdir(parents=True, exist_ok=True)
    with mock.patch("pathlib.Path.home", return_value=scratch):
        spec = importlib.util.spec_from_file_location("ref_data_utils_base", REF / "data_utils/data_utils/base.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    return mod


class _Feat:
    """Carrier of the two config fields `_aggregate_layers` reads."""

    def __init__(self, layers, layer_aggregation):
        self.layers = layers
        self.layer_aggregation = layer_aggregation


def main() -> None:
    base = load_base()
    TimedArray = base.TimedArray
    rng = np.random.default_rng(20251004)

    # ---- G10: _overlap_slice decision table -------------------------------------------------------
    rows = []
    freqs = [2.0, 1 / 1.49, 0.0, 50.0]
    for case in range(600):
        f = freqs[case % 4]
        arr_start = float(np.round(rng.uniform(-5, 40), rng.integers(0, 4)))
        n = int(rng.integers(1, 60))
        arr_dur = float(np.round(rng.uniform(0, 20), 2)) if f == 0 else None
        q_start = float(np.round(arr_start + rng.uniform(-15, 35), rng.integers(0, 4)))
        q_dur = float(np.round(rng.uniform(0, 30), rng.integers(0, 3))) if case % 7 else 0.0
        if f:
            ta = TimedArray(frequency=f, start=arr_start, data=np.zeros((2, n), dtype=np.float32))
        else:
            ta = TimedArray(frequency=0, start=arr_start, duration=arr_dur, data=np.zeros((2,), dtype=np.float32))
        try:
            got = ta._overlap_slice(q_start, q_dur)
            err = 0
        except RuntimeError:
            got, err = None, 1
        if got is None:
            rows.append([f, arr_start, n, ta.duration, q_start, q_dur, 0, 0.0, 0.0, -1, -1, err])
        else:
            s, d, sl = got
            i0, cnt = (sl.start, sl.stop - sl.start) if sl is not None else (-1, -1)
            rows.append([f, arr_start, n, ta.duration, q_start, q_dur, 1, s, d, i0, cnt, err])
    np.savez_compressed(HERE / "g10_overlap_slices.npz", table=np.asarray(rows, dtype=np.float64),
                        columns=np.asarray("frequency arr_start arr_len arr_duration q_start q_duration valid out_start out_duration "
                                           "first count raised".split()))

    # ---- G11: segment assembly --------------------------------------------------------------------
    agg = {m: _extract_method(REF / f"data_utils/data_utils/features/{m}.py", c, "_aggregate_layers")
           for m, c in (("text", "LLAMA3p2"), ("audio", "Wav2VecBert"), ("video", "VJEPA2"))}
    out: dict[str, np.ndarray] = {}
    layer_cfgs = [([0.5, 0.75, 1.0], "group_mean"), ([0.0, 0.5, 1.0], None), ([1.0], "group_mean"), ([0.75], None)]
    out["layer_cfg_layers"] = np.asarray([(c[0] + [-1.0] * 3)[:3] for c in layer_cfgs], dtype=np.float64)  # -1 pads
    out["layer_cfg_group_mean"] = np.asarray([c[1] == "group_mean" for c in layer_cfgs])

    # dense (audio / video) features: two back-to-back movie events per timeline, 2 Hz states [n_states, D, T_ev]
    n_states, D = 9, 6
    ev_start = np.asarray([3.0, 64.5])
    ev_len = np.asarray([123, 40])
    ev_dur = ev_len / 2.0 + np.asarray([0.0, 0.3])          # video passes event.duration (may disagree slightly)
    states = [rng.standard_normal((n_states, D, int(n))).astype(np.float32) for n in ev_len]
    seg_start = np.asarray([3.0, 10.25, 50.0, 60.49, 80.0, 0.0, 84.0, 200.0])
    seg_dur = np.asarray([20.0, 14.9, 30.0, 10.0, 7.0, 2.0, 10.0, 5.0])
    out.update(dense_ev_start=ev_start, dense_ev_len=ev_len, dense_ev_dur=ev_dur, dense_seg_start=seg_start, dense_seg_dur=seg_dur)
    for i, s in enumerate(states):
        out[f"dense_states{i}"] = s
    for ci, (layers, la) in enumerate(layer_cfgs):
        for flavour in ("audio", "video"):
            feat = _Feat(layers, la)
            for si, (s0, sd) in enumerate(zip(seg_start, seg_dur)):
                res = TimedArray(aggregation="sum", start=float(s0), frequency=2.0, duration=float(sd))
                for e in range(2):
                    if flavour == "audio":   # audio.py:236-251: duration from the data length
                        ta = TimedArray(data=states[e], start=float(ev_start[e]), frequency=2.0)
                    else:                    # video.py:172-189: event.duration passed (validated, then overridden)
                        ta = TimedArray(data=states[e], start=float(ev_start[e]), frequency=2.0, duration=float(ev_dur[e]))
                    sub = ta.overlap(start=float(s0), duration=float(sd))
                    if sub is None:
                        sub = ta.overlap(start=ta.start, duration=0)
                    sub.data = agg[flavour](feat, sub.data)
                    res += sub
                out[f"dense_{flavour}_cfg{ci}_seg{si}"] = np.asarray(res.data)

    # word features: frequency-0 values that hold for the word's duration (text.py:190-202)
    n_words, Dw, ns_w = 40, 5, 7
    w_start = np.sort(np.round(rng.uniform(0, 30, n_words), 2))
    w_dur = np.round(rng.uniform(0.0, 1.3, n_words), 2)
    w_dur[[3, 17]] = 0.0
    w_lat = rng.standard_normal((n_words, ns_w, Dw)).astype(np.float32)
    wseg_start = np.asarray([0.0, 4.2, 12.75, 29.0, 40.0])
    wseg_dur = np.asarray([10.0, 8.0, 16.3, 5.0, 4.0])
    out.update(word_start=w_start, word_dur=w_dur, word_states=w_lat, word_seg_start=wseg_start, word_seg_dur=wseg_dur)
    for ci, (layers, la) in enumerate(layer_cfgs):
        feat = _Feat(layers, la)
        for si, (s0, sd) in enumerate(zip(wseg_start, wseg_dur)):
            res = TimedArray(aggregation="sum", start=float(s0), frequency=2.0, duration=float(sd))
            for w in range(n_words):
                # the reference hands every word of the segment's event list to the feature, overlapping or not
                res += TimedArray(frequency=0, duration=float(w_dur[w]), start=float(w_start[w]), data=agg["text"](feat, w_lat[w]))
            out[f"word_cfg{ci}_seg{si}"] = np.asarray(res.data)

    # fmri target: whole recording at 1/1.49 Hz, shifted by -4.47 s (neuro.py:141-153), output grid 1/1.49 Hz
    V, n_tr = 7, 90
    fm = rng.standard_normal((V, n_tr)).astype(np.float32)
    f_start, f_freq = 2.0, 1 / 1.49
    fseg_start = np.asarray([2.0, 20.0, 100.0, 130.0, 0.0])
    fseg_dur = np.asarray([149.0, 29.8, 40.0, 20.0, 3.0])
    out.update(fmri_data=fm, fmri_start=np.asarray(f_start), fmri_seg_start=fseg_start, fmri_seg_dur=fseg_dur)
    for si, (s0, sd) in enumerate(zip(fseg_start, fseg_dur)):
        res = TimedArray(aggregation="sum", start=float(s0), frequency=1 / 1.49, duration=float(sd))
        res += TimedArray(data=fm, frequency=f_freq, start=f_start - 4.47, duration=n_tr * 1.49)
        out[f"fmri_seg{si}"] = np.asarray(res.data)

    np.savez_compressed(HERE / "g11_segment_assembly.npz", **out)
    print("wrote g10_overlap_slices.npz, g11_segment_assembly.npz:", len(rows), "slice rows,", len(out), "arrays")


if __name__ == "__main__":
    main()
