"""ORACLE (test infrastructure, never imported by the product path): numpy restatement of how the reference cuts a
segment's feature tensor out of cached extractor outputs.

Follows /root/reference/data_utils/data_utils/base.py:64-211 (`TimedArray`: constructor, `_overlap_slice`, `overlap`,
`__iadd__` with aggregation="sum") and the assembly loops of the feature plugins
(features/text.py:85-124,190-202; audio.py:78-120,236-251; video.py:172-189; neuro.py:60-106,141-153).
Pinned by tests/golden/g10_overlap_slices.npz and g11_segment_assembly.npz, both produced by running the reference's
own base.py (tests/golden/make_golden_timeline.py).

Written as plain functions over (start, sample-count, duration) triples instead of a class: the state a `TimedArray`
carries is just those three numbers plus its data.
"""

from __future__ import annotations

import typing as tp

import numpy as np

from .tribe_ref import aggregate_layers


def to_ind(freq: float, seconds: float) -> int:
    """base.py:48-52: round-half-even of seconds * frequency."""
    return int(round(seconds * freq))


def sampled_duration(freq: float, n: int) -> float:
    """base.py:109-110: an array sampled at `freq` lasts n / freq seconds whatever duration was passed in."""
    return n / freq


def check_sampled(freq: float, n: int, duration: float | None) -> None:
    """base.py:96-108: the constructor's shape validation for a sampled array with an explicit duration."""
    if duration is None:
        return
    expected = max(1, to_ind(freq, duration))
    if n == 0:
        raise ValueError("Last dimension is empty but frequency is not null")
    if abs(n - expected) > 2:
        raise ValueError(f"Data has incorrect (last) dimension for duration {duration} and frequency {freq} (expected {expected})")


def overlap_slice(freq: float, arr_start: float, arr_len: int, arr_duration: float, q_start: float,
                  q_duration: float) -> tuple[float, float, int, int] | None:
    """base.py:164-196.  Returns (start_sec, duration_sec, first, count); first = count = -1 when freq == 0."""
    if q_duration < 0:
        raise ValueError(f"duration should be >=0, got duration={q_duration}")
    lo = max(q_start, arr_start)
    hi = min(q_start + q_duration, arr_start + arr_duration)
    if hi < lo:
        return None
    if hi == lo and arr_duration and q_duration:
        return None
    if not freq:
        return lo, hi - lo, -1, -1
    first = to_ind(freq, lo - arr_start)
    count = to_ind(freq, hi - lo)
    if count <= 0:
        count = 1
    if first > arr_len - count:
        first = arr_len - count
    if first < 0:
        raise RuntimeError("overlap start before the array")
    return first / freq + arr_start, count / freq, first, count


def _accumulate(out: np.ndarray | None, out_freq: float, out_start: float, out_len: int, piece: np.ndarray, piece_freq: float,
                piece_start: float, piece_duration: float) -> np.ndarray | None:
    """`out += piece` for aggregation="sum" (base.py:130-162).  `out` is None until the first piece fixes its shape."""
    out_duration = sampled_duration(out_freq, out_len)
    if piece_freq and out_freq != piece_freq:
        if abs(out_freq - piece_freq) * max(out_duration, piece_duration) >= 0.5:
            raise ValueError("Cannot add with different (non-0) frequencies")
    if out is None:
        lead = piece.shape[:-1] if piece_freq else piece.shape
        out = np.zeros(lead + (out_len,), dtype=piece.dtype)
    mine = overlap_slice(out_freq, out_start, out_len, out_duration, piece_start, piece_duration)
    n_piece = piece.shape[-1] if piece_freq else 0
    theirs = overlap_slice(piece_freq, piece_start, n_piece, piece_duration, out_start, out_duration)
    if mine is None or theirs is None:
        return out
    a0, an = mine[2], mine[3]
    if piece_freq:
        b0, bn = theirs[2], theirs[3]
        out[..., a0:a0 + an] += piece[..., b0:b0 + bn]   # numpy raises if the two rounded lengths cannot broadcast
    else:
        out[..., a0:a0 + an] += piece[..., None]
    return out


def out_len(freq: float, duration: float) -> int:
    """base.py:83-89: number of samples of the (initially empty) output array."""
    return max(1, to_ind(freq, duration))


def assemble_dense(events: tp.Sequence[tuple[float, np.ndarray, float | None]], seg_start: float, seg_duration: float,
                   layers: tp.Sequence[float], layer_aggregation: str | None, freq: float = 2.0) -> np.ndarray | None:
    """Audio (duration None, audio.py:236-251) / video (event duration given, video.py:172-189) feature of one segment.
    events: (event_start, states [n_states, D, T_event], event_duration or None).  -> [L, D, T] (or None without events)."""
    n_out = out_len(freq, seg_duration)
    out = None
    for ev_start, states, ev_duration in events:
        n = states.shape[-1]
        check_sampled(freq, n, ev_duration)
        dur = sampled_duration(freq, n)
        sub = overlap_slice(freq, ev_start, n, dur, seg_start, seg_duration)
        if sub is None:
            sub = overlap_slice(freq, ev_start, n, dur, ev_start, 0.0)
        s_start, s_dur, first, count = sub
        check_sampled(freq, count, s_dur)
        piece = aggregate_layers(states[..., first:first + count], layers, layer_aggregation)
        out = _accumulate(out, freq, seg_start, n_out, piece, freq, s_start, sampled_duration(freq, count))
    return out


def assemble_words(word_start: np.ndarray, word_duration: np.ndarray, word_states: np.ndarray, seg_start: float, seg_duration: float,
                   layers: tp.Sequence[float], layer_aggregation: str | None, freq: float = 2.0) -> np.ndarray | None:
    """Text feature of one segment (text.py:85-124,190-202).  word_states [n_words, n_states, D] -> [L, D, T]."""
    n_out = out_len(freq, seg_duration)
    out = None
    for w in range(len(word_start)):
        piece = aggregate_layers(word_states[w], layers, layer_aggregation)
        out = _accumulate(out, freq, seg_start, n_out, piece, 0.0, float(word_start[w]), float(word_duration[w]))
    return out


def assemble_fmri(data: np.ndarray, rec_start: float, seg_start: float, seg_duration: float, tr: float = 1.49,
                  shift: float = 4.47) -> np.ndarray:
    """fMRI target of one segment (neuro.py:60-106,141-153): the whole recording, moved `shift` seconds earlier."""
    freq = 1 / tr
    n_out = out_len(freq, seg_duration)
    n = data.shape[-1]
    check_sampled(freq, n, n * tr)
    return _accumulate(None, freq, seg_start, n_out, data, freq, rec_start - shift, sampled_duration(freq, n))
