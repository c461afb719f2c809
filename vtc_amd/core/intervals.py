"""Sample-index intervals -> second-based segment rows.

Mirrors the behaviour of the reference's src/core/intervals.py:72-108 (intervals_to_pairs,
intervals_to_segments, merge_pairs): seconds = samples / 16000, every bound rounded to 3 decimals,
rows with non-positive duration dropped.
"""

from __future__ import annotations

import numpy as np

from ..segma_compat.utils.conversions import frames_to_seconds


def _to_seconds(sample_index) -> float:
    return round(float(frames_to_seconds(sample_index)), 3)


def merge_pairs(pairs):
    """Union of (onset, offset) intervals; touching intervals fuse (reference intervals.py:18-32)."""
    merged: list = []
    for on, off in sorted(pairs):
        if merged and on <= merged[-1][1]:
            if off > merged[-1][1]:
                merged[-1] = (merged[-1][0], off)
        else:
            merged.append((on, off))
    return merged


def total_duration(pairs) -> float:
    return sum(off - on for on, off in pairs)


def intervals_to_pairs(intervals):
    secs = [(_to_seconds(s), _to_seconds(e)) for s, e, _ in intervals]
    return merge_pairs([p for p in secs if p[1] > p[0]])


def intervals_to_segments(intervals, uid: str) -> list[dict]:
    """One dict {uid, onset, offset, duration, label} per interval with duration > 0."""
    segments = []
    for start, end, label in intervals:
        onset, offset = _to_seconds(start), _to_seconds(end)
        duration = round(offset - onset, 3)
        if duration <= 0:
            continue
        segments.append({"uid": uid, "onset": onset, "offset": offset, "duration": duration, "label": label})
    return segments


def intervals_to_columns(starts, ends):
    """Array form of intervals_to_segments for the decoder's output (sample indices): (onset, offset, duration, keep)
    as float64 arrays and the duration > 0 mask.  Decoder bounds are whole frames (start = 320 i, end = 320 j + 400),
    i.e. exact 3-decimal numbers of seconds, for which rint(x * 1000) / 1000 is the same double as Python's
    round(x, 3); tests/test_host.py checks the equality against the scalar form."""
    onset = np.rint(np.asarray(starts, dtype=np.float64) / 16000.0 * 1000.0) / 1000.0
    offset = np.rint(np.asarray(ends, dtype=np.float64) / 16000.0 * 1000.0) / 1000.0
    duration = np.rint((offset - onset) * 1000.0) / 1000.0
    return onset, offset, duration, duration > 0
