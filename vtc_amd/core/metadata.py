"""Per-file VTC metadata rows (schema of the reference's src/core/metadata.py:12-50, which downstream
readers depend on: src/pipeline/package.py:68-77, dataloader/adapters/vtc.py:97-126)."""

from __future__ import annotations

import json
import math

META_COLUMNS = ("uid", "vtc_threshold", "vtc_speech_dur", "vtc_n_segments", "vtc_label_counts",
                "vtc_max_sigmoid", "vtc_mean_sigmoid", "error")


def vtc_error_row(uid: str, error: str) -> dict:
    """Row of a file whose inference raised: NaN statistics, zero segments, the message in `error`."""
    return {
        "vtc_threshold": math.nan, "vtc_speech_dur": 0.0, "vtc_n_segments": 0, "vtc_label_counts": "{}",
        "vtc_max_sigmoid": math.nan, "vtc_mean_sigmoid": math.nan, "error": error, "uid": uid,
    }


def vtc_meta_row(uid: str, threshold: float, segments: list, max_sigmoid: float, mean_sigmoid: float) -> dict:
    """Row of a processed file: total speech seconds (3 dp), segment count and per-label counts as JSON."""
    per_label: dict = {}
    total = 0.0
    for seg in segments:
        total += seg["duration"]
        per_label[seg["label"]] = per_label.get(seg["label"], 0) + 1
    return {
        "uid": uid, "vtc_threshold": threshold, "vtc_speech_dur": round(total, 3),
        "vtc_n_segments": len(segments), "vtc_label_counts": json.dumps(per_label),
        "vtc_max_sigmoid": max_sigmoid, "vtc_mean_sigmoid": mean_sigmoid, "error": "",
    }


def vtc_meta_row_columns(uid: str, threshold: float, durations, label_codes, labels, max_sigmoid: float,
                         mean_sigmoid: float) -> dict:
    """vtc_meta_row from column arrays (durations float64, label_codes int, in segment order): the same left-to-right
    sum of durations and the same first-appearance order of the labels in the JSON."""
    import numpy as np

    n = int(len(durations))
    per_label: dict = {}
    if n <= 4096:
        # the usual case (a recording's few dozen segments): plain Python on lists is the same sequential float64 sum and
        # the same first-appearance order, at a fifth of the cost of numpy's small-array calls (this runs per file, under
        # the GIL the reader and GPU threads share)
        total = 0.0
        for d in np.asarray(durations, dtype=np.float64).tolist():
            total += d
        for c in np.asarray(label_codes).tolist():
            name = labels[c]
            per_label[name] = per_label.get(name, 0) + 1
    else:
        total = float(np.cumsum(np.asarray(durations, dtype=np.float64))[-1])   # cumsum adds sequentially
        codes, first, counts = np.unique(np.asarray(label_codes), return_index=True, return_counts=True)
        for k in np.argsort(first, kind="stable"):
            per_label[labels[int(codes[k])]] = int(counts[k])
    return {
        "uid": uid, "vtc_threshold": threshold, "vtc_speech_dur": round(total, 3),
        "vtc_n_segments": n, "vtc_label_counts": json.dumps(per_label),
        "vtc_max_sigmoid": max_sigmoid, "vtc_mean_sigmoid": mean_sigmoid, "error": "",
    }
