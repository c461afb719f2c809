"""segma.inference stand-in (reference call sites src/pipeline/vtc.py:236-243, 84-91).

apply_model_on_audio -> libvtc_amd vtc_infer_waveform (GPU)
apply_thresholds     -> records the per-label bounds, mapped exactly to logit space
create_intervals     -> libvtc_amd vtc_decode (GPU threshold + run-length)
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from ..audio import read_pcm
from ..engine import RF_SIZE

_LAST_ENGINE = None   # engine of the most recent apply_model_on_audio call (create_intervals runs on it)


def apply_model_on_audio(audio_path, model, conv_settings=None, device="cuda", batch_size: int = 128,
                         chunk_duration_s: float = 4.0) -> torch.Tensor:
    """File -> (n_frames, n_labels) float32 logits on `device`.

    Windows of chunk_duration_s are stepped by T*320 samples and their frames concatenated (global frame
    g <-> sample g*320; reference src/analysis/chunk_artifact.py:60-83).  Raises on unreadable or too
    short audio; the reference host converts that into an error row (vtc.py:245-249)."""
    global _LAST_ENGINE
    pcm, sr = read_pcm(Path(audio_path))
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    eng = model.engine(chunk_duration_s, batch_size, dev)
    _LAST_ENGINE = eng
    host = torch.from_numpy(np.ascontiguousarray(pcm))
    gpu = eng.prepare_audio(host.pin_memory().to(eng.device, non_blocking=True) if host.numel() >= (1 << 20) else host.to(eng.device), sr)
    if gpu.numel() < RF_SIZE:
        raise ValueError(f"audio too short: {gpu.numel()} samples < receptive field {RF_SIZE}")
    return eng.infer_waveform(gpu, batch_size)


def _sorted_f32(idx: np.ndarray) -> np.ndarray:
    """Order-preserving map from uint32 rank to float32 (rank 0 = most negative)."""
    idx = idx.astype(np.uint32)
    bits = np.where(idx & np.uint32(0x80000000), idx & np.uint32(0x7FFFFFFF), ~idx)
    return bits.astype(np.uint32).view(np.float32)


def _boundary(prob: float, want_ge: bool) -> float:
    """Smallest float32 x with sigmoid(x) >= prob (want_ge) or largest x with sigmoid(x) <= prob, using the
    same fp32 torch.sigmoid the reference's CPU path applies, so logit-space masks equal prob-space masks."""
    p = np.float32(prob)
    lo_r, hi_r = int(0x00800000), int(0xFF7FFFFF)   # ranks of -FLT_MAX .. +FLT_MAX (skip NaN/inf patterns)

    def sig(rank: int) -> np.float32:
        # 64 copies so that torch takes its vectorised kernel, the one a whole logits tensor goes through
        # (its scalar tail path can differ by an ulp, so agreement with torch is exact up to that ulp)
        x = torch.from_numpy(np.repeat(_sorted_f32(np.array([rank], dtype=np.uint32)), 64))
        return np.float32(torch.sigmoid(x)[0].item())

    if want_ge:
        if sig(hi_r) < p:
            return float("inf")
        if sig(lo_r) >= p:
            return float("-inf")
        while hi_r - lo_r > 1:          # invariant: sig(lo_r) < p <= sig(hi_r)
            mid = (lo_r + hi_r) // 2
            if sig(mid) >= p:
                hi_r = mid
            else:
                lo_r = mid
        return float(_sorted_f32(np.array([hi_r], dtype=np.uint32))[0])
    if sig(lo_r) > p:
        return float("-inf")
    if sig(hi_r) <= p:
        return float("inf")
    while hi_r - lo_r > 1:              # invariant: sig(lo_r) <= p < sig(hi_r)
        mid = (lo_r + hi_r) // 2
        if sig(mid) <= p:
            lo_r = mid
        else:
            hi_r = mid
    return float(_sorted_f32(np.array([lo_r], dtype=np.uint32))[0])


_BOUND_CACHE: dict = {}


def logit_bounds(thresholds: dict, labels) -> tuple[np.ndarray, np.ndarray]:
    """Per-label [lower_bound, upper_bound] on sigmoid(logit) -> inclusive fp32 bounds on the logit."""
    lo, hi = [], []
    for lab in labels:
        lb, ub = float(thresholds[lab]["lower_bound"]), float(thresholds[lab]["upper_bound"])
        for key in (("ge", lb), ("le", ub)):
            if key not in _BOUND_CACHE:
                _BOUND_CACHE[key] = _boundary(key[1], key[0] == "ge")
        lo.append(_BOUND_CACHE[("ge", lb)])
        hi.append(_BOUND_CACHE[("le", ub)])
    return np.asarray(lo, dtype=np.float32), np.asarray(hi, dtype=np.float32)


class Thresholded:
    """What apply_thresholds hands to create_intervals: the logits plus logit-space bounds."""

    def __init__(self, logits: torch.Tensor, thresholds: dict):
        self.logits = logits
        self.thresholds = thresholds

    def detach(self):
        return self

    def cpu(self):
        return self


def apply_thresholds(logits: torch.Tensor, thresholds: dict, device="cpu") -> Thresholded:
    return Thresholded(logits.detach(), thresholds)


def create_intervals(thresholded: Thresholded, conv_settings, l_encoder, engine=None) -> list:
    """[(start_sample, end_sample, label)] for every run of active frames i..j:
    (i*320, j*320 + 400) (reference src/analysis/chunk_artifact.py:11-13,75-83); label-major, time-ordered."""
    eng = engine or _LAST_ENGINE
    if eng is None:
        raise RuntimeError("create_intervals needs a live VtcEngine (call apply_model_on_audio first)")
    logits = thresholded.logits
    if logits.numel() == 0:
        return []
    lo, hi = logit_bounds(thresholded.thresholds, l_encoder.labels)
    dev_logits = logits.to(eng.device, torch.float32).contiguous()
    s, e, lab, _, _ = eng.decode(dev_logits, lo, hi)
    names = l_encoder.labels
    return [(int(a), int(b), names[int(c)]) for a, b, c in zip(s, e, lab)]
