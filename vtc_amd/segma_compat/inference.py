"""segma.inference stand-in (reference call sites src/pipeline/vtc.py:236-243, 84-91).

apply_model_on_audio -> libvtc_amd vtc_infer_waveform (GPU)
apply_thresholds     -> libvtc_amd vtc_threshold_mask (GPU; bounds mapped exactly to logit space): a real 0/1 uint8 tensor
create_intervals     -> libvtc_amd vtc_decode (GPU threshold + run-length)
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from ..audio import read_pcm
from ..engine import RF_SIZE, live_engines


def apply_model_on_audio(audio_path, model, conv_settings=None, device="cuda", batch_size: int = 128,
                         chunk_duration_s: float = 4.0) -> torch.Tensor:
    """File -> (n_frames, n_labels) float32 logits on `device`.

    Windows of chunk_duration_s are stepped by T*320 samples and their frames concatenated (global frame
    g <-> sample g*320; reference src/analysis/chunk_artifact.py:60-83).  Raises on unreadable or too
    short audio; the reference host converts that into an error row (vtc.py:245-249)."""
    pcm, sr = read_pcm(Path(audio_path))
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    eng = model.engine(chunk_duration_s, batch_size, dev)
    host = torch.from_numpy(np.ascontiguousarray(pcm))
    gpu = eng.prepare_audio(host.pin_memory().to(eng.device, non_blocking=True) if host.numel() >= (1 << 20) else host.to(eng.device), sr)
    if gpu.numel() < RF_SIZE:
        raise ValueError(f"audio too short: {gpu.numel()} samples < receptive field {RF_SIZE}")
    logits = eng.infer_waveform(gpu, batch_size)
    # the device copy rides along through the host's .cpu(): thresholds and decode then run on it, nothing is re-uploaded
    return _track(logits, {"engine": eng, "dev": logits, "version": logits._version})


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


class _Tracked(torch.Tensor):
    """A plain tensor to every consumer (ops on it return ordinary tensors, torch.save stores an ordinary tensor slice)
    that remembers where it came from on the device: the logits apply_model_on_audio returns keep their device copy
    across the host's `.cpu()` (reference vtc.py:252), and the mask apply_thresholds returns keeps the device logits and
    the logit-space bounds, so create_intervals decodes on the GPU without uploading anything again."""
    __torch_function__ = torch._C._disabled_torch_function_impl

    def cpu(self, *a, **k):
        out = super().cpu(*a, **k)
        return self if out is self else _track(out, getattr(self, "_vtc", None))

    def detach(self):
        return _track(super().detach(), getattr(self, "_vtc", None))


def _track(t: torch.Tensor, info) -> torch.Tensor:
    t = t.as_subclass(_Tracked)
    t._vtc = info
    return t


def _engine_for(t: torch.Tensor, engine=None):
    """Engine for a tensor that carries no provenance (made or changed by the caller): the newest open engine with as
    many labels as the tensor has columns, on the tensor's device if it lives on a GPU."""
    if engine is not None:
        return engine
    found = live_engines(t.shape[1] if t.dim() == 2 else None, t.device if t.is_cuda else None)
    if not found:
        raise RuntimeError("no live VtcEngine for this tensor: load a model and run apply_model_on_audio first "
                           "(vtc_amd has no CPU path)")
    return found[0]


def apply_thresholds(logits: torch.Tensor, thresholds: dict, device="cpu") -> torch.Tensor:
    """(n_frames, n_labels) uint8 mask on `device`: 1 where lower_bound <= sigmoid(logit) <= upper_bound of the column's
    label (reference call site vtc.py:84-90; columns follow the order of the `thresholds` dict, which the reference builds
    from l_encoder.labels).  Computed on the GPU (vtc_threshold_mask) from the device copy of the logits."""
    info = getattr(logits, "_vtc", None)
    labels = tuple(thresholds.keys())
    if logits.dim() != 2 or logits.shape[1] != len(labels):
        raise ValueError(f"logits of shape {tuple(logits.shape)} do not match {len(labels)} thresholded labels")
    if info is not None and logits._version == info["version"]:
        eng, dev_logits = info["engine"], info["dev"]
    else:
        eng = _engine_for(logits)
        dev_logits = logits.detach().to(eng.device, torch.float32).contiguous()
    lo, hi = logit_bounds(thresholds, labels)
    if logits.shape[0] == 0:
        return torch.zeros((0, len(labels)), dtype=torch.uint8, device=device)
    mask = eng.threshold_mask(dev_logits, lo, hi).to(device)
    return _track(mask, {"engine": eng, "dev": dev_logits, "lo": lo, "hi": hi, "labels": labels, "version": mask._version})


def create_intervals(thresholded: torch.Tensor, conv_settings, l_encoder, engine=None) -> list:
    """[(start_sample, end_sample, label)] for every run of active frames i..j:
    (i*320, j*320 + 400) (reference src/analysis/chunk_artifact.py:11-13,75-83); label-major, time-ordered.
    A mask that still is what apply_thresholds returned is decoded from the device logits it carries (vtc_decode); a mask
    the caller made or changed is uploaded as it is and decoded on the GPU all the same."""
    names = list(l_encoder.labels)
    if thresholded.numel() == 0:
        return []
    info = getattr(thresholded, "_vtc", None)
    if info is not None and "lo" in info and thresholded._version == info["version"] and tuple(names) == info["labels"]:
        eng, dev, lo, hi = engine or info["engine"], info["dev"], info["lo"], info["hi"]
    else:
        eng = _engine_for(thresholded, engine)
        dev = (thresholded.detach().to(eng.device) != 0).to(torch.float32).contiguous()    # 1.0 / 0.0
        lo, hi = np.full(len(names), 0.5, np.float32), np.full(len(names), np.inf, np.float32)
    s, e, lab, _, _ = eng.decode(dev, lo, hi)
    return [(int(a), int(b), names[int(c)]) for a, b, c in zip(s, e, lab)]
