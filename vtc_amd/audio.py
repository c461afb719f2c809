"""Audio front-end of the hot path: file -> mono 16 kHz float32 in [-1, 1).

The reference leaves decoding to segma (torchaudio/torchcodec, requirements.txt:393-394); the rule for
non-16 kHz or multi-channel input is unpinned.  Here: WAV (PCM 8/16/24/32, float)
through scipy with mmap; channels are averaged; other rates are resampled with a polyphase filter.
"""

from __future__ import annotations

from math import gcd
from pathlib import Path

import numpy as np

SAMPLE_RATE = 16_000


def to_float32(data: np.ndarray) -> np.ndarray:
    if data.dtype == np.int16:
        out = data.astype(np.float32) / np.float32(32768.0)
    elif data.dtype == np.int32:
        out = (data.astype(np.float64) / 2147483648.0).astype(np.float32)
    elif data.dtype == np.uint8:
        out = (data.astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
    elif data.dtype in (np.float32, np.float64):
        out = data.astype(np.float32, copy=False)
    else:
        raise ValueError(f"unsupported sample type {data.dtype}")
    return out


def read_audio(path) -> np.ndarray:
    """Decode one file to mono 16 kHz float32.  Raises on anything unreadable; the host turns the
    exception into an error row (reference src/pipeline/vtc.py:245-249)."""
    from scipy.io import wavfile

    path = Path(path)
    if path.suffix.lower() not in (".wav", ".wave"):
        raise ValueError(f"{path.name}: only WAV input is supported by this build (got '{path.suffix}')")
    try:
        sr, data = wavfile.read(str(path), mmap=True)
    except ValueError:
        sr, data = wavfile.read(str(path), mmap=False)   # 24-bit PCM cannot be memory-mapped
    x = to_float32(np.asarray(data))
    if x.ndim == 2:
        x = x.mean(axis=1, dtype=np.float32)
    if sr != SAMPLE_RATE:
        from scipy.signal import resample_poly
        g = gcd(SAMPLE_RATE, int(sr))
        x = resample_poly(x, SAMPLE_RATE // g, int(sr) // g).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


def write_wav(path, samples: np.ndarray, sample_rate: int = SAMPLE_RATE) -> None:
    """Write float audio as 16-bit PCM (the format of the synthetic benchmark files)."""
    from scipy.io import wavfile

    pcm = np.clip(np.round(np.asarray(samples, dtype=np.float64) * 32768.0), -32768, 32767).astype(np.int16)
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    wavfile.write(str(path), sample_rate, pcm)
