"""Clip-grid alignment: the one piece of the reference's clip builder that belongs to the VTC path
(/root/reference/src/packaging/clips.py:48-61, used at :681 when a cut point is chosen).

Windows are independent and start every T*320 samples (src/analysis/chunk_artifact.py:71-73), so a clip that is cut at
a multiple of the window step sees, on re-inference, exactly the windows the full-file run saw: its logits are the
full-file rows from cut/320 on (bit for bit on this implementation; tests/test_gpu_parity.py checks it).  Everything
else in the clip builder (cut-point search over VAD / VTC gaps, tar writing) is outside SURVEY.md section 8.
"""

from __future__ import annotations

_SR: int = 16_000
_RF_STEP: int = 320
CHUNK_STEP_F: int = 199 * _RF_STEP          # 63,680 samples: 199 frames per 4 s window
CHUNK_STEP_S: float = CHUNK_STEP_F / _SR    # 3.98 s


def chunk_step_samples(frames_per_window: int) -> int:
    """Window step for a model with `frames_per_window` frames (vtc_frames_per_window): 199 -> 63,680."""
    return int(frames_per_window) * _RF_STEP


def snap_to_grid(t: float, file_duration: float, step_s: float = CHUNK_STEP_S) -> float:
    """Nearest multiple of the window step (Python's round: ties to even), clamped to [0, file_duration]."""
    snapped = round(t / step_s) * step_s
    return max(0.0, min(snapped, file_duration))
