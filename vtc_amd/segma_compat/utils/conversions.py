"""segma.utils.conversions stand-in (used by reference src/core/intervals.py:10,95-96)."""

SAMPLE_RATE = 16_000


def frames_to_seconds(n_samples, sample_rate: int = SAMPLE_RATE) -> float:
    """The reference feeds it *sample-index* interval bounds (src/core/intervals.py:72-75,89)."""
    return n_samples / sample_rate


def seconds_to_frames(seconds: float, sample_rate: int = SAMPLE_RATE) -> int:
    return int(round(seconds * sample_rate))
