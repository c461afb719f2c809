"""Drop-in for the part of the `segma` package that the reference's VTC step imports
(/root/reference/src/pipeline/vtc.py:41-47, src/core/intervals.py:10):

    segma.config.load_config            -> vtc_amd.segma_compat.config.load_config
    segma.config.base.Config            -> vtc_amd.segma_compat.config.base.Config
    segma.inference.apply_model_on_audio, apply_thresholds, create_intervals
    segma.models.Models
    segma.utils.encoders.MultiLabelEncoder
    segma.utils.conversions.frames_to_seconds

`install_as_segma()` registers these modules under the name `segma` so the unmodified reference host
code imports them.
"""

from __future__ import annotations

import sys


def install_as_segma() -> None:
    import importlib
    names = ["", ".config", ".config.base", ".inference", ".models", ".utils", ".utils.encoders", ".utils.conversions"]
    for n in names:
        sys.modules["segma" + n] = importlib.import_module(__name__ + n)
