"""segma.config.base stand-in: the three fields the host reads (reference src/pipeline/vtc.py:184-195)
plus the engine knobs of this implementation."""

from __future__ import annotations

from dataclasses import dataclass, field

DEFAULT_CLASSES = ["FEM", "MAL", "KCHI", "OCH"]  # reference src/core/__init__.py:7


@dataclass
class DataConfig:
    classes: list = field(default_factory=lambda: list(DEFAULT_CLASSES))


@dataclass
class ModelConfig:
    name: str = "surgical_hydra"   # "HuBERT surgical_hydra", reference src/analysis/chunk_artifact.py:40


@dataclass
class AudioConfig:
    chunk_duration_s: float = 4.0  # shipped VTC-2.0 value, chunk_artifact.py:47
    sample_rate: int = 16_000


@dataclass
class EngineConfig:
    mfma_dtype: str = "f16"        # "f16" | "bf16" MFMA operands (fp32 accumulate); DESIGN.md "Precision"
    max_batch: int = 0             # 0 = the batch_size handed to apply_model_on_audio
    synthetic_seed: int = 0        # used when the checkpoint path is "synthetic[:seed]"
    bf16_rescue: bool = True       # CLI: a file whose activations leave the fp16 range is re-run with bf16 storage


@dataclass
class Config:
    data: DataConfig = field(default_factory=DataConfig)
    model: ModelConfig = field(default_factory=ModelConfig)
    audio: AudioConfig = field(default_factory=AudioConfig)
    engine: EngineConfig = field(default_factory=EngineConfig)
