"""segma.config stand-in: load_config(path) -> Config (reference call site src/pipeline/vtc.py:183)."""

from __future__ import annotations

from dataclasses import fields, is_dataclass
from pathlib import Path

import yaml

from .base import Config


def _fill(obj, values: dict, where: str):
    known = {f.name: f for f in fields(obj)}
    for k, v in (values or {}).items():
        if k not in known:
            continue  # segma's YAML carries training-only keys the inference path never reads
        cur = getattr(obj, k)
        if is_dataclass(cur):
            if not isinstance(v, dict):
                raise ValueError(f"{where}.{k} must be a mapping")
            _fill(cur, v, f"{where}.{k}")
        else:
            setattr(obj, k, type(cur)(v) if not isinstance(cur, list) else list(v))


def load_config(path) -> Config:
    path = Path(path)
    with open(path, "r") as f:
        raw = yaml.safe_load(f) or {}
    if not isinstance(raw, dict):
        raise ValueError(f"{path}: top level of a model config must be a mapping")
    cfg = Config()
    _fill(cfg, raw, "config")
    if cfg.audio.chunk_duration_s <= 0:
        raise ValueError("audio.chunk_duration_s must be positive")
    return cfg
