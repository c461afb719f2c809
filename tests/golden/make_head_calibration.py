#!/usr/bin/env python3
"""Head calibration of the synthetic weight sets ("head scaled so logit std ~ 2").

For every (family, seed) the fp32 CPU oracle runs the uncalibrated weights of vtc_amd.synth.make_weights on the calibration
clip (synth_audio(40 s, seed 9000), 10 s windows, full windows only) and records the std of the logits pooled over
frames and labels (family "trained": after removing the per-label mean, which is recorded too).  make_weights(..., calibrated=True) multiplies head.weight and head.bias by 2 / that std, so the oracle's
logit std on the clip is exactly 2.  Writes tests/golden/head_calibration.json and prints the HEAD_CALIBRATION dict to paste into
vtc_amd/synth.py (the product package cannot import from tests/ or oracle/).

    python tests/golden/make_head_calibration.py
"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import vtc_oracle as O  # noqa: E402
from vtc_amd import synth  # noqa: E402

CLIP_SECONDS, CLIP_SEED, CHUNK = 40.0, 9000, 10.0


def main():
    torch.set_num_threads(8)
    audio = synth.synth_audio(CLIP_SECONDS, seed=CLIP_SEED)
    plan = O.plan_windows(len(audio), CHUNK)
    n = len(plan.starts) - 1
    x = torch.from_numpy(np.stack([audio[s: s + plan.window] for s in plan.starts[:n]]))
    table, doc = {}, {"clip": {"seconds": CLIP_SECONDS, "seed": CLIP_SEED, "chunk_duration_s": CHUNK, "windows": n},
                      "target_logit_std": 2.0, "entries": {}}
    center = {}
    for family in ("stress", "hubert", "trained"):
        for seed in range(8 if family == "hubert" else 4):   # the prescribed family on eight seeds
            w = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_weights(seed=seed, family=family).items()}
            with torch.no_grad():
                logits = O.encoder_logits(w, x).reshape(-1, 4)
            entry = {}
            if family == "trained":      # a trained head is balanced: remove the per-label mean first (head.bias -= mean)
                mean = logits.mean(dim=0)
                logits = logits - mean
                center[f"{family}:{seed}"] = [round(float(v), 6) for v in mean]
                entry["label_mean_logit"] = center[f"{family}:{seed}"]
            raw = float(logits.std())
            table[f"{family}:{seed}"] = round(raw, 6)
            entry.update({"uncalibrated_logit_std": round(raw, 6), "head_scale": round(2.0 / raw, 6)})
            doc["entries"][f"{family}:{seed}"] = entry
            print(family, seed, entry, flush=True)
    (Path(__file__).resolve().parent / "head_calibration.json").write_text(json.dumps(doc, indent=1) + "\n")
    print("HEAD_CALIBRATION: dict = " + json.dumps(table))
    print("HEAD_CENTER: dict = " + json.dumps(center))


if __name__ == "__main__":
    main()
