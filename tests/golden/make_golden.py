#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the BUILD container, not on the GPU box).

What pins what:
  * encoder restatement  <- transformers.HubertModel (local install, random-init from a local HubertConfig,
    never from_pretrained) loaded with the same seeded state dict; max |diff| recorded in golden_encoder.json.
  * meta rows            <- the reference's own src/core/metadata.py imported from /root/reference
    (stdlib-only module; it imported without objection).
  * geometry KAT         <- constants printed in /root/reference/src/analysis/chunk_artifact.py:5-13,60-83.
  * decode / segments / merge / shard KATs are hand-derived from the closed-form rules;
    segma itself is absent, so those stay "parity unpinned" against the real package.
Outputs: tests/golden/*.json, tests/golden/encoder_1s.npz
"""

from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))

from oracle import vtc_oracle as O  # noqa: E402


def encoder_golden():
    w = O.make_weights(seed=0)
    rng = np.random.default_rng(1)
    wav = rng.standard_normal((2, 16000), dtype=np.float32) * np.float32(0.1)
    wav[1, 9000:] = 0.0
    with torch.no_grad():
        st = O.encoder_stages(w, torch.from_numpy(wav))
    info = {"seed_weights": 0, "seed_input": 1, "input": "default_rng(1).standard_normal((2,16000))*0.1; row 1 zero from 9000"}
    try:
        from transformers import HubertConfig, HubertModel
        m = HubertModel(HubertConfig()).eval()
        missing, unexpected = m.load_state_dict({k: v for k, v in w.items() if not k.startswith("head.")}, strict=False)
        assert unexpected == [] and missing == ["masked_spec_embed"], (missing, unexpected)
        with torch.no_grad():
            ref = m(torch.from_numpy(wav)).last_hidden_state
        info["hf_hubert_max_abs_diff_last_hidden"] = float((ref - st["layer11"]).abs().max())
        import transformers
        info["transformers_version"] = transformers.__version__
        assert info["hf_hubert_max_abs_diff_last_hidden"] <= 1e-5
    except ImportError:
        info["hf_hubert_max_abs_diff_last_hidden"] = None
    arrays = {"logits": st["logits"].numpy()}
    for k, v in st.items():
        if k == "logits":
            continue
        a = v.numpy()
        arrays[k + "_slice"] = a[:, :8, :8].copy()
        arrays[k + "_sum"] = np.array([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum()])
    np.savez_compressed(HERE / "encoder_1s.npz", **arrays)
    (HERE / "golden_encoder.json").write_text(json.dumps(info, indent=1))
    print("encoder golden:", info)


def meta_golden():
    sys.path.insert(0, "/root/reference")
    from src.core import metadata as ref  # the reference's own module (stdlib only)
    cases = [
        ("uid1", 0.4, [{"label": "KCHI", "duration": 1.0}, {"label": "KCHI", "duration": 2.0},
                       {"label": "FEM", "duration": 3.0}], 0.9, 0.6),      # tests/test_metadata.py:30-43
        ("uid2", 0.5, [], 0.0, 0.0),                                        # tests/test_metadata.py:45-49
        ("f", 0.5, [{"label": "OCH", "duration": 0.123}, {"label": "MAL", "duration": 0.4565}], 0.9991, 0.0312),
    ]
    out = {"meta_rows": [], "error_rows": []}
    for c in cases:
        out["meta_rows"].append({"args": list(c), "row": ref.vtc_meta_row(*c)})
    for uid, err in (("x", "boom"), ("file001", "some error")):
        row = ref.vtc_error_row(uid, err)
        out["error_rows"].append({"args": [uid, err],
                                  "row": {k: (None if isinstance(v, float) and v != v else v) for k, v in row.items()}})
    (HERE / "golden_meta.json").write_text(json.dumps(out, indent=1))
    print("meta golden from the reference module:", len(out["meta_rows"]), "+", len(out["error_rows"]))


def kat_golden():
    kat = {
        # chunk_artifact.py:5-13,60-83 and SURVEY 8c G1
        "geometry": {"n_frames": {"399": 0, "400": 1, "4800": 14, "64000": 199, "160000": 499},
                     "N_WINDOWS_4s": 199, "CHUNK_STEP_F": 63680, "FULL_CHUNK_DUR_F": 63760,
                     "peak_seconds": {"1": 63760 / 16000, "2": (397 * 320 + 400) / 16000, "3": (596 * 320 + 400) / 16000,
                                      "4": (795 * 320 + 400) / 16000, "5": (994 * 320 + 400) / 16000}},
        # G3: masks (frames x 2 labels "A","B") -> triples, label-major then time
        "decode": [
            {"mask": [[0, 0], [1, 0], [1, 1], [0, 1], [1, 1]], "labels": ["A", "B"],
             "intervals": [[320, 1040, "A"], [1280, 1680, "A"], [640, 1680, "B"]]},
            {"mask": [[1, 1]], "labels": ["A", "B"], "intervals": [[0, 400, "A"], [0, 400, "B"]]},
            {"mask": [[0, 0], [0, 0]], "labels": ["A", "B"], "intervals": []},
        ],
        # G4: tests/test_intervals.py:153-168
        "segments": [
            {"intervals": [[0, 16000, "KCHI"]], "uid": "file001",
             "rows": [{"uid": "file001", "onset": 0.0, "offset": 1.0, "duration": 1.0, "label": "KCHI"}]},
            {"intervals": [[0, 0, "FEM"]], "uid": "x", "rows": []},
            {"intervals": [[320, 1040, "MAL"]], "uid": "u",
             "rows": [{"uid": "u", "onset": 0.02, "offset": 0.065, "duration": 0.045, "label": "MAL"}]},
        ],
        # G6: collar 0.3, min_on 0.1 (src/utils.py:323-390)
        "merge": {
            "collar": 0.3, "min_on": 0.1,
            "rows": [
                ["a", "KCHI", 0.0, 1.0], ["a", "KCHI", 1.29, 2.0],     # gap 0.29 -> merged
                ["a", "KCHI", 2.3, 3.0],                                # gap 0.30 -> merged (not > collar)
                ["a", "KCHI", 3.31, 4.0],                               # gap 0.31 -> new group
                ["a", "FEM", 0.0, 10.0], ["a", "FEM", 2.0, 5.0], ["a", "FEM", 10.2, 10.25],   # nested + collar
                ["a", "MAL", 5.0, 5.09],                                # 0.09 -> dropped
                ["a", "OCH", 5.0, 5.1],                                 # 0.10 -> kept
                ["b", "KCHI", 7.0, 7.5], ["b", "KCHI", 0.5, 0.9],       # other uid, unsorted input
            ],
            "expected": [
                ["a", "FEM", 0.0, 10.25, 10.25], ["a", "KCHI", 0.0, 3.0, 3.0], ["a", "KCHI", 3.31, 4.0, 0.69],
                ["a", "OCH", 5.0, 5.1, 0.1], ["b", "KCHI", 0.5, 0.9, 0.4], ["b", "KCHI", 7.0, 7.5, 0.5],
            ],
        },
        # G7: src/utils.py:278-285
        "shard": {"n": 10, "count": 3, "slices": [[0, 1, 2, 3], [4, 5, 6], [7, 8, 9]]},
    }
    (HERE / "golden_kat.json").write_text(json.dumps(kat, indent=1))
    print("KAT golden written")


if __name__ == "__main__":
    encoder_golden()
    meta_golden()
    kat_golden()
