#!/usr/bin/env python3
"""Batch-size sweep on one MI355X: hot-path audio-hours/s (inputs resident, forward + decode, bench.py's step) and device
memory for 4 s and 10 s windows.  The reference's batch table (src/pipeline/resources.py:180-188) was measured on H100 /
A40; this is where the MI355X table of vtc_amd/pipeline/vtc.py comes from.

    python tools/batch_sweep.py --out profiles/r02_batch_sweep.json
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vtc_amd import synth  # noqa: E402
from vtc_amd.engine import VtcEngine  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=float, nargs="+", default=[4.0, 10.0])
    ap.add_argument("--batches", type=int, nargs="+", default=[16, 32, 64, 128, 192, 256, 384, 512])
    ap.add_argument("--seconds", type=float, default=1.5, help="timed region per point")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    w = synth.make_weights(seed=0, family="hubert", calibrated=True)
    dev = torch.device("cuda:0")
    rows = []
    for chunk in args.chunks:
        for batch in args.batches:
            torch.cuda.empty_cache()
            free0 = torch.cuda.mem_get_info(dev)[0]
            eng = VtcEngine(w, n_labels=4, chunk_duration_s=chunk, max_batch=batch, device=dev)
            used = free0 - torch.cuda.mem_get_info(dev)[0]
            T, W, step = eng.frames_per_window, eng.window_samples, eng.step
            n = (batch - 1) * step + W
            arena = torch.from_numpy(synth.synth_audio(min(n / 16000.0, 400.0), seed=3)).to(dev)
            arena = arena.repeat(-(-n // arena.numel()))[:n].contiguous()
            start = np.arange(batch, dtype=np.int64) * step
            valid = np.full(batch, W, np.int32)
            dst = np.arange(batch, dtype=np.int64) * T
            keep = np.full(batch, T, np.int32)
            bufs = [torch.empty((batch * T, 4), device=dev), torch.empty((batch * T, 4), device=dev)]
            rows_ = np.array([0, batch * T], dtype=np.int64)
            lo, hi = np.zeros(4, np.float32), np.full(4, np.inf, np.float32)
            pending = False

            def one(i):
                nonlocal pending
                eng.forward_windows(arena, start, valid, dst, keep, bufs[i % 2])
                if pending:
                    eng.decode_collect()
                eng.decode_launch(bufs[i % 2], rows_, lo, hi, summary=True)
                pending = True

            for i in range(3):
                one(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            k = 0
            while time.perf_counter() - t0 < args.seconds or k < 4:
                one(k)
                k += 1
            eng.decode_collect()
            pending = False
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            hours = k * batch * step / 16000.0 / 3600.0
            rows.append({"chunk_duration_s": chunk, "batch": batch, "audio_hours_per_s": round(hours / dt, 3),
                         "ms_per_step": round(1e3 * dt / k, 3), "engine_device_gb": round(used / 2 ** 30, 2)})
            print(json.dumps(rows[-1]), flush=True)
            eng.close()
            del arena, bufs
    doc = {"what": "hot path (forward + threshold / run-length decode, inputs resident in HBM) on one MI355X, f16 MFMA operands",
           "points": rows}
    if args.out:
        Path(args.out).write_text(json.dumps(doc, indent=1) + "\n")


if __name__ == "__main__":
    main()
