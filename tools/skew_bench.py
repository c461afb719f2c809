#!/usr/bin/env python3
"""Short-file throughput of the CLI (the 9,990 files of 1-30 s that make up most of BASELINE configs[4]): N PCM16 files
with log-uniform durations in /dev/shm -> parquet, the CLI's stage times, and the ideal the GPU alone would allow
(windows x ms per forward).  Diagnostic for the per-file host cost.

    python tools/skew_bench.py --files 10000 --max-seconds 30 --prefetch 4
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vtc_amd import synth  # noqa: E402
from vtc_amd.audio import write_wav  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=10000)
    ap.add_argument("--min-seconds", type=float, default=1.0)
    ap.add_argument("--max-seconds", type=float, default=30.0)
    ap.add_argument("--chunk", type=float, default=4.0)
    ap.add_argument("--batch", type=int, default=164)
    ap.add_argument("--prefetch", type=int, default=4)
    ap.add_argument("--no_group", action="store_true")
    args = ap.parse_args()
    root = Path(tempfile.mkdtemp(prefix="vtc_skew_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None))
    try:
        (root / "manifests").mkdir()
        (root / "audio").mkdir()
        rng = np.random.default_rng(5)
        base = synth.synth_audio(97.3, seed=5)
        dur = np.exp(rng.uniform(np.log(args.min_seconds), np.log(args.max_seconds), args.files))
        paths = []
        for i, d in enumerate(dur):
            p = root / "audio" / f"s{i:05d}.wav"
            write_wav(p, np.resize(np.roll(base, -int(rng.integers(len(base)))), int(round(d * 16000))))
            paths.append(p)
        (root / "manifests" / "skew.csv").write_text("path\n" + "\n".join(str(p) for p in paths) + "\n")
        (root / "manifests" / "warm.csv").write_text("path\n" + "\n".join(str(p) for p in paths[:64]) + "\n")
        (root / "cfg.yml").write_text(f"model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: {args.chunk}\n")
        os.chdir(root)
        import torch
        from vtc_amd.pipeline import vtc as cli
        cli.main("warm", config=str(root / "cfg.yml"), checkpoint="synthetic:0", batch_size=args.batch, prefetch=args.prefetch)
        torch.cuda.synchronize()
        t0 = time.time()
        cli.main("skew", config=str(root / "cfg.yml"), checkpoint="synthetic:0", batch_size=args.batch, prefetch=args.prefetch,
                 no_group=args.no_group)
        wall = time.time() - t0
        last = json.loads((root / "logs" / "benchmarks.jsonl").read_text().strip().splitlines()[-1])
        step = 199 * 320 if args.chunk == 4.0 else None
        windows = int(sum(-(-(int(round(d * 16000)) - 399) // step) for d in dur)) if step else None
        print(json.dumps({"files": args.files, "audio_hours": float(dur.sum() / 3600), "windows": windows,
                          "forwards_if_full": windows / args.batch if windows else None, "wall_s": wall,
                          "file_loop_s": last["wall_seconds"], "files_per_s": args.files / last["wall_seconds"],
                          "audio_hours_per_s": last["total_audio_seconds"] / 3600 / last["wall_seconds"],
                          "stage_seconds": last["extra"]["stage_seconds"], "prefetch": args.prefetch}), flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
