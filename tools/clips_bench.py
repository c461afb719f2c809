#!/usr/bin/env python3
"""Throughput of the packaged-clips caller (vtc_amd.analysis.vtc_on_clips) with and without cross-clip batching.

Builds WebDataset tars of synthetic PCM16 clips (durations uniform in [--min_s, --max_s]) in a scratch directory and runs
the caller twice: clips batched together (default) and one clip per forward (the reference's call pattern).

    python tools/clips_bench.py --clips 2000
"""
from __future__ import annotations

import argparse
import io
import json
import os
import sys
import tarfile
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vtc_amd import synth  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=2000)
    ap.add_argument("--min_s", type=float, default=3.0)
    ap.add_argument("--max_s", type=float, default=30.0)
    ap.add_argument("--chunk", type=float, default=4.0)
    ap.add_argument("--batch", type=int, default=164)
    ap.add_argument("--format", default="wav", choices=["wav", "flac"], help="clip members: wav, or flac as the reference's packer "
                    "writes them (src/packaging/writer.py:36; 24 distinct clips encoded by the test encoder and reused)")
    ap.add_argument("--skip_per_clip", action="store_true")
    args = ap.parse_args()
    from scipy.io import wavfile

    root = Path(tempfile.mkdtemp(prefix="vtc_clips_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None))
    rng = np.random.default_rng(5)
    base = synth.synth_audio(args.max_s * 8, seed=77)
    pcm = np.clip(np.round(base * 32768), -32768, 32767).astype(np.int16)
    total_s = 0.0
    for name in ("batched", "per_clip"):
        (root / "output" / name / "shards").mkdir(parents=True)
    per_tar = 500
    durs = rng.uniform(args.min_s, args.max_s, args.clips)
    flacs = []
    if args.format == "flac":
        from oracle import flac_codec           # a dev tool may use the test encoder to make inputs; the product never does
        for d in np.linspace(args.min_s, args.max_s, 24):
            n = int(d * 16000)
            o = int(rng.integers(0, len(pcm) - n))
            flacs.append((n, flac_codec.encode(pcm[o:o + n].astype(np.int64), 16000, subframe="lpc")))
        print(f"{len(flacs)} distinct FLAC clips encoded", flush=True)
    for t in range(0, args.clips, per_tar):
        path = root / "output" / "batched" / "shards" / f"shards-{t // per_tar:04d}.tar"
        with tarfile.open(path, "w") as tf:
            for c in range(t, min(t + per_tar, args.clips)):
                if flacs:
                    n, data = flacs[int(rng.integers(len(flacs)))]
                else:
                    n = int(durs[c] * 16000)
                    o = int(rng.integers(0, len(pcm) - n))
                    buf = io.BytesIO()
                    wavfile.write(buf, 16000, pcm[o:o + n])
                    data = buf.getvalue()
                info = tarfile.TarInfo(f"rec_{c:06d}.{args.format}")
                info.size = len(data)
                tf.addfile(info, io.BytesIO(data))
                total_s += n / 16000.0
        os.link(path, root / "output" / "per_clip" / "shards" / path.name)
    (root / "manifests").mkdir()
    for name in ("batched", "per_clip"):
        (root / "manifests" / f"{name}.csv").write_text("path\n")
    (root / "cfg.yml").write_text(f"model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: {args.chunk}\n")
    os.chdir(root)
    from vtc_amd.analysis import vtc_on_clips as clips
    import torch
    out = {"clips": args.clips, "audio_hours": total_s / 3600.0, "chunk_duration_s": args.chunk, "batch": args.batch}
    out["format"] = args.format
    for name, per_clip in (("batched", False), ("per_clip", True)):
        if per_clip and args.skip_per_clip:
            continue
        t0 = time.time()
        clips.main(name, config=str(root / "cfg.yml"), checkpoint="synthetic:0", batch_size=args.batch, per_clip=per_clip)
        torch.cuda.synchronize()
        wall = time.time() - t0
        out[name] = {"wall_s": wall, "clips_per_s": args.clips / wall, "audio_hours_per_s": total_s / 3600.0 / wall}
    print(json.dumps(out), flush=True)
    import shutil
    shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
