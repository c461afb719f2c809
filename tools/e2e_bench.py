#!/usr/bin/env python3
"""End-to-end CLI throughput (SURVEY.md section 8d metric as the reference quotes it: WAV files in -> parquet out).

Generates a slice of BASELINE config C2 (N files x 600 s of 16 kHz PCM16, seeds 1000+i) in a scratch directory,
runs `vtc_amd.pipeline.vtc.main` over it in-process with seeded synthetic weights and 10 s windows / batch 64,
and prints one JSON line with PCIe- and file-I/O-inclusive audio-hours/sec.  bench.py's `value` excludes host
I/O by contract; this is the number to quote beside it.

    python tools/e2e_bench.py --files 60 --seconds 600
"""

from __future__ import annotations

import argparse
import json
import os
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
    ap.add_argument("--files", type=int, default=60)
    ap.add_argument("--seconds", type=float, default=600.0)
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic signals (files cycle through them)")
    ap.add_argument("--chunk", type=float, default=10.0)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--prefetch", type=int, default=4)
    ap.add_argument("--format", default="wav", choices=["wav", "flac"], help="container of the generated files (flac: encoded by "
                    "the test encoder of oracle/flac_codec.py, decoded by the library's vtc_flac_decode on the reader threads)")
    ap.add_argument("--tile", type=int, default=1, help="build each signal by repeating a --seconds/--tile long one (fast "
                    "generation of day-long files)")
    ap.add_argument("--no_group", action="store_true", help="one forward per file (A/B against cross-file batching)")
    ap.add_argument("--dir", default=None, help="scratch directory (default: a fresh temp dir, /dev/shm if present)")
    args = ap.parse_args()

    root = Path(args.dir) if args.dir else Path(tempfile.mkdtemp(prefix="vtc_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None))
    (root / "manifests").mkdir(parents=True, exist_ok=True)
    (root / "audio").mkdir(exist_ok=True)
    t0 = time.time()
    sigs = [np.tile(synth.synth_audio(args.seconds / args.tile, seed=1000 + i), args.tile)
            for i in range(min(args.distinct, args.files))]
    print(f"signals ready after {time.time() - t0:.1f} s", flush=True)
    paths = []
    if args.format == "flac":
        from oracle import flac_codec           # a tool may use the test encoder to make inputs; the product never does
        blobs = [flac_codec.encode(np.clip(np.round(s * 32768.0), -32768, 32767).astype(np.int64), 16000) for s in sigs]
    for i in range(args.files):
        p = root / "audio" / f"f{i:05d}.{args.format}"
        if i >= len(sigs):
            os.link(paths[i % len(sigs)], p)          # same bytes as an earlier file: a hard link, no extra space
        elif args.format == "flac":
            p.write_bytes(blobs[i % len(blobs)])
        else:
            write_wav(p, sigs[i % len(sigs)])
        paths.append(p)
        if i % 16 == 15 or args.seconds > 3600:
            print(f"wrote {i + 1}/{args.files} files", flush=True)
    (root / "manifests" / "c2.csv").write_text("path\n" + "\n".join(str(p) for p in paths) + "\n")
    (root / "cfg.yml").write_text(f"model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: {args.chunk}\n")
    gen_s = time.time() - t0

    os.chdir(root)
    from vtc_amd.pipeline import vtc as cli
    import torch

    # warm-up on one file so that library load / weight packing is not in the timed region (a3 cost, once per process)
    (root / "manifests" / "warm.csv").write_text("path\n" + str(paths[0]) + "\n")
    cli.main("warm", config=str(root / "cfg.yml"), checkpoint="synthetic:0", batch_size=args.batch, prefetch=args.prefetch)
    torch.cuda.synchronize()
    t0 = time.time()
    cli.main("c2", config=str(root / "cfg.yml"), checkpoint="synthetic:0", batch_size=args.batch, prefetch=args.prefetch,
             no_group=args.no_group)
    torch.cuda.synchronize()
    wall = time.time() - t0
    hours = args.files * args.seconds / 3600.0
    import pyarrow.parquet as pq
    last = json.loads((root / "logs" / "benchmarks.jsonl").read_text().strip().splitlines()[-1])
    n_raw = pq.read_table(root / "output" / "c2" / "vtc_raw" / "shard_0.parquet").num_rows
    import resource
    peak_rss_gb = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    print(json.dumps({"peak_rss_gb": peak_rss_gb, "metric": "VTC audio-hours/sec end-to-end (WAV -> parquet, one process, one GPU)",
                      "value": hours / wall, "unit": "audio-hours/sec",
                      "steady_state_value": last["total_audio_seconds"] / 3600.0 / last["wall_seconds"],
                      "note": "value includes model construction + weight packing (once per process); steady_state_value is "
                              "the CLI's own file loop + parquet writes", "files": args.files, "file_seconds": args.seconds,
                      "audio_hours": hours, "wall_s": wall, "raw_segments": n_raw, "generate_s": gen_s,
                      "chunk_duration_s": args.chunk, "batch": args.batch, "prefetch": args.prefetch, "no_group": args.no_group, "format": args.format}), flush=True)
    if not args.dir:
        import shutil
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
