#!/usr/bin/env python3
"""Node-level end-to-end throughput: N CLI workers (one per GPU) started by vtc_amd.launcher on one manifest,
files in -> parquet out.  This is the metric SURVEY.md 8(d) defines ("audio-hours per wall-second per node at 1/2/4/8
MI355X, end to end through the CLI"); the reference's form of it is the SLURM array (slurm/vtc.slurm:9-10) whose tasks
each log one record (src/pipeline/vtc.py:403-418).

Workload: a slice of BASELINE config C3 (N x files_per_worker files x 600 s of 16 kHz PCM16; two distinct signals, the
other files are hard links, so /dev/shm holds 38 MB whatever N is), 10 s windows, batch 64, seeded synthetic weights.
This process never touches the GPU: it writes the files, starts the launcher as a child and reads logs/benchmarks.jsonl.

Reported: node rate over the span in which the workers' file loops ran (total audio / (latest loop end - earliest loop
start): honest when the workers do not start together), the per-worker steady-state rates, and the rate over the whole
launcher call (interpreter start-up, model construction and weight packing of every worker included).

    python tools/e2e_node_bench.py --workers 8
    python tools/e2e_node_bench.py --workers 2 --devices 0,0        # rehearsal: two workers on a one-GPU box
"""

from __future__ import annotations

import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=2)
    ap.add_argument("--devices", default="", help="passed to the launcher (e.g. 0,0 for a rehearsal on one GPU)")
    ap.add_argument("--files-per-worker", type=int, default=240)
    ap.add_argument("--seconds", type=float, default=600.0)
    ap.add_argument("--chunk", type=float, default=10.0)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--timeout", type=float, default=600.0)
    ap.add_argument("--launcher-args", default="", help="extra launcher options, space separated (e.g. --no_affinity)")
    args = ap.parse_args()

    import numpy as np  # noqa: F401
    from vtc_amd import synth
    from vtc_amd.audio import write_wav

    n_files = args.workers * args.files_per_worker
    root = Path(tempfile.mkdtemp(prefix="vtc_node_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None))
    try:
        (root / "manifests").mkdir()
        (root / "audio").mkdir()
        paths = []
        for i in range(n_files):
            p = root / "audio" / f"f{i:05d}.wav"
            if i < 2:
                write_wav(p, synth.synth_audio(args.seconds, seed=1000 + i))
            else:
                os.link(paths[i % 2], p)
            paths.append(p)
        (root / "manifests" / "node.csv").write_text("path\n" + "\n".join(str(p) for p in paths) + "\n")
        (root / "cfg.yml").write_text(f"model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: {args.chunk}\n")
        cmd = [sys.executable, "-m", "vtc_amd.launcher", "node", "--gpus", str(args.workers)]
        if args.devices:
            cmd += ["--devices", args.devices]
        cmd += args.launcher_args.split()
        cmd += ["--", "--config", str(root / "cfg.yml"), "--checkpoint", "synthetic:0", "--batch_size", str(args.batch)]
        env = dict(os.environ, PYTHONPATH=str(ROOT) + os.pathsep + os.environ.get("PYTHONPATH", ""))
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE",
                  "TORCHELASTIC_RUN_ID"):
            env.pop(k, None)        # the workers are plain CLI processes, not ranks of the bench's process group
        t0 = time.time()
        c0 = os.times()
        r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=args.timeout)
        wall = time.time() - t0
        c1 = os.times()
        cpu_s = (c1.children_user - c0.children_user) + (c1.children_system - c0.children_system)   # launcher + all workers
        if r.returncode != 0:
            raise RuntimeError(f"launcher exited with {r.returncode}: {(r.stdout + r.stderr)[-1500:]}")
        recs = [json.loads(l) for l in (root / "logs" / "benchmarks.jsonl").read_text().strip().splitlines()]
        assert len(recs) == args.workers, f"{len(recs)} benchmark records for {args.workers} workers"
        import pyarrow.parquet as pq
        n_meta = sum(pq.read_metadata(f).num_rows for f in (root / "output" / "node" / "vtc_meta").glob("shard_*.parquet"))
        n_raw = sum(pq.read_metadata(f).num_rows for f in (root / "output" / "node" / "vtc_raw").glob("shard_*.parquet"))
        assert n_meta == n_files, f"{n_meta} meta rows for {n_files} files"
        hours = sum(x["total_audio_seconds"] for x in recs) / 3600.0
        span = max(x["extra"]["loop_end_unix"] for x in recs) - min(x["extra"]["loop_start_unix"] for x in recs)
        print(json.dumps({
            "metric": "VTC audio-hours/sec (node), end to end through the CLI: WAV files -> vtc_raw/vtc_merged/vtc_meta parquet",
            "value": hours / span, "unit": "audio-hours/sec", "workers": args.workers, "devices": args.devices or "0..N-1", "launcher_args": args.launcher_args,
            "files": n_files, "file_seconds": args.seconds, "audio_hours": hours, "file_loop_span_s": span,
            "per_worker_steady_state": [round(x["total_audio_seconds"] / 3600.0 / x["wall_seconds"], 3) for x in recs],
            "per_worker_file_loop_s": [x["wall_seconds"] for x in recs],
            # where a worker's file loop went: seconds its GPU thread waited for the GPU (collect), for its reader threads
            # (gpu_wait), and the final parquet write; a host-bound node shows up as gpu_wait_for_readers growing with N
            "per_worker_gpu_busy_wait_s": [round(x["extra"].get("stage_seconds", {}).get("gpu_collect", 0.0), 2) for x in recs],
            "per_worker_wait_for_readers_s": [round(x["extra"].get("stage_seconds", {}).get("gpu_wait", 0.0), 2) for x in recs],
            "per_worker_write_s": [round(x["extra"].get("stage_seconds", {}).get("write", 0.0), 2) for x in recs],
            "with_startup_and_model_construction": hours / wall, "launcher_wall_s": wall, "raw_segments": n_raw,
            "chunk_duration_s": args.chunk, "batch": args.batch, "host_cpus": len(os.sched_getaffinity(0)),
            # measured host budget: CPU time (user + system) of the launcher and all its workers, start-up and model
            # construction included; host_cores_busy is what the run kept busy on average
            "cpu_seconds": cpu_s, "cpu_seconds_per_audio_hour": cpu_s / hours, "host_cores_busy": cpu_s / wall,
            "what": "vtc_amd.launcher: one worker process per GPU, file-level sharding (--array_id/--array_count), no collective; "
                    "file I/O from /dev/shm, H2D, GPU front-end, forward, decode, segment rows, collar merge, parquet"}), flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
