#!/usr/bin/env python3
"""Single-node multi-GPU launcher for the VTC step: one worker process per GPU, file-level sharding.

Replaces the reference's SLURM job array (slurm/vtc.slurm:10 `--array=0-2`, slurm/pipeline.sh:88-93): worker
i sees GPU i only and runs `python -m vtc_amd.pipeline.vtc DATASET --array_id i --array_count N`, so outputs
stay `shard_{i}.parquet` and the resume rule (union over all shards) is unchanged.  There is no collective:
workers are independent and the launcher only waits for them.

    python -m vtc_amd.launcher chunks30 --gpus 8 -- --checkpoint VTC-2.0/model/best.ckpt --balance

Host budget.  The workers share the node's cores, and what a worker needs depends on the input container
(tools/host_budget.py, profiles/r02_host_budget.json: a reader thread decodes ~34 audio-h/s of 16 kHz PCM16 WAV but
~1.3-1.5 audio-h/s of FLAC, so one GPU at 15 audio-h/s keeps <1 reader busy on WAV and 10-12 on FLAC).  The launcher
therefore gives worker i the i-th equal slice of the CPUs this process may run on (contiguous: on the usual two-socket node
GPUs 0-3 hang off the first socket's cores, 4-7 off the second's), pins it there (VTC_WORKER_CPUS, applied by the worker with
sched_setaffinity before it starts its threads), and sizes its pools from the slice: reader threads (--prefetch) =
slice - 2 (one GPU thread, one main thread), at least 2, at most 16; OMP / torch intra-op threads = 1 per worker (the host
side is numpy / arrow column work, not BLAS); FLAC long-stream decode threads = min(4, slice / 2).  --no_affinity
leaves placement to the OS; an explicit --prefetch after "--" wins over the computed one.
"""

from __future__ import annotations

import argparse
import os
import subprocess
import sys


def worker_cpu_slices(n_workers: int, per_worker: int = 0, cpus=None) -> list:
    """Equal contiguous slices of the CPUs this process may use, one per worker (every worker gets at least one CPU;
    with fewer CPUs than workers the slices wrap around and overlap)."""
    cpus = sorted(os.sched_getaffinity(0)) if cpus is None else sorted(cpus)
    k = per_worker if per_worker > 0 else max(1, len(cpus) // n_workers)
    return [[cpus[(i * k + j) % len(cpus)] for j in range(min(k, len(cpus)))] for i in range(n_workers)]


def thread_budget(n_cpus: int) -> dict:
    """Pools of one worker that owns n_cpus cores: see the module docstring."""
    return {"readers": max(2, min(16, n_cpus - 2)), "intra_op": 1, "flac_threads": max(1, min(4, n_cpus // 2))}


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("dataset")
    ap.add_argument("--gpus", type=int, default=0, help="workers to start (0 = every visible GPU)")
    ap.add_argument("--devices", default="", help="comma-separated device ids, one per worker (default 0..gpus-1); "
                                                  "repeating an id puts several workers on one GPU (rehearsals on a small box)")
    ap.add_argument("--per_gpu", type=int, default=1,
                    help="workers per GPU (default 1).  Two workers sharing a GPU fill each other's partial batches and tile "
                         "rounds: +3 %% on uniform 10-minute files, +12 %% on a skewed manifest (DESIGN.md section 8); each "
                         "worker holds its own engine (4-12 GB) and is its own shard (array_count = gpus x per_gpu)")
    ap.add_argument("--cpus_per_worker", type=int, default=0, help="cores per worker (0 = equal share of this process's CPUs)")
    ap.add_argument("--no_affinity", action="store_true", help="do not pin the workers to CPU slices")
    ap.add_argument("--keep_thread_env", action="store_true",
                    help="keep OMP_NUM_THREADS / MKL_NUM_THREADS of the calling environment (default: 1 per worker)")
    argv = list(sys.argv[1:] if argv is None else argv)
    rest = []
    if "--" in argv:                     # everything after the first "--" goes to every worker, untouched
        cut = argv.index("--")
        argv, rest = argv[:cut], argv[cut + 1:]
    args = ap.parse_args(argv)
    devices = [d.strip() for d in args.devices.split(",") if d.strip()]
    n = args.gpus or len(devices)
    if n <= 0:
        import torch
        n = torch.cuda.device_count()   # counting devices does not initialise the GPU
    if n < 1:
        print("no GPU visible", file=sys.stderr)
        return 2
    if devices and len(devices) != n:
        print(f"--devices names {len(devices)} devices for {n} workers", file=sys.stderr)
        return 2
    if not devices:
        devices = [str(i) for i in range(n)]
    if args.per_gpu > 1:                 # worker j of GPU i is shard i * per_gpu + j
        devices = [d for d in devices for _ in range(args.per_gpu)]
        n = len(devices)
    # device ids are positions in what THIS process sees: if the caller already restricted the visible GPUs, map through its list
    outer = [d.strip() for d in os.environ.get("HIP_VISIBLE_DEVICES", "").split(",") if d.strip()]
    if outer:
        try:
            devices = [outer[int(d)] for d in devices]
        except (ValueError, IndexError):
            print(f"--devices {devices} does not fit HIP_VISIBLE_DEVICES={','.join(outer)}", file=sys.stderr)
            return 2
    procs = []
    slices = worker_cpu_slices(n, args.cpus_per_worker)
    for i in range(n):
        env = dict(os.environ)
        env["HIP_VISIBLE_DEVICES"] = devices[i]
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        extra = []
        budget = thread_budget(len(slices[i]))
        if not args.no_affinity:
            env["VTC_WORKER_CPUS"] = ",".join(str(c) for c in slices[i])
        if not args.keep_thread_env:       # N workers x (all cores) OpenMP threads is never what the caller's setting meant
            env["OMP_NUM_THREADS"] = env["MKL_NUM_THREADS"] = str(budget["intra_op"])
        else:
            env.setdefault("OMP_NUM_THREADS", str(budget["intra_op"]))
            env.setdefault("MKL_NUM_THREADS", str(budget["intra_op"]))
        env.setdefault("VTC_FLAC_THREADS", str(budget["flac_threads"]))
        if not any(a == "--prefetch" or a.startswith("--prefetch=") for a in rest):
            extra = ["--prefetch", str(budget["readers"])]
        cmd = ([sys.executable, "-m", "vtc_amd.pipeline.vtc", args.dataset, "--array_id", str(i), "--array_count", str(n)]
               + extra + rest)
        procs.append(subprocess.Popen(cmd, env=env))
    rc = 0
    for i, p in enumerate(procs):
        code = p.wait()
        if code != 0:
            print(f"worker {i} exited with {code}", file=sys.stderr)
            rc = rc or code
    return rc


if __name__ == "__main__":
    sys.exit(main())
