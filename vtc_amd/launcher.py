#!/usr/bin/env python3
"""Single-node multi-GPU launcher for the VTC step: one worker process per GPU, file-level sharding.

Replaces the reference's SLURM job array (slurm/vtc.slurm:10 `--array=0-2`, slurm/pipeline.sh:88-93): worker
i sees GPU i only and runs `python -m vtc_amd.pipeline.vtc DATASET --array_id i --array_count N`, so outputs
stay `shard_{i}.parquet` and the resume rule (union over all shards) is unchanged.  There is no collective:
workers are independent and the launcher only waits for them.

    python -m vtc_amd.launcher chunks30 --gpus 8 -- --checkpoint VTC-2.0/model/best.ckpt --balance
"""

from __future__ import annotations

import argparse
import os
import subprocess
import sys


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("dataset")
    ap.add_argument("--gpus", type=int, default=0, help="workers to start (0 = every visible GPU)")
    ap.add_argument("--devices", default="", help="comma-separated device ids, one per worker (default 0..gpus-1); "
                                                  "repeating an id puts several workers on one GPU (rehearsals on a small box)")
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
    procs = []
    for i in range(n):
        env = dict(os.environ)
        env["HIP_VISIBLE_DEVICES"] = devices[i]
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "vtc_amd.pipeline.vtc", args.dataset, "--array_id", str(i), "--array_count", str(n)] + rest
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
