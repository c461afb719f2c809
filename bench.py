#!/usr/bin/env python3
"""VTC hot-path benchmark (BASELINE.json metric: audio-hours/sec).

A *step* is one pass of the hot path over one batch: `--batch` windows of `--chunk` seconds of
synthetic 16 kHz audio (already resident in HBM) -> BabyHuBERT forward -> stitched logits ->
threshold/run-length decode to (start, end, label) triples on the host.  Workload at N=1 is
BASELINE.json configs[1]: fixed 10 s windows, batch 64, 16-bit MFMA transformer.

N > 1: one process per GPU, files/windows are sharded, no collective in the data path (the
reference runs one process per GPU as a SLURM array, slurm/vtc.slurm:9-10); the only communication is the timing
barrier + max-over-ranks.  The ranks come either from an external launcher (torch.distributed.run sets WORLD_SIZE /
RANK / LOCAL_RANK) or, when WORLD_SIZE is unset and --gpus N > 1, from this script itself: the parent starts N fresh
child processes before it touches the GPU, waits for them and exits with their status; it never initialises HIP.
`n_gpus` in the JSON line is the size of the initialised process group, not the flag.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from vtc_amd import synth  # noqa: E402
from vtc_amd.engine import VtcEngine  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0   # dense 16-bit MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def algorithmic_gflop_per_window(T: int, L: list[int]) -> float:
    """BASELINE.md section 2 formulae (head excluded)."""
    kern = (10, 3, 3, 3, 3, 2, 2)
    conv = L[0] * 512 * 1 * 10 + sum(L[i] * 512 * 512 * kern[i] for i in range(1, 7))
    proj = T * 512 * 768
    pos = T * 768 * 48 * 128
    lin = 12 * T * (4 * 768 * 768 + 2 * 768 * 3072)
    att = 12 * 2 * T * T * 768
    return 2.0 * (conv + proj + pos + lin + att) / 1e9


def conv_lengths(w: int) -> list[int]:
    out = []
    for k, s in zip((10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)):
        w = (w - k) // s + 1
        out.append(w)
    return out


def measured_traffic(*kernel_substrs: str):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_hbm_traffic.json), launch-weighted
    over the kernels whose name contains one of kernel_substrs; None when no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_hbm_traffic.json")))
    if not files:
        return None
    ks = [k for k in json.load(open(files[-1]))["kernels"] if any(sub in k["kernel"] for sub in kernel_substrs)]
    n = sum(k["launches_profiled"] for k in ks)
    if not n:
        return None
    return sum((k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]) * k["launches_profiled"] for k in ks) / n


def cpu_baseline(chunk_s: float, n_windows: int, batch: int) -> dict:
    """The fp32 CPU oracle (oracle/vtc_oracle.py) timed on this host's cores on a bounded sample."""
    from oracle import vtc_oracle as O
    threads = torch.get_num_threads()
    w = O.make_weights(seed=0, family="hubert", calibrated=True)
    W = int(round(chunk_s * 16000))
    T = O.conv_out_len(W)
    audio = synth.synth_audio((n_windows * T * 320 + W) / 16000.0, seed=1000)
    t0 = time.perf_counter()
    logits = O.apply_model_on_waveform(w, audio, chunk_s, batch_size=batch)
    mask = O.apply_thresholds(logits, {l: {"lower_bound": 0.5, "upper_bound": 1.0} for l in O.DEFAULT_LABELS})
    O.create_intervals(mask)
    dt = time.perf_counter() - t0
    hours = logits.shape[0] * 320 / 16000.0 / 3600.0
    return {"value": hours / dt, "unit": "audio-hours/sec", "cores": threads, "kind": "port",
            "sample": f"{logits.shape[0] // T} windows x {chunk_s:g} s, batch {batch}, fp32 PyTorch-CPU oracle, {dt:.1f} s"}


def end_to_end_cli(n_files: int, chunk_s: float, batch: int) -> dict:
    """SURVEY 8(d) quotes the metric end to end through the CLI (files in, parquet out); `value` above is the hot path
    with inputs resident, as the bench contract asks.  This runs tools/e2e_bench.py as a child process on a bounded
    sample (n_files x 600 s PCM16 WAV in /dev/shm) and reports the CLI's own file-loop rate beside it.  Never fatal."""
    import subprocess
    import sys
    try:
        tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "e2e_bench.py")
        r = subprocess.run([sys.executable, tool, "--files", str(n_files), "--seconds", "600", "--distinct", "2",
                            "--chunk", str(chunk_s), "--batch", str(batch)], capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        return {"value": d["steady_state_value"], "unit": "audio-hours/sec", "files": d["files"], "file_seconds": d["file_seconds"],
                "audio_hours": d["audio_hours"], "with_model_construction": d["value"],
                "what": "vtc_amd.pipeline.vtc CLI in one process: WAV files in /dev/shm -> vtc_raw/vtc_merged/vtc_meta parquet "
                        "(file I/O, H2D, GPU front-end, forward, decode, segment rows, merge, parquet)"}
    except Exception as e:   # the headline number must not depend on scratch space or a child process
        return {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}


def end_to_end_node(n_workers: int, files_per_worker: int, chunk_s: float, batch: int, devices: str = "") -> dict:
    """The N > 1 form of end_to_end_cli: tools/e2e_node_bench.py starts vtc_amd.launcher with one CLI worker per GPU on a
    bounded manifest (n_workers x files_per_worker files of 600 s in /dev/shm) and reports the node's file-loop rate.
    The resident-input `value` scales linearly by construction (independent ranks); this leg is where host decode, PCIe
    and parquet of N workers share one host.  Runs as a child of rank 0 after every rank has closed its engine.  Never fatal."""
    import subprocess
    try:
        tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "e2e_node_bench.py")
        cmd = [sys.executable, tool, "--workers", str(n_workers), "--files-per-worker", str(files_per_worker), "--chunk", str(chunk_s),
               "--batch", str(batch)]
        if devices:
            cmd += ["--devices", devices]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"value": None, "error": (r.stdout + r.stderr)[-400:]}
        d = json.loads(lines[-1])
        keep = ("value", "unit", "workers", "files", "file_seconds", "audio_hours", "file_loop_span_s", "per_worker_steady_state",
                "per_worker_gpu_busy_wait_s", "per_worker_wait_for_readers_s", "per_worker_write_s",
                "with_startup_and_model_construction", "launcher_wall_s", "host_cpus", "cpu_seconds_per_audio_hour",
                "host_cores_busy", "what")
        return {k: d[k] for k in keep}
    except Exception as e:
        return {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}


def spawn_ranks(n: int, argv: list[str]) -> int:
    """Start n rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), forward
    their output, and return the worst exit status.  Runs before any GPU call: the parent only waits."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VTC_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:      # one rank failed: the others would wait at the barrier forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def spawn_check() -> None:
    """--spawn-check: the N > 1 plumbing alone (rendezvous, barrier, max over ranks, the rank-0 line) with no GPU work;
    tests/test_host.py runs it over gloo with two ranks."""
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n, worst = dist.get_world_size(), float(t.item())
        dist.barrier()
        dist.destroy_process_group()
    else:
        n, worst = 1, 1.0
    if rank == 0:
        print(json.dumps({"spawn_check": True, "n_gpus": n, "max_over_ranks": worst,
                          "spawned_by_bench": os.environ.get("VTC_BENCH_SPAWNED") == "1"}), flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--chunk", type=float, default=10.0)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-windows", type=int, default=120, help="windows of the bounded CPU-oracle sample (~15 s on 16 cores)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the bounded end-to-end CLI run (files -> parquet) at N=1")
    ap.add_argument("--e2e-files", type=int, default=240, help="600 s PCM16 files of the end-to-end sample (40 audio-hours, a 2.5 s file loop)")
    ap.add_argument("--e2e-files-per-worker", type=int, default=480,
                    help="N > 1: 600 s files per CLI worker of the node-level end-to-end leg (80 audio-hours per worker)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event timing")
    ap.add_argument("--profile-every", type=int, default=10,
                    help="steps of the timed region whose launches are bracketed by HIP events: every n-th (two event "
                         "records per launch cost ~7 us of stream time, ~0.75 ms per fully profiled step)")
    ap.add_argument("--spawn-check", action="store_true", help="only exercise the N > 1 launch plumbing (no GPU work)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))     # the parent never touches the GPU
    if args.spawn_check:
        spawn_check()
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal on a box with fewer GPUs than ranks: VTC_BENCH_REHEARSE=1 puts every rank on the visible GPUs round-robin
    # and uses gloo for the timing barrier / max (RCCL refuses two ranks on one device).  Never set by the driver.
    rehearse = os.environ.get("VTC_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank %= max(torch.cuda.device_count(), 1)
    # VTC_BENCH_FORCE_DIST=1 (under torchrun with one rank) runs the RCCL init / barrier / max path on a one-GPU box
    use_dist = world > 1 or os.environ.get("VTC_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        world = dist.get_world_size()      # what the initialised group reports, not the flag
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    weights = synth.make_weights(seed=0, family="hubert", calibrated=True)   # HuBERT-base init, logit std 2
    eng = VtcEngine(weights, n_labels=4, chunk_duration_s=args.chunk, max_batch=args.batch, mfma_dtype=args.dtype,
                    device=dev)
    T, W, step = eng.frames_per_window, eng.window_samples, eng.step
    L = conv_lengths(W)

    # one arena of contiguous synthetic audio per rank (file-level sharding: each rank has its own files)
    n_arena = 2
    arena_len = (args.batch - 1) * step + W
    arenas = [torch.from_numpy(synth.synth_audio(arena_len / 16000.0, seed=1000 + rank * 16 + i)[:arena_len]).to(dev)
              for i in range(n_arena)]
    start = np.arange(args.batch, dtype=np.int64) * step
    valid = np.full(args.batch, W, dtype=np.int32)
    dst = np.arange(args.batch, dtype=np.int64) * T
    keep = np.full(args.batch, T, dtype=np.int32)
    logits = torch.empty((args.batch * T, 4), dtype=torch.float32, device=dev)
    lo = np.zeros(4, dtype=np.float32)            # sigmoid threshold 0.5 <=> logit 0
    hi = np.full(4, np.inf, dtype=np.float32)

    # A step = forward of one batch + threshold / run-length decode of its logits, in the order the CLI's GPU thread uses
    # (vtc_amd/pipeline/vtc.py): forward(i), collect(i-1), decode_launch(i).  Every step's runs reach the host inside the
    # timed region (the last collect comes before the closing synchronize); two logits buffers alternate because the
    # decode of step i-1 is still reading its buffer while forward(i) writes.
    logits2 = [logits, torch.empty_like(logits)]
    rows = np.array([0, args.batch * T], dtype=np.int64)
    in_flight = [False]

    def collect() -> int:
        if not in_flight[0]:
            return 0
        in_flight[0] = False
        return len(eng.decode_collect()[1])

    def one_step(i: int) -> int:
        buf = logits2[i % 2]
        eng.forward_windows(arenas[i % n_arena], start, valid, dst, keep, buf)
        n = collect()
        eng.decode_launch(buf, rows, lo, hi, summary=True)
        in_flight[0] = True
        return n

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()

    for i in range(args.warmup):
        one_step(i)
    collect()
    barrier()
    every = max(1, args.profile_every)
    sampled = 0
    eng.profile_read()   # drain anything recorded before the timed region
    t0 = time.perf_counter()
    nseg = 0
    for i in range(args.steps):
        sample = not args.no_profile and i % every == 0
        if sample:
            eng.profile_enable(True)
            sampled += 1
        nseg += one_step(i)
        if sample:
            eng.profile_enable(False)
    nseg += collect()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    prof = eng.profile_read() if not args.no_profile else None
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()

    if use_dist and world > 1 and not args.no_e2e:
        # the node-level CLI leg below starts one fresh worker process per GPU: give the GPUs back first, on every rank
        eng.close()
        del arenas, logits, logits2
        torch.cuda.empty_cache()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()      # together, before rank 0 goes on alone (CPU baseline / end-to-end legs)
    if rank == 0:
        hours_per_step = args.batch * step / 16000.0 / 3600.0
        value = world * args.steps * hours_per_step / elapsed
        gflop_win = algorithmic_gflop_per_window(T, L)
        out = {
            "metric": "VTC audio-hours/sec (node)", "value": value, "unit": "audio-hours/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic (seeded harmonic bursts + noise; seeded HuBERT-base-initialised weights, head calibrated to logit std 2)",
            "config": {"workload": f"BASELINE configs[1]: fixed {args.chunk:g} s windows ({W} samples -> {T} frames), "
                                   f"batch {args.batch}, forward + threshold/run-length decode, inputs resident in HBM",
                       "batch_windows": args.batch, "chunk_duration_s": args.chunk, "frames_per_window": T,
                       "parallelism": f"file-sharded x{world}, no collective"},
            "algorithmic_gflop_per_window": gflop_win,
            "end_to_end_tflops": world * args.steps * args.batch * gflop_win / 1e3 / elapsed,
            "segments_per_step": nseg / max(args.steps, 1),
        }
        if prof is not None:
            g = prof["gemm_mfma"]
            ach = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
            out["roofline"] = {"bound": "mfma", "kernel": "gemm256_kernel / gemm_kernel / posconv_kernel (all dense contractions: conv1-6 "
                               "implicit GEMM, projection, pos-conv, QKV, out-proj, FFN)", "achieved": ach, "peak": MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS, "traffic": measured_traffic("gemm", "posconv_kernel"),
                               "launches_per_step": g["launches"] / sampled, "sampled_steps": sampled,
                               "avg_launch_ms": g["ms"] / max(g["launches"], 1)}
            c = prof["conv0_apply"]
            if c["ms"] > 0:
                gbs = c["bytes"] / (c["ms"] * 1e-3) / 1e9
                out["roofline_hbm"] = {"bound": "hbm", "kernel": "conv0_mfma_kernel (conv0 + GroupNorm + GELU -> 16-bit channel-last)",
                                       "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                       "traffic": measured_traffic("conv0_mfma", "conv0_apply"), "avg_launch_ms": c["ms"] / max(c["launches"], 1)}
            out["kernel_ms_per_step"] = {k: v["ms"] / sampled for k, v in prof.items() if v["launches"]}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.chunk, args.cpu_windows, 8)
        if not args.no_e2e and world == 1:
            eng.close()
            out["end_to_end_cli"] = end_to_end_cli(args.e2e_files, args.chunk, args.batch)
        elif not args.no_e2e:
            ndev = max(torch.cuda.device_count(), 1)
            devs = ",".join(str(i % ndev) for i in range(world)) if rehearse else ""
            out["end_to_end_cli"] = end_to_end_node(world, args.e2e_files_per_worker, args.chunk, args.batch, devs)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
