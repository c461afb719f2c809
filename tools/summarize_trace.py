#!/usr/bin/env python3
"""Per-kernel durations of the TIMED steps of a rocprofv3 --kernel-trace run of bench.py: the launches of the warm-up steps
are left out (a kernel's first launch in a process carries one-off set-up -- code object load, scratch -- of up to several
milliseconds, which rocprofv3's own kernel_stats.csv averages in: in the round-3 run one 7.1 ms first launch adds 27 us to
the average of 264 launches of the LayerNorm-residual GEMM).  Also splits the GEMM rows by grid size where one kernel serves
several shapes (out-proj and FFN2 share a template instance).
usage: summarize_trace.py KERNEL_TRACE.csv OUT.csv WARMUP_STEPS TIMED_STEPS"""
import collections
import csv
import sys

trace, out, warm, timed = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
per = collections.OrderedDict()
for r in csv.DictReader(open(trace)):
    key = (r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    per.setdefault(key, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows = []
for (name, gx, gy, gz), v in per.items():
    v.sort()
    if len(v) % (warm + timed):
        keep = v[len(v) // 2:]                       # not a per-step kernel (engine set-up copies): second half of its launches
    else:
        keep = v[len(v) // (warm + timed) * warm:]   # drop the launches of the warm-up steps
    d = [x[1] for x in keep]
    rows.append({"Name": name, "Grid": f"{gx}x{gy}x{gz}", "Calls": len(d), "CallsPerStep": round(len(d) / timed, 2),
                 "TotalDurationNs": sum(d), "AverageNs": sum(d) / len(d), "MinNs": min(d), "MaxNs": max(d),
                 "LaunchesLeftOut": len(v) - len(d)})
rows.sort(key=lambda r: -r["TotalDurationNs"])
with open(out, "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0]))
    w.writeheader()
    w.writerows(rows)
print("wrote", out, len(rows), "rows; per timed step:", round(sum(r["TotalDurationNs"] for r in rows) / timed / 1e6, 3), "ms of kernels")
