#!/usr/bin/env python3
"""GPU idle time between kernels from a rocprofv3 --kernel-trace CSV (second half of the trace = steady state):
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --no-profile ... ; python3 tools/trace_gaps.py OUT"""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# take the last 40 % of the trace (steady state)
rows = rows[int(len(rows) * 0.5):]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
print("kernels", len(rows), "span_us", span / 1e3, "busy_us", busy / 1e3, "busy_frac", busy / span, "mean_gap_us", sum(gaps) / len(gaps) / 1e3)
