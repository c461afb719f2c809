#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes with SQ / GRBM counters into one JSON: per kernel (all launches of a pass averaged) every
counter per launch, plus derived figures that can be set beside DESIGN.md's in-loop estimates:
  mfma_busy_frac      = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES * 4)   (cycles a SIMD's matrix pipe was busy / SIMD-cycles
                        of busy CUs; both are summed over the chip)
  wait_frac           = SQ_WAIT_ANY / SQ_WAVE_CYCLES                         (wave-cycles parked on s_waitcnt / barriers)
  lds_conflict_frac   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  eff_clock_ghz       = GRBM_GUI_ACTIVE / 8 / average kernel time            (MI355X_MICROARCH.md, DVFS give-back); only for
                        kernels of 50 us and more: GRBM_GUI_ACTIVE also counts the launch's set-up and drain, which for a
                        short kernel is longer than the kernel (the figure came out at 4-9 "GHz" for them)
usage: summarize_sq.py KERNEL_STATS.csv OUT.json PASS1.csv [PASS2.csv ...]"""
import collections, csv, json, sys

stats = {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(sys.argv[1]))}
agg = collections.OrderedDict()
dropped = collections.Counter()
TIME_LIKE = ("SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")
for path in sys.argv[3:]:
    per_pass = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        per_pass.setdefault(r["Kernel_Name"], collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in per_pass.items():
        # a kernel's first launch in a process can carry one-off set-up (code object load, scratch allocation): launches whose
        # duration-like counter is more than 8 x the median of that kernel's launches are left out of every counter of the pass
        bad = set()
        for n in TIME_LIKE:
            v = d.get(n)
            if v and len(v) >= 4:
                med = sorted(v)[len(v) // 2]
                bad |= {i for i, x in enumerate(v) if med > 0 and x > 8.0 * med}
        dropped[k] += len(bad)
        tgt = agg.setdefault(k, collections.defaultdict(list))
        for n, v in d.items():
            tgt[n] += [x for i, x in enumerate(v) if i not in bad]
out = []
for k, d in agg.items():
    c = {n: sum(v) / len(v) for n, v in d.items()}
    row = {"kernel": k, "launches_profiled": max(len(v) for v in d.values()), "outlier_launches_left_out": dropped[k],
           "avg_ns": stats.get(k), "per_launch": c}
    g = c.get
    if g("SQ_BUSY_CU_CYCLES"):
        row["mfma_busy_frac"] = g("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * c["SQ_BUSY_CU_CYCLES"])
    if g("SQ_WAVE_CYCLES"):
        row["wait_frac"] = g("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        row["issue_stall_frac"] = g("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
    if g("SQ_LDS_IDX_ACTIVE"):
        row["lds_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    if g("GRBM_GUI_ACTIVE") and stats.get(k) and stats[k] >= 50_000.0:
        row["eff_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / stats[k]
    out.append(row)
out.sort(key=lambda r: -(r["avg_ns"] or 0) * r["launches_profiled"])
json.dump({"note": "rocprofv3 --pmc, separate passes over bench.py --steps 2 --warmup 1 (counter values are chip-wide sums per launch, "
                   "averaged over the launches of the pass); avg_ns from the --kernel-trace --stats pass", "kernels": out},
          open(sys.argv[2], "w"), indent=1)
print("wrote", sys.argv[2], len(out), "kernels")
