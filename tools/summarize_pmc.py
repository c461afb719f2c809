#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of `bench.py --no-profile`) into
profiles/<tag>_hbm_traffic.json: HBM bytes per launch per kernel.  FETCH_SIZE is doubled (gfx950 reports half the
bytes of wide coalesced reads, MI355X_MICROARCH.md section HBM); both counters are in KiB.
usage: summarize_pmc.py FETCH.csv WRITE.csv OUT.json"""
import collections, csv, json, sys

def load(path, name):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        agg.setdefault((r["Kernel_Name"], r["Grid_Size"]), []).append(float(r["Counter_Value"]))
    return agg

f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = []
for k, v in f.items():
    wv = w.get(k, [0.0])
    out.append({"kernel": k[0], "grid": int(k[1]), "launches_profiled": len(v),
                "fetch_bytes_per_launch": 2.0 * 1024 * sum(v) / len(v),
                "write_bytes_per_launch": 1024 * sum(wv) / len(wv)})
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over bench.py --steps 2; FETCH_SIZE x2 (gfx950)",
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(out), "kernels")
