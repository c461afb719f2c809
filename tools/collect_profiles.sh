#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r02
# Passes (each its own run, the program directly after "--"; counters never together with a trace domain):
#   1. --kernel-trace --stats           -> <tag>_bench_kernel_stats.csv (rocprofv3's own summary, warm-up step included) and
#                                          <tag>_bench_kernel_steady.csv (tools/summarize_trace.py: the 10 timed steps only)
#   2. --pmc FETCH_SIZE                 \
#   3. --pmc WRITE_SIZE                 -> <tag>_hbm_traffic.json (tools/summarize_pmc.py, FETCH_SIZE doubled on gfx950)
#   4. --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
#   5. --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE
#                                       -> <tag>_sq_counters.json (tools/summarize_sq.py)
# Everything lands in gpurun_out/profiles_<tag>/; copy what is to be judged into profiles/.
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-e2e --no-profile"
run() { name=$1; shift; echo "== $name"; timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- $BENCH --steps ${STEPS:-2} --warmup 1 > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }; }
STEPS=10 run stats --kernel-trace --stats || exit 1
run fetch --pmc FETCH_SIZE || exit 1
run write --pmc WRITE_SIZE || exit 1
run sq1 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU || exit 1
run sq2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE || exit 1
cd "$ROOT"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_bench_kernel_stats.csv"
# the same trace without the warm-up step (first launches carry one-off set-up of up to milliseconds): what bench.py's own
# HIP events, which bracket timed steps only, are to be compared with
python3 tools/summarize_trace.py "$(find "$OUT/stats" -name '*kernel_trace.csv' | head -1)" "$OUT/${TAG}_bench_kernel_steady.csv" 1 10
python3 tools/summarize_pmc.py "$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)" \
        "$(find "$OUT/write" -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_hbm_traffic.json"
python3 tools/summarize_sq.py "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_sq_counters.json" \
        "$(find "$OUT/sq1" -name '*counter_collection.csv' | head -1)" "$(find "$OUT/sq2" -name '*counter_collection.csv' | head -1)"
ls -la "$OUT"/*.csv "$OUT"/*.json
