#!/bin/bash
# Dev tool: A/B of tools/gemm_bench_<variant> builds (e.g. LLVM scheduler options) on the model's GEMM shapes, alternating passes.
#   VARIANTS="base nopost" PASSES=3 ./tools/sched_sweep.sh
for pass in $(seq 1 ${PASSES:-2}); do
  for v in $VARIANTS; do
    echo "== $v pass $pass"
    for spec in "conv2 256" "qkvL 192" "ffn1L 256" "oprojL 192" "ffn2L 192"; do
      set -- $spec
      timeout -k 5 60 ./tools/gemm_bench_$v 30 "$1 " $2 | grep -v "^ *check" || exit 1
    done
  done
done
