#!/bin/bash
# Dev tool (GPU box): the GPU test suite against a build of the library whose HOST code is instrumented with UBSan
# (device code cannot be; GPU sanitizers are not available on this pool).  Build first, in the build container:
#   cd vtc_amd/csrc && hipcc -O3 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Xarch_host -fsanitize=undefined \
#       -Xarch_host -fno-sanitize=vptr,function -shared-libsan -shared vtc_api.hip -o ../libvtc_amd_ubsan.so
# Reports go to gpurun_out/ubsan_log.<pid>; none is written when nothing is found.
cp vtc_amd/libvtc_amd.so /tmp/base.so && cp vtc_amd/libvtc_amd_ubsan.so vtc_amd/libvtc_amd.so
export LD_LIBRARY_PATH=/opt/rocm/lib/llvm/lib/clang/22/lib/linux:$LD_LIBRARY_PATH
export UBSAN_OPTIONS=print_stacktrace=1:log_path=$PWD/gpurun_out/ubsan_log
rm -f gpurun_out/ubsan_log.*
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/r3c_ubsan_tests.log 2>&1
rc=$?
cp /tmp/base.so vtc_amd/libvtc_amd.so
tail -3 gpurun_out/r3c_ubsan_tests.log
ls gpurun_out/ | grep ubsan_log | head
cat gpurun_out/ubsan_log.* 2>/dev/null | grep "runtime error" | sort | uniq -c | sort -rn | head -20
exit $rc
