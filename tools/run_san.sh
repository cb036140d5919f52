#!/bin/bash
# CPU sanitizer run of the host code (AddressSanitizer + UndefinedBehaviorSanitizer), no GPU needed:
# builds libcugo_hip_san.so (make SAN=1) and runs the host test-suite against it — symbolic plan
# replay, synthetic generator, shard ranges, host-only cugo_chol_analyze, and the plan-only graphs
# that drive the whole flattening + structure build (tests/test_host.py).
#   tools/run_san.sh thread [pytest args] : ThreadSanitizer instead (libcugo_hip_tsan.so, make TSAN=1) —
#   the worker threads of the flattening, the structure build and the nested dissection
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "thread" ]; then
  shift
  make -C cuda-bundle-adjustment_amd TSAN=1 -j8 -s
  RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.tsan-x86_64.so)
  export LD_PRELOAD="$RT" TSAN_OPTIONS="halt_on_error=1:exitcode=66:report_signal_unsafe=0:ignore_noninstrumented_modules=1" OPENBLAS_NUM_THREADS=1 OMP_NUM_THREADS=1 \
         CUGO_LIB="$PWD/cuda-bundle-adjustment_amd/libcugo_hip_tsan.so"
  exec python -m pytest tests/test_host.py -x -q "$@"
fi
make -C cuda-bundle-adjustment_amd SAN=1 -j8 -s
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
# python itself is not instrumented: preload the runtime, leak checking off (the interpreter leaks)
export LD_PRELOAD="$RT" ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1" \
       UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1" CUGO_LIB="$PWD/cuda-bundle-adjustment_amd/libcugo_hip_san.so"
python -m pytest tests/test_host.py -x -q "$@"
