#!/bin/bash
# Runs the CPU test suite (-m "not gpu") with the oracle and the host helpers built under AddressSanitizer +
# UndefinedBehaviorSanitizer (make -C oracle sanitize; make -C vulkan-compute-tests_amd host-sanitize), then the progressive PNG encoder's
# threads under ThreadSanitizer (tools/tsan_png.cpp).  CPU build only.
# The Python interpreter is not instrumented, so libasan is preloaded; leak checking is off (CPython's own allocations).
#   tools/run_cpu_sanitizers.sh [log]          default log: profiles/r06_cpu_sanitizers.log
set -o pipefail
cd "$(dirname "$0")/.."
log=${1:-profiles/r06_cpu_sanitizers.log}
make -s -C oracle sanitize && make -s -C vulkan-compute-tests_amd host-sanitize || exit 1
asan=$(gcc -print-file-name=libasan.so)
{
  echo "# $(date -u +%F) $(gcc --version | head -1): -fsanitize=address,undefined -fno-sanitize-recover=all"
  echo "# oracle/_san/liboracle.so (oracle.cpp, oracle_core.h), vulkan-compute-tests_amd/lib_san/libmc_hostutil.so (hostutil_c.cpp, pngWriter.cpp)"
  LD_PRELOAD=$asan ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    MC_ORACLE_LIB_PATH=$PWD/oracle/_san/liboracle.so MC_HOSTUTIL_LIB_PATH=$PWD/vulkan-compute-tests_amd/lib_san/libmc_hostutil.so \
    python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1
  echo "exit code: $?"
  # the progressive PNG encoder under ThreadSanitizer (its workers read rows while later bands are still being written): tools/tsan_png.cpp
  echo "# -fsanitize=thread: tools/tsan_png.cpp + hostutil_c.cpp + pngWriter.cpp (pngwriter::Progressive against the one-shot encoder, 1 / 4 / 8 threads, 1 .. 700 bands)"
  mkdir -p tools/bin && g++ -std=c++17 -O1 -g -fsanitize=thread -Iinclude tools/tsan_png.cpp vulkan-compute-tests_amd/host/hostutil_c.cpp \
    vulkan-compute-tests_amd/host/pngWriter.cpp -o tools/bin/tsan_png -lz -lpthread && tools/bin/tsan_png 2>&1 | tail -20
  echo "exit code: $?"
} | tee "$log"
