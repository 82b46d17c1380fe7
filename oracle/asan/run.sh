#!/bin/sh
# CPU sanitizer run (build container): oracle + host helpers under AddressSanitizer / UBSan.
#   make -C oracle asan-test
# python itself is not instrumented, so the sanitizer runtimes are preloaded; leak checking is off (the interpreter
# and torch keep memory until exit).  Selected tests: everything in tests/test_oracle_golden.py, and the tests of
# tests/test_cpu_product.py that only use the host helpers (game shims, known answers, noise spec, table-net twins).
set -e
cd "$(dirname "$0")/../.."
ASAN_SO=$(gcc -print-file-name=libasan.so)
UBSAN_SO=$(gcc -print-file-name=libubsan.so)
export LD_PRELOAD="$ASAN_SO:$UBSAN_SO"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1"
export CARO_UNDER_ASAN=1  # (tests that shorten themselves under the sanitizers look at this)
export UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
# CARO_ASAN_K narrows the selection further (tests/test_sanitizers.py runs the search / rules part only)
K="not test_library_exports and not test_config_struct and not test_engine_fails_loudly and not test_winograd2d_predicate and not test_hot_kernels and not test_net_kernel_touches_m0"
if [ -n "$CARO_ASAN_K" ]; then K="($K) and ($CARO_ASAN_K)"; fi
exec python3 -m pytest -q -p oracle.asan.plugin -p no:cacheprovider tests/test_oracle_golden.py tests/test_cpu_product.py \
  -k "$K" "$@"
