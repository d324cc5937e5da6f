#!/bin/bash
# CPU-only sanitizer runs of the native code that is not device code (never on the GPU box's kernels):
#   the oracle's C restatement (pthread pools: GemvPool, column split) and the C++ host mirror (program / op lifetimes),
#   built with ASan + UBSan, then with TSan, under the CPU tests that drive them.
#   tools/run_sanitizers.sh [asan|tsan ...]      (default: both)
# Reference precedent: std.testing.allocator leak / race checks, src/thread_pool.zig:180-199.
set -e -o pipefail
cd "$(dirname "$0")/.."
KINDS=${@:-asan tsan}
for san in $KINDS; do
  TESTS="tests/test_llama_host.py tests/test_oracle_w8a8.py tests/test_oracle_kat.py"
  # the 2-rank gloo run goes through torch.multiprocessing: fine under ASan; under TSan the spawned interpreters never finish
  # their rendezvous (the runtime serialises torch's own threads), so the thread sanitizer covers the pthread pools only
  if [ "$san" = asan ]; then TESTS="$TESTS tests/test_sharded_gloo.py"; fi
  make -s -C oracle SAN=$san
  python3 -c "import __graft_entry__ as g; g.build_host(san='$san')"
  rt=$(gcc -print-file-name=lib$san.so)
  echo "== $san: $TESTS"
  # detect_leaks=0: the interpreter itself is not leak-clean; the libraries' own errors (overflow, use-after-free, UB, races) abort
  env LD_PRELOAD="$rt" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
      TSAN_OPTIONS="halt_on_error=1 report_signal_unsafe=0" \
      ZGML_ORACLE_LIB=$PWD/oracle/_build/$san/libzgml_oracle.so ZGML_HOST_LIB=$PWD/zgml_amd/lib/$san/libzgml_host.so \
      timeout -k 10 600 python3 -m pytest $TESTS -x -q -m "not gpu" -p no:cacheprovider
done
echo SANITIZERS_OK
