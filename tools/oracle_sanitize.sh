#!/bin/bash
# The CPU test suite (-m "not gpu": oracle against the golden vectors, the host-logic models, the gloo world-2 merge) with the
# oracle built under AddressSanitizer + UndefinedBehaviorSanitizer.  CPU only: GPU sanitizers are not available on this pool.
# Usage: tools/oracle_sanitize.sh [log]     (exit code = pytest's; any "runtime error" line of UBSan fails the run too)
cd "$(dirname "$0")/.." || exit 1
LOG=${1:-/tmp/oracle_sanitize.log}
make -C oracle -s sanitize || exit 1
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0 \
  VDB_ORACLE_SO=$PWD/oracle/_san/libvdb_oracle_asan.so \
  timeout 2400 python -m pytest tests/ -x -q -s -m "not gpu" -p no:cacheprovider > "$LOG" 2>&1
rc=$?
n=$(grep -c "runtime error\|ERROR: AddressSanitizer" "$LOG")
tail -1 "$LOG"; echo "sanitizer findings: $n"
[ "$rc" -eq 0 ] && [ "$n" -eq 0 ]
