#!/bin/bash
# Sanitizer runs of the threaded host layer (ssvio_amd/host: prefetcher threads, Backend.Async worker + map mutex) and of
# the CPU oracle, in the BUILD container (no GPU needed: the host tests run on the oracle Compute).
#   tools/sanitize.sh            -> profiles/<round>/sanitizers.log
# ASan + UBSan: every test of tests/test_host.py (PNG fuzzing, units, whole sequences, asynchronous backend).
# TSan: the sequence tests incl. Backend.Async (the data-race check of the keyframe queue / map mutex protocol).
set -u
cd "$(dirname "$0")/.."
LOG=${1:-profiles/r02/sanitizers.log}
mkdir -p "$(dirname "$LOG")"
{
  echo "# $(date -u +%Y-%m-%dT%H:%MZ)  g++ $(g++ -dumpversion)"
  echo "## ASan + UBSan (detect_leaks=1, halt_on_error=1): python -m pytest tests/test_host.py"
  SSX_SANITIZE=asan ASAN_OPTIONS=detect_leaks=1:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests/test_host.py -q -x 2>&1 | tail -15
  echo "## TSan (halt_on_error=1): sequences on the oracle incl. Backend.Async (3 repetitions)"
  SSX_SANITIZE=tsan TSAN_OPTIONS=halt_on_error=1:second_deadlock_stack=1 \
    python -m pytest tests/test_host.py -q -x -k "state_machine or tracking_between or asynchronous or forward_drive" 2>&1 | tail -15
} | tee "$LOG"
